mkdir -p gpurun_out/exp21
for w in dragon dragon_4k theater; do
  timeout -k 10 500 python tools/share_all.py --workload $w --frames 40 --check > gpurun_out/exp21/share_$w.txt 2>&1 || echo "$w failed rc=$?"
  cat gpurun_out/exp21/share_$w.txt
done
