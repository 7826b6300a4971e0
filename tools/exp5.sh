mkdir -p gpurun_out/exp5
for w in theater dragon_4k; do
  timeout -k 10 400 python tools/share_all.py --workload $w --frames 30 > gpurun_out/exp5/share_$w.txt 2>&1 || echo "$w failed rc=$?"
  cat gpurun_out/exp5/share_$w.txt
done
