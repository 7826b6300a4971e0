mkdir -p gpurun_out/exp2
for v in base t512 t256; do
  echo "== $v: K frames in flight on K contexts, 1/8 share"
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/lanes_k_time.py 8 > gpurun_out/exp2/lanesk_$v.txt 2>&1 || echo "variant $v failed rc=$?"
  cat gpurun_out/exp2/lanesk_$v.txt
done
for v in base t512; do
  echo "== $v: front forced inside the frame kernel"
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/share_all.py --workload dragon --indices 0,3 --front 2 --check > gpurun_out/exp2/front2_$v.txt 2>&1 || echo "variant $v failed rc=$?"
  tail -5 gpurun_out/exp2/front2_$v.txt
done
