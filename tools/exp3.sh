mkdir -p gpurun_out/exp3
for v in base t512x2; do
echo "== batches, $v (frame kernel, automatic front)"
FLX_LIB=$PWD/build/variants/lib_$v.so FLX_BATCHES=1,2,3,4,8 timeout -k 10 300 python tools/batch_time.py 8 2>&1 | tee gpurun_out/exp3/batch_$v.txt
done
