set -x
mkdir -p gpurun_out/exp1
for v in base t512 t512x2; do
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/share_all.py --workload dragon --indices 0,3 --check > gpurun_out/exp1/$v.txt 2>&1 || echo "variant $v failed rc=$?"
  tail -8 gpurun_out/exp1/$v.txt
done
FLX_LIB=$PWD/build/variants/lib_base.so timeout -k 10 300 python tools/share_all.py --workload dragon --check > gpurun_out/exp1/base_all8.txt 2>&1; tail -12 gpurun_out/exp1/base_all8.txt
