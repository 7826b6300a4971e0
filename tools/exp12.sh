mkdir -p gpurun_out/exp12
for v in c_base f64 f512 fall; do
  echo "== $v"
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/chain_modes.py 2>&1 | grep "chain 1, 400"
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 8 2>&1 | grep -A3 "^   7 " | head -3
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/chain_stats.py --count 8 --index 0 --frames 8 2>&1 | tail -4
done
