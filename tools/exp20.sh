for v in sh3 sh4 sh5 sh4r16; do
  echo "== $v"
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/server_check.py --count 8 --index 0 --no-check --modes 2 --stats 2>&1 | tail -2
  FLX_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 200 python tools/server_check.py --count 1 --no-check --modes 2 --stats --time-frames 60 2>&1 | tail -2
done
