for v in nowarm warm nowarm warm; do echo "== $v"; FLX_LIB=$PWD/build/variants/lib_$v.so FRONT=3 N=20 bash tools/kt_eighth.sh p_$v 2>&1 | grep -E "front|frame "; done
