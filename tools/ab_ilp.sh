for rep in 1 2; do python tools/variant_time.py base wfilp wfsvilp; done
for v in base wfilp wfsvilp base wfsvilp; do echo "== $v"; FLX_LIB=$PWD/build/variants/lib_$v.so python tools/share_all.py --count 8 --indices 0,3 --no-single 2>&1 | tail -4; done
