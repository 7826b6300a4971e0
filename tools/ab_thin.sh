for rep in 1 2; do for t in 8 16; do echo "== 1/$t"; FLX_TILES=$t python tools/variant_time.py base fs1 fs3 dw8 dw32; done; done
