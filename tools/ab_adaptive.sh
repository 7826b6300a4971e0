# head (build/variants/lib_head.so: the commit before) against the working tree with the adaptive tile order off and on: whole frame, a half, a quarter, an eighth, a sixteenth
for cfg in "" "" "FLX_TILES=2" "FLX_TILES=4" "FLX_TILES=4" "FLX_TILES=8" "FLX_TILES=16"; do
  echo "== ${cfg:-whole frame}"
  env $cfg python tools/variant_time.py head
  env $cfg FLX_ADAPTIVE=0 python tools/variant_time.py now
  env $cfg FLX_ADAPTIVE=1 python tools/variant_time.py now
done
