for t in "" "FLX_TILES=2"; do
echo "== ${t:-whole}"
env $t python tools/variant_time.py now
env $t FLX_ORDER_MODE=2 python tools/variant_time.py allstamp
env $t FLX_ORDER_MODE=0 python tools/variant_time.py allstamp
env $t python tools/variant_time.py now
env $t FLX_ORDER_MODE=2 python tools/variant_time.py allstamp
env $t FLX_ORDER_MODE=0 python tools/variant_time.py allstamp
done
