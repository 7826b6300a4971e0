#!/bin/bash
# tools/tile_time.py (one rank's share of a tile-sharded frame) once per variant library in build/variants (GPU box).
for so in build/variants/lib_*.so; do
  echo "== $(basename $so)"
  FLX_LIB=$PWD/$so timeout -k 10 120 python tools/tile_time.py "$@" 2>&1 | tail -4
done
