#!/bin/bash
# Run a diagnostic script once per variant library in build/variants (on the GPU box).
for so in build/variants/lib_*.so; do
  echo "== $(basename $so)"
  FLX_LIB=$PWD/$so timeout -k 10 120 python "$@" 2>&1 | tail -${TAIL:-3}
done
