#!/bin/bash
# Run bench.py once per variant library in build/variants (on the GPU box); prints value / ms per step.
for so in build/variants/lib_*.so; do
  FLX_LIB=$PWD/$so timeout -k 10 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s %9.1f Mray/s  %7.3f ms/step  kernel %.3f ms' % ('$(basename $so)', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
