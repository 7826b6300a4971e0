#!/bin/bash
# Run bench.py once per variant library in build/variants (on the GPU box); prints value / ms per step.
# Every variant's stderr is kept (build/variants/<name>.err) and its exit status printed: a variant that
# produced no bench line says why.
for so in build/variants/lib_*.so; do
  name=$(basename $so .so)
  FLX_LIB=$PWD/$so timeout -k 10 ${TMO:-120} python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-pmc "$@" > build/variants/$name.out 2> build/variants/$name.err
  rc=$?
  python - "$name" "$rc" build/variants/$name.out <<'PY'
import json, sys
name, rc, path = sys.argv[1:4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    extra = ""
    if "batched" in d:
        extra = "  batched %.3f ms/frame" % d["batched"]["ms_per_frame"]
    print("%-28s rc=%s %9.1f Mray/s  %7.3f ms/step  kernel %.3f ms%s" % (name, rc, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], extra))
except Exception as e:
    print("%-28s rc=%s NO BENCH LINE (%s); stderr tail: %s" % (name, rc, e, open(path[:-4] + ".err").read()[-400:].replace("\n", " | ")))
PY
done
