#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of a few frames, once per variant library in build/variants; GPU box.
#   tools/ab_kstats.sh <kernel name substring> [pmc_pass.py arguments]      e.g.  tools/ab_kstats.sh k_primary --workload dragon --frames 10
pat=$1; shift
root=$PWD
for so in build/variants/lib_*.so; do
  name=$(basename $so .so)
  out=$root/gpurun_out/kstats_$name
  rm -rf $out; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp FLX_LIB=$root/$so timeout -k 10 ${TMO:-180} rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/pmc_pass.py "$@" > $out/log.txt 2>&1)
  python3 - "$name" "$pat" $out <<'PY'
import csv, glob, sys
name, pat, out = sys.argv[1:4]
files = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
if not files: print("%-24s no kernel stats" % name); sys.exit(0)
for r in csv.DictReader(open(files[0])):
    if pat in r["Name"]: print("%-24s %-44s calls %4s  avg %.4f ms" % (name, r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
