#!/bin/bash
# usage: kt.sh <name> (env passes through)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$1 -- python3 $GRAFT_REPO_ROOT/tools/eighth_trace.py > $GRAFT_REPO_ROOT/gpurun_out/kt_$1.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^frame" gpurun_out/kt_$1.log
python3 - $1 <<PY
import csv, glob, sys
f = glob.glob("gpurun_out/kt_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in csv.reader(open(f)):
    if r[0].startswith("__amd") or r[0] == "Name": continue
    print("   %-60s calls %s avg %.1f us" % (r[0][:60], r[1], float(r[3]) / 1e3))
PY
