cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_sort_$1 -- python3 $GRAFT_REPO_ROOT/tools/tile_sort_time.py > $GRAFT_REPO_ROOT/gpurun_out/kt_sort_$1.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $1 <<PY
import csv, glob, sys
f = glob.glob("gpurun_out/kt_sort_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_tile_order" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for k, n in enumerate((4080, 8100, 16200, 32400)):
    seg = sorted(d[20 * k: 20 * k + 20])
    print("%6d tiles: k_tile_order median %.1f us (min %.1f)" % (n, seg[len(seg) // 2], seg[0]))
PY
