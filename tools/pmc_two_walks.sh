#!/bin/bash
# Counters of the frame kernel with one and with two walk jobs per lane (profiles/r05_two_walks.txt): run on the GPU box from the repository root, with
# `make -C web-ray-tracer_amd/csrc EXPERIMENTS=1` built.  rocprofv3 --pmc alone (no trace domains), one pass per variant; the program after `--` is python3 itself.
set -u
OUT=${1:-gpurun_out/r5_two_walks}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export FLX_LIB="$GRAFT_REPO_ROOT/web-ray-tracer_amd/flexlight_hip/libflexlight_hip_experiments.so"
for j in 1 2; do
  export FLX_WALK_JOBS=$j
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY \
    --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_j$j" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_pass.py" --workload dragon --frames 3 > "$GRAFT_REPO_ROOT/$OUT/pmc_j$j.log" 2>&1 || exit 1
done
cd "$GRAFT_REPO_ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for j in (1, 2):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/pmc_j%d/**/*counter_collection.csv" % (out, j), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in rows.items():
        if "k_wf_frame" not in k:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        print("jobs %d  %s  launches %d" % (j, k, len(next(iter(c.values())))))
        print("   SQ_INSTS_VALU %.4g  SQ_INSTS_SALU %.4g  lane utilisation %.3f  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES %.3f  SQ_WAVES %d  SQ_BUSY_CYCLES %.4g" % (
            m["SQ_INSTS_VALU"], m["SQ_INSTS_SALU"], m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]) if m.get("SQ_ACTIVE_INST_VALU") else float("nan"),
            m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAVES"], m["SQ_BUSY_CYCLES"]))
PY
