#!/bin/bash
# Hardware counters of one workload's kernels, several rocprofv3 --pmc passes (one counter set per pass; --pmc alone, no tracing):
#   tools/pmc_run.sh <out dir under gpurun_out/> "<pmc_pass.py args>" "SET1 counters..." "SET2 counters..." ...
# Prints, per pass, the mean of every counter per kernel (tools/pmc_summary.py).  Run on the GPU box.
out=$1; args=$2; shift 2
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i + 1))
  # a counter set the hardware cannot collect makes rocprofv3 abort inside the program's first HIP call and then sit in its own signal
  # handler ("finalizing after signal 6...", profiles/r03_pmc_hang_record.txt): every pass runs under a timeout with a hard kill behind it
  timeout -k 10 ${TMO:-300} rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python3 tools/pmc_pass.py $args > "$out/p$i.log" 2>&1 || echo "pass $i failed or timed out (rc $?), see $out/p$i.log"
  for f in $(find "$out/p$i" -name "*counter_collection.csv"); do python3 tools/pmc_summary.py "$f"; done
done
