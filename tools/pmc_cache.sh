#!/bin/bash
# Cache counters of one workload's kernels: is the scene L2-resident, what do the fabric-side read requests look like?
# ONE counter per rocprofv3 pass, TCP_* / TCC_* only (no TA_*, no GRBM_*: profiles/r03_pmc_hang_record.txt), every pass under a timeout.
#   tools/pmc_cache.sh <out dir under gpurun_out/> "<pmc_pass.py args>"        GPU box
out=$1; args=$2
tools/pmc_run.sh "$out" "$args" \
  "TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum" "TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum" "TCC_HIT_sum" "TCC_MISS_sum" \
  "TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum" "FETCH_SIZE" "WRITE_SIZE"
