#!/bin/bash
# The round's measurements on the GPU box, into gpurun_out/<dir>/ (copy what is to be judged into profiles/):
#   bench.py for every BASELINE workload (each with its own counter passes), per-kernel statistics of the headline workload and of the filter
#   frame (rocprofv3 --kernel-trace --stats), the JavaScript frame loop, the share scaling and the two-rank rehearsal of bench.py --gpus 2.
#   tools/round_bench.sh <dir name>
out=gpurun_out/$1; mkdir -p $out
root=$PWD
for w in dragon cornell cornell_obj theater dragon_4k dragon_100k; do
  timeout -k 10 400 python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err; echo "bench $w rc $?"
done
(cd /tmp && export TMPDIR=/tmp && for w in dragon cornell_obj; do timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt_$w -- python3 $root/tools/pmc_pass.py --workload $w --frames 20 > $root/$out/kt_$w.log 2>&1; cp $(find $root/$out/kt_$w -name "*kernel_stats.csv" | head -1) $root/$out/kernel_stats_$w.csv; done)
timeout -k 10 200 python tools/share_time.py > $out/share_time.txt 2>&1
timeout -k 10 200 node tools/js_loop.js tests/golden/ref_dragon.flxs.gz --frames 300 > $out/js_loop.txt 2>&1
timeout -k 10 200 node tools/js_loop.js tests/golden/ref_dragon.flxs.gz --frames 300 --move 1 >> $out/js_loop.txt 2>&1
timeout -k 10 600 python bench.py --gpus 2 --one-device --steps 10 --batch 4 --no-cpu-baseline > $out/bench_2ranks_one_device.json 2> $out/bench_2ranks_one_device.err; echo "bench 2 ranks rc $?"
FLX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29717 timeout -k 10 600 python bench.py --steps 20 --no-cpu-baseline > $out/bench_rccl_one_rank.json 2> $out/bench_rccl_one_rank.err; echo "bench rccl 1 rank rc $?"
python3 - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print("%-34s %9.1f Mray/s %8.3f ms (%s)  one at a time %s  pipelined %s  batched %s | %s %.3f ms frac %s lane %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d.get("headline", {}).get("mode", "one_frame_per_pass"),
              ("%.3f" % d["one_frame_per_pass"]["ms_per_step"]) if d.get("one_frame_per_pass") else "-",
              ("%.3f" % d["pipelined"]["ms_per_frame"]) if d.get("pipelined") else "-", ("%.3f" % d["batched"]["ms_per_frame"]) if d.get("batched") else "-",
              r["kernel"].split(" ")[0], r["kernel_ms"], r["frac"] and round(r["frac"], 3), r["valu_lane_utilisation"] and round(r["valu_lane_utilisation"], 3)))
    except Exception as e:
        print(os.path.basename(f), "NO LINE", e)
PY
