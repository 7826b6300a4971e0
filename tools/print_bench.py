#!/usr/bin/env python3
"""the figures of a bench.py line that matter at a glance: print_bench.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("%s: %.4f ms per step = %.1f %s (N = %d); dominant kernel %s %.4f ms, frac %s; pipelined %s; shared %s" % (
        f, d["ms_per_step"], d["value"], d["unit"], d["n_gpus"], (r.get("kernel") or "?").split(" (")[0], r.get("kernel_ms") or 0, r.get("frac"),
        (d.get("pipelined") or {}).get("ms_per_frame"), (d.get("shared") or {}).get("ms_per_frame")))
    if d.get("speedup"):
        s = d["speedup"]
        print("   one context %.3f ms; latency mode %.2fx; throughput mode %s" % (s["single_context_ms"], s["latency_mode"]["speedup"], {k: s["throughput_mode"][k] for k in ("mode", "speedup", "frames_in_flight")} if s.get("throughput_mode") else None))
