#!/usr/bin/env python3
"""Per-rank frame times of a tile-sharded frame, every rank's share rendered in turn on one GPU.  usage: tile_balance.py N [tile_rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
times = []
for r in range(n):
    p = sc.frame_params(use_filter=0)
    p.tile_rows, p.tile_count, p.tile_index = rows, n, r
    for _ in range(2): ctx.render(p)
    times.append(min(ctx.render(p) and ctx.last_frame_ms()[0] for _ in range(5)))
print("N=%d strips of %d rows: per-rank ms %s  max %.3f  mean %.3f" % (n, rows, " ".join("%.2f" % t for t in times), max(times), sum(times) / n))
