#!/usr/bin/env python3
"""A rank's share of the 1080p dragon frame ONE FRAME AT A TIME (begin, end, begin, end: nothing in flight beside it) through the frame server's launch against
its own launches (k_wf_front + k_wf_frame + k_resolve per frame).  GPU box.  usage: share_one_at_a_time.py [--count 8] [--index 3]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
ap = argparse.ArgumentParser()
ap.add_argument("--count", type=int, default=8)
ap.add_argument("--index", type=int, default=3)
ap.add_argument("--frames", type=int, default=100)
a = ap.parse_args()
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0)
if a.count > 1:
    p.tile_rows, p.tile_count, p.tile_index = 8, a.count, a.index
want = ctx.render(p)[0]
for lanes, chain, label in ((1, 0, "own launches, one lane"), (2, 3, "frame server (2 slots), one frame posted at a time"), (3, 3, "frame server (3 slots), one frame posted at a time")):
    ctx.set_frame_lanes(lanes)
    ctx.set_frame_chain(chain)
    best = 1e9
    for rep in range(3):
        for _ in range(5):
            ctx.frame_begin(p, device=True); ctx.frame_end()
        t0 = time.perf_counter()
        for _ in range(a.frames):
            ctx.frame_begin(p, device=True); ctx.frame_end()
        best = min(best, (time.perf_counter() - t0) * 1e3 / a.frames)
    ctx.frame_begin(p)
    kind = ctx.last_chained()
    got = ctx.frame_end()[0]
    print("%-56s %.3f ms per frame  (kind %d)  %s" % (label, best, kind, "equal" if np.array_equal(got.view(np.uint32), want.view(np.uint32)) else "DIFFERS"), flush=True)
