#!/usr/bin/env python3
"""What would MORE than two frame lanes buy?  K contexts on one GPU (each with one lane and its own copy of the scene), frames begun round robin so that
K frames are in flight, each context rendering the same share of the dragon frame: ms per frame (wall clock over 120 frames).  GPU box.
usage: lanes_k_time.py [share ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
shares = [int(a) for a in sys.argv[1:]] or [1, 8]
KMAX = 6
ctxs = []
for _ in range(KMAX):
    c = capi.Context(0)
    c.update_scene(sc)
    c.set_pipeline(3)
    c.set_frame_lanes(1)
    ctxs.append(c)
FRAMES = 120
print("%-22s" % "frames in flight" + "".join("   1/%-2d share" % n for n in shares) + "    (ms per frame)")
for k in range(1, KMAX + 1):
    row = []
    for n in shares:
        p = sc.frame_params(use_filter=0)
        if n > 1:
            p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
        best = 1e9
        for rep in range(3):
            for i in range(k):
                ctxs[i].frame_begin(p, device=True)
            for i in range(2 * k):                       # warm
                ctxs[i % k].frame_end(); ctxs[i % k].frame_begin(p, device=True)
            t0 = time.perf_counter()
            for i in range(FRAMES):
                ctxs[i % k].frame_end(); ctxs[i % k].frame_begin(p, device=True)
            dt = time.perf_counter() - t0
            for i in range(k):
                ctxs[(FRAMES + i) % k].frame_end()
            best = min(best, dt * 1e3 / FRAMES)
        row.append("%13.3f" % best)
    print("%-22d" % k + " ".join(row))
