#!/usr/bin/env python3
"""One rank's share of the tile-sharded dragon frame in the FRAME LOOP (flx_frame_begin / _end, one frame in flight): ms per frame on one and
on two lanes, wall clock over 60 frames.  GPU box.   usage: share_lanes_time.py [N ...]   env FLX_WORKLOAD=dragon|dragon_4k|theater"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
work = os.environ.get("FLX_WORKLOAD", "dragon")
sc = Scene.golden("theater" if work == "theater" else "dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
ctx.set_pipeline(3)
size = dict(width=3840, height=2160) if work == "dragon_4k" else {}
shares = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
FRAMES = 60
print("%-10s" % "lanes" + "".join("   1/%-2d share" % n for n in shares) + "    (ms per frame, %d frames, one in flight)" % FRAMES)
for lanes in (1, 2):
    ctx.set_frame_lanes(lanes)
    row = []
    for n in shares:
        p = sc.frame_params(use_filter=0, **size)
        if n > 1:
            p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
        best = 1e9
        for rep in range(3):
            ctx.frame_begin(p, device=True)
            for _ in range(4):
                ctx.frame_begin(p, device=True)
                ctx.frame_end()
            t0 = time.perf_counter()
            for _ in range(FRAMES):
                ctx.frame_begin(p, device=True)
                ctx.frame_end()
            dt = time.perf_counter() - t0
            ctx.frame_end()
            best = min(best, dt * 1e3 / FRAMES)
        row.append("%13.3f" % best)
    print("%-10d" % lanes + " ".join(row))
