#!/usr/bin/env python3
"""Frame time of ONE rank's share of the tile-sharded dragon frame (what each of N GPUs renders before the gather), on one GPU, for
both organisations of the bounce loop (rounds: a kernel pair per bounce; frame kernel: one persistent launch).  GPU box.
usage: share_time.py [N ...]      env FLX_WORKLOAD=dragon|dragon_4k|theater, FLX_ORGS=1,2"""
import os, sys
import numpy as np
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
print("%-14s" % "organisation" + "".join("   1/%-2d share" % n for n in shares) + "    (ms per frame: median of 20, min)   speed-up of the 1/%d share over the whole frame" % shares[-1])
for org in [int(a) for a in os.environ.get("FLX_ORGS", "1,2").split(",")]:
    ctx.set_wavefront_organisation(org)
    row, med = [], []
    for n in shares:
        p = sc.frame_params(use_filter=0, **size)
        if n > 1:
            p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
        for _ in range(3):
            ctx.render(p)
        ms = []
        for _ in range(20):
            ctx.render(p)
            ms.append(ctx.last_frame_ms()[0])
        med.append(float(np.median(ms)))
        row.append("%6.3f (%5.3f)" % (med[-1], min(ms)))
    print("%-14s" % {1: "rounds", 2: "frame kernel"}[org] + " ".join(row) + "    %.2fx" % (med[0] / med[-1]))
