#!/usr/bin/env python3
"""Frame kernel with the front of the frame inside the launch (flx_set_frame_front 1: primary rays and bounce-0 shading by its shade waves) against
k_primary + k_wf_shade0 in front of it (0): ms per frame for the whole frame and for one rank's share of it, and that both give the same frame and
the same work counters.  GPU box.       usage: front_time.py [N ...]      env FLX_WORKLOAD=dragon|dragon_4k|theater"""
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
ctx.set_wavefront_organisation(2)
size = dict(width=3840, height=2160) if work == "dragon_4k" else {}
extra = dict(samples=int(os.environ["FLX_SPP"])) if "FLX_SPP" in os.environ else {}
shares = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
print("%-22s" % "front of the frame" + "".join("   1/%-2d share  " % n for n in shares) + "  (ms per frame: median of 20, min)")
frames, counters = {}, {}
for front in (0, 2, 3, 0, 2, 3):
    ctx.set_frame_front(front)
    row = []
    for n in shares:
        p = sc.frame_params(use_filter=0, **size, **extra)
        if n > 1:
            p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
        for _ in range(3):
            ctx.render(p)
        ms = []
        for _ in range(20):
            img = ctx.render(p)[0]
            ms.append(ctx.last_frame_ms()[0])
        row.append("%6.3f (%5.3f)" % (float(np.median(ms)), min(ms)))
        frames.setdefault(n, []).append(np.asarray(img).copy())
        counters.setdefault(n, []).append(ctx.render(p, counters=True)[1])
    print("%-22s" % {2: "inside the launch", 0: "kernels in front", 3: "ONE kernel in front"}[front] + " ".join(row))
for n, f in frames.items():
    print("1/%d share: same frame: %s   same counters: %s" % (n, all(np.array_equal(f[0].view(np.uint32), g.view(np.uint32)) for g in f[1:]),
                                                               all(c == counters[n][0] for c in counters[n][1:])))
