#!/usr/bin/env python3
"""Frame time with and without the shading's per-triangle table (flx_debug_set_angle_table) on the BASELINE workloads; GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
for name, kw in (("dragon", {}), ("theater", {}), ("cornell_obj", {}), ("dragon", dict(width=3840, height=2160))):
    sc = Scene.golden(name)
    ctx = capi.Context(0)
    ctx.update_scene(sc)
    p = sc.frame_params(**kw)
    res = {}
    for rep in range(2):
        for on in (0, 1):
            ctx.set_angle_table(on)
            for _ in range(3): ctx.render(p)
            ms = []
            for _ in range(15):
                ctx.render(p)
                ms.append(ctx.last_frame_ms()[0])
            res.setdefault(on, []).append(float(np.median(ms)))
    print("%-12s %4dx%-4d  every shade computes it: %s ms   table: %s ms" % (name, p.width, p.height, " / ".join("%.3f" % v for v in res[0]), " / ".join("%.3f" % v for v in res[1])), flush=True)
    ctx.close()
