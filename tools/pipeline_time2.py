#!/usr/bin/env python3
"""Small scenes: the persistent path kernel (pipeline 2), the per-pixel kernel (1) and the wavefront pipeline as rounds / frame kernel. GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
ctx = capi.Context(0)
for name, kw in [("theater", dict(width=1920, height=1080, samples=16, max_reflections=6)), ("theater", dict(width=1920, height=1080, samples=4, max_reflections=3)),
                 ("cornell_obj", dict(width=1920, height=1080, samples=4, max_reflections=3)), ("cornell", dict(width=1920, height=1080, samples=8, max_reflections=4))]:
    sc = Scene.golden(name)
    ctx.update_scene(sc)
    p = sc.frame_params(use_filter=0, **kw)
    row = []
    for pipe, org in [(1, 0), (2, 0), (3, 1), (3, 2)]:
        ctx.set_pipeline(pipe); ctx.set_wavefront_organisation(org)
        ms = []
        for i in range(8):
            ctx.render(p)
            if i >= 2: ms.append(ctx.last_frame_ms()[0])
        row.append(float(np.median(ms)))
    print("%-12s %s: per-pixel %.3f  persistent paths %.3f  wavefront rounds %.3f  frame kernel %.3f ms" % (name, kw, *row))
