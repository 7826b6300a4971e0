#!/usr/bin/env python3
"""Small scenes, filter off: per-pixel kernel against persistent path kernel by samples x bounces (where should flx_set_pipeline(0) switch?); GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
for name in ["cornell", "cornell_obj", "theater"]:
    sc = Scene.golden(name)
    ctx = capi.Context(0)
    ctx.update_scene(sc)
    for spp, b in [(1, 1), (2, 2), (4, 3), (8, 3), (8, 6), (16, 6)]:
        p = sc.frame_params(width=1920, height=1080, samples=spp, max_reflections=b, use_filter=0)
        row = []
        for pipe in (1, 2):
            ctx.set_pipeline(pipe)
            for _ in range(2): ctx.render(p)
            ms = []
            for _ in range(5):
                ctx.render(p)
                ms.append(ctx.last_frame_ms()[0])
            row.append(min(ms))
        print("%-12s %2d spp %d bounces: per-pixel %.3f ms, persistent paths %.3f ms" % (name, spp, b, row[0], row[1]), flush=True)
    ctx.close()
