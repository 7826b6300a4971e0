#!/usr/bin/env python3
"""Frame time of a BASELINE scene under each kernel organisation (flx_set_pipeline 1 / 2 / 3); GPU box.  usage: pipeline_time.py scene"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
for name in sys.argv[1:] or ["cornell", "cornell_obj", "dragon", "theater"]:
    sc = Scene.golden(name)
    ctx = capi.Context(0)
    ctx.update_scene(sc)
    p = sc.frame_params(width=1920, height=1080, use_filter=0)
    for pipe in (3, 2, 1):
        ctx.set_pipeline(pipe)
        for _ in range(2): ctx.render(p)
        ms = []
        for _ in range(5):
            ctx.render(p)
            ms.append(ctx.last_frame_ms()[0])
        print("%-12s pipeline %d: %.3f ms per frame" % (name, pipe, min(ms)))
    ctx.close()
