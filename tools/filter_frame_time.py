#!/usr/bin/env python3
"""1080p filter frames of the BASELINE scenes (their spp / bounces): frame and trace-kernel time; GPU box.  usage: filter_frame_time.py [scene ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
for name in sys.argv[1:] or ["cornell_obj", "cornell", "theater", "dragon"]:
    sc = Scene.golden(name)
    ctx = capi.Context(0)
    ctx.update_scene(sc)
    p = sc.frame_params(width=1920, height=1080, use_filter=1)
    for _ in range(3): ctx.render(p)
    ms, km = [], []
    for _ in range(7):
        ctx.render(p)
        a, b = ctx.last_frame_ms()
        ms.append(a); km.append(b)
    print("%-12s %2d spp %d bounces, filter on: frame %.3f ms, trace kernel %.3f ms" % (name, p.samples, p.max_reflections, min(ms), min(km)), flush=True)
    ctx.close()
