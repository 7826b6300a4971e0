#!/usr/bin/env python3
"""A rank's eighth of the dragon frame, rendered synchronously N times (for rocprofv3 --kernel-trace --stats): which kernels does such a frame consist of?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0); ctx.update_scene(sc); ctx.set_frame_chain(0)
if os.environ.get("FRONT"): ctx.set_frame_front(int(os.environ["FRONT"]))
p = sc.frame_params(width=1920, height=1080, samples=8, max_reflections=4, use_filter=0)
p.tile_rows, p.tile_count, p.tile_index = 8, int(os.environ.get("FLX_TILES", "8")), 0
for _ in range(int(os.environ.get("N", "30"))): ctx.render(p)
print("frame %.3f ms, kernel %.3f ms" % ctx.last_frame_ms())
