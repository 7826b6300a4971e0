#!/usr/bin/env python3
"""The filter chain's kernels on a frame nobody covers (camera turned away) against the cornell.obj frame: what do the passes cost
before any tap is taken?  Run under rocprofv3 --kernel-trace --stats; GPU box.  usage: filter_empty_time.py [away]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene, view_matrix
sc = Scene.golden("cornell_obj")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(width=1920, height=1080, use_filter=1)
if len(sys.argv) > 1 and sys.argv[1] == "away":
    cam = sc.meta["camera"]
    p.view_matrix[:] = view_matrix(cam["fx"] + 3.14159, cam["fy"], cam["fov"], p.width, p.height).tolist()
for _ in range(20): ctx.render(p)
ctx.close()
