#!/usr/bin/env python3
"""Two frames in flight on two lanes, with (--gathered) or without a one-rank communicator: run under rocprofv3 --kernel-trace and
tools/lanes_trace_report.py shows whether the lanes' kernels overlap on the GPU.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
gathered = "--gathered" in sys.argv
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0, tile=(8, 0, 1) if gathered else (0, 0, 0))
if gathered:
    ctx.comm_init_rank(capi.comm_unique_id(), 1, 0)
import time
for phase in range(2):
    t0 = time.time()
    n = 6 if phase == 0 else 24
    for i in range(n):
        if gathered: ctx.frame_begin_gathered(p, root=0)
        else: ctx.frame_begin(p, device=True)
        if ctx.frames_in_flight() == 2: ctx.frame_end()
    while ctx.frames_in_flight(): ctx.frame_end()
    ctx.sync()
    dt = time.time() - t0
print("gathered" if gathered else "plain", "%.3f ms per frame" % (dt / n * 1e3))
