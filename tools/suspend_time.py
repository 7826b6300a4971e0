#!/usr/bin/env python3
"""Frame time (one frame per pass) of the dragon frame by walk scheduler / suspension setting (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
import torch
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(sys.argv[1] if len(sys.argv) > 1 else "dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0)
out = torch.zeros((p.height, p.width, 4), dtype=torch.float32, device="cuda")
for sched, susp in ((0, 0), (0, 16), (0, 64), (0, 128), (0, 256), (0, 512), (2, 16), (2, 64)):
    ctx.set_walk_scheduler(sched, susp)
    for _ in range(3): ctx.render_device(p, out.data_ptr())
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(20): ctx.render_device(p, out.data_ptr())
    ctx.sync()
    print("scheduler %d suspend %3d: %.3f ms per frame" % (sched, susp, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
