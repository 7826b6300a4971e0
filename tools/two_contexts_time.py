#!/usr/bin/env python3
"""Do two frames in flight on two streams overlap on the GPU?  Frames of the dragon workload rendered alternately by two contexts
of the same GPU (each with its own stream and workspace), nothing waits on the host; against the same frames on one context (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
import torch
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(sys.argv[1] if len(sys.argv) > 1 else "dragon")
p = sc.frame_params(use_filter=0)
ctxs = [capi.Context(0) for _ in range(3)]
outs = [torch.zeros((p.height, p.width, 4), dtype=torch.float32, device="cuda") for _ in ctxs]
for c in ctxs: c.update_scene(sc)
for n in (1, 2, 3):
    for k in range(6): ctxs[k % n].render_device(p, outs[k % n].data_ptr())
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    K = 60
    for k in range(K): ctxs[k % n].render_device(p, outs[k % n].data_ptr())
    for c in ctxs: c.sync()
    print("%d context(s) in rotation: %.3f ms per frame" % (n, (time.perf_counter() - t0) / K * 1e3), flush=True)
