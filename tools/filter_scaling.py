#!/usr/bin/env python3
"""Is the per-pixel trace kernel of a filter frame bound by throughput or by the length of a thread?  Its duration by frame size and
samples per pixel (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(sys.argv[1] if len(sys.argv) > 1 else "cornell_obj")
ctx = capi.Context(0)
ctx.update_scene(sc)
for (w, h) in ((960, 540), (1920, 1080), (3840, 2160)):
    for spp in (1, 2, 4, 8):
        p = sc.frame_params(width=w, height=h, samples=spp, use_filter=1)
        for _ in range(2): ctx.render(p)
        ms = []
        for _ in range(4):
            ctx.render(p)
            ms.append(ctx.last_frame_ms()[1])
        print("%4dx%4d %d spp: trace kernel %.3f ms = %.1f ns per pixel-sample" % (w, h, spp, min(ms), min(ms) * 1e6 / (w * h * spp)), flush=True)
