#!/usr/bin/env python3
"""A/B of the per-pixel kernel: samples one after the other (k_trace_pixels) against side by side (k_trace_samples), 1080p filter frames of the BASELINE scenes at their
spp / bounces; frame and trace-kernel time (flx_last_frame_ms, min of 9) and whether the two frames are the same bits.  usage: sample_parallel_time.py [scene ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
for name in sys.argv[1:] or ["cornell_obj", "cornell", "theater", "dragon"]:
    sc = Scene.golden(name)
    ctx = capi.Context(0)
    ctx.update_scene(sc)
    for filt in (1, 0):
        p = sc.frame_params(width=1920, height=1080, use_filter=filt)
        if not filt:
            ctx.set_pipeline(1)
        frames = []
        for on in (0, 1):
            ctx.set_sample_parallel(on)
            for _ in range(3): out = ctx.render(p)[0]
            ms, km = [], []
            for _ in range(9):
                out = ctx.render(p)[0]
                a, b = ctx.last_frame_ms()
                ms.append(a); km.append(b)
            frames.append(out)
            print("%-12s %2d spp %d bounces, filter %s, samples %s: frame %.3f ms, trace kernel %.3f ms" % (name, p.samples, p.max_reflections, "on " if filt else "off",
                  "side by side    " if on else "one after another", min(ms), min(km)), flush=True)
        print("     same bits: %s" % bool(np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))), flush=True)
    ctx.close()
