#!/usr/bin/env python3
"""Whole-frame time of the wavefront pipeline as rounds and as the frame kernel over frame sizes and batch sizes (dragon, 8 spp, 4 bounces):
where is the crossover?  GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(os.environ.get("FLX_SCENE", "dragon"))
ctx = capi.Context(0)
ctx.update_scene(sc)
ctx.set_pipeline(3)
print("%-58s %10s %12s %14s %14s %14s" % ("frame", "rounds", "rounds, one", "frame kernel", "frame kernel,", "frame kernel +"))
print("%-58s %10s %12s %14s %14s %14s" % ("", "", "front kernel", "", "one front k.", "its front"))
for w, h, spp, frames in [(960, 540, 8, 1), (1280, 720, 8, 1), (1920, 1080, 2, 1), (1920, 1080, 8, 1), (2560, 1440, 8, 1), (3840, 2160, 8, 1), (1920, 1080, 8, 2), (1920, 1080, 8, 4), (1920, 1080, 8, 16), (1920, 1080, 8, 32)]:
    p = sc.frame_params(width=w, height=h, samples=spp, use_filter=0)
    row = []
    for org, front in ((1, 0), (1, 3), (2, 0), (2, 3), (2, 2)):
        ctx.set_wavefront_organisation(org)
        ctx.set_frame_front(front)
        ms = []
        for i in range(3 + (12 if frames == 1 else 4)):
            if frames == 1:
                ctx.render(p)
            else:
                ctx.render_batch([p] * frames)
            if i >= 3:
                ms.append(ctx.last_frame_ms()[0] / frames)
        row.append(float(np.median(ms)))
    print("%-58s %10.3f %12.3f %14.3f %14.3f %14.3f   %s" % ("%dx%d x %d spp, %d frame(s) per pass (%.1f M paths)" % (w, h, spp, frames, w * h * spp * frames / 1e6), row[0], row[1], row[2], row[3], row[4],
                                              ["rounds", "rounds, one front kernel", "frame kernel", "frame kernel, one front kernel", "frame kernel + its front"][int(np.argmin(row))]))
