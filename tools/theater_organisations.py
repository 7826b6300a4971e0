#!/usr/bin/env python3
"""The theater (BASELINE configs[4]: 23 entries, 9 lights, 1080p 16 spp 6 bounces) under every kernel organisation the library has: the persistent path kernel
(its default), the per-pixel kernel, and the wavefront pipeline as rounds and as the frame kernel (walk waves + shade waves).  GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("theater")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0)
print("theater %dx%d spp %d bounces %d" % (p.width, p.height, p.samples, p.max_reflections))
ref = None
for pipe, org, label in ((2, 0, "persistent path kernel (k_paths)"), (3, 1, "wavefront pipeline, rounds"), (3, 2, "wavefront pipeline, frame kernel"), (1, 0, "per-pixel kernel")):
    ctx.set_pipeline(pipe)
    ctx.set_wavefront_organisation(org)
    try:
        img = ctx.render(p)[0]
        for _ in range(2): ctx.render(p)
        ms = []
        for _ in range(5):
            ctx.render(p)
            ms.append(ctx.last_frame_ms()[0])
        same = "" if ref is None else ("  equal" if np.array_equal(img.view(np.uint32), ref.view(np.uint32)) else "  DIFFERS")
        if ref is None: ref = img
        print("%-40s pipeline %d organisation %d (ran %d): %.3f ms per frame%s" % (label, ctx.last_pipeline(), org, ctx.last_organisation(), min(ms), same), flush=True)
    except capi.FlexLightHipError as e:
        print("%-40s %s" % (label, e))
