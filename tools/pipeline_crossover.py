#!/usr/bin/env python3
"""Where does the wavefront pipeline overtake the persistent path kernel?  Synthetic triangle soups of growing size (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from flexlight_hip import capi
import synth_scene
ctx = capi.Context(0)
for objs, tris in ((1, 30), (2, 100), (4, 250), (8, 500), (8, 2000), (16, 4000)):
    sc = synth_scene.make(seed=5, n_objects=objs, tris_per_object=tris, n_transforms=2, n_lights=1, textured=False, width=1920, height=1080, samples=4, bounces=3)
    ctx.update_scene(sc)
    p = sc.frame_params()
    line = "%6d entries:" % sc.meta["textureLength"]
    for pipe in (3, 2):
        ctx.set_pipeline(pipe)
        for _ in range(2): ctx.render(p)
        ms = min(ctx.render(p) and ctx.last_frame_ms()[0] for _ in range(4))
        line += "  pipeline %d %.3f ms" % (pipe, ms)
    print(line)
