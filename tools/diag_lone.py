#!/usr/bin/env python3
"""Cycles per wave-iteration of the walk kernel when one wave runs alone (tiny frame) vs a full machine (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
for (w, h, spp) in ((1, 1, 1), (2, 2, 1), (8, 8, 1), (64, 64, 1), (1920, 1080, 8)):
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=1, use_filter=0)
    _, cnt, _ = ctx.render(p, counters=True)
    d = ctx.get_diag()
    it = d[0]
    fold, refill, inner, life, waves = d[8:13]
    print("%dx%d x%d: waves %d iterations %d visits %d  cycles per wave-iteration %.0f  lifetime/wave %.0f" % (w, h, spp, waves, it, cnt["closest_visits"] + cnt["shadow_visits"], inner / max(1, it), life / max(1, waves)))
