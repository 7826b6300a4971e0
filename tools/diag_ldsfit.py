#!/usr/bin/env python3
"""Cycles per wave-iteration of the bounce-0 walk kernel on a scene whose whole threaded tree fits the LDS tree top (no global
entry fetch at all) against the dragon scene (a share of the fetches goes to L2): what the global fetches cost a trip (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
import synth_scene
ctx = capi.Context(0)
for name, sc in (("synthetic, 3 transforms, fits LDS", synth_scene.make(seed=2, n_objects=3, tris_per_object=110, n_transforms=3, n_lights=2)),
                 ("synthetic, 3 transforms, 4x the entries", synth_scene.make(seed=2, n_objects=3, tris_per_object=440, n_transforms=3, n_lights=2)),
                 ("dragon", Scene.golden("dragon"))):
    ctx.update_scene(sc)
    ctx.set_pipeline(3)
    p = sc.frame_params(width=1920, height=1080, samples=8, max_reflections=1, use_filter=0)
    _, cnt, _ = ctx.render(p, counters=True)
    d = ctx.get_diag()
    it = d[0]
    fold, refill, inner, life, waves = d[8:13]
    visits = cnt["closest_visits"] + cnt["shadow_visits"]
    print("%-42s entries %6d: waves %d iterations %d visits %d lane utilisation %.2f  cycles per wave-iteration %.0f  lifetime/wave %.0f" % (
        name, sc.arrays["geometry"].size // 12, waves, it, visits, visits / max(1, it * 64), inner / max(1, it), life / max(1, waves)))
