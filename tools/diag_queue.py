#!/usr/bin/env python3
"""Scheduler statistics of the queue walk kernel for bounce 0 of one frame (GPU box; FLX_WALK_SCHEDULER=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = Scene.golden(name)
ctx = capi.Context(0)
ctx.update_scene(sc)
ctx.set_walk_scheduler(1)
p = sc.frame_params(max_reflections=1, use_filter=0)
_, cnt, _ = ctx.render(p, counters=True)
d = ctx.get_diag()
names = ["box", "tri", "xform", "new"]
life, waves = d[14], max(1, d[15])
print("waves %d, mean lifetime %.0f cycles" % (waves, life / waves))
for q in range(4):
    ops, lanes, cyc = d[q], d[4 + q], d[8 + q]
    print("  %-5s batches %9d  walks %11d  mean batch %5.1f  cycles/batch %7.0f  share of lifetime %.3f" % (names[q], ops, lanes, lanes / max(1, ops), cyc / max(1, ops), cyc / max(1, life)))
print("  idle trips %d (%.3f of lifetime), claim+pop share %.3f" % (d[12], d[13] / max(1, life), d[16] / max(1, life)))
print("  visits %d" % (cnt["closest_visits"] + cnt["shadow_visits"]))
