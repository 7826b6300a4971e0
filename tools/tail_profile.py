#!/usr/bin/env python3
"""When do the walk workgroups of a frame's walk kernel thin out?  Cycles (since the workgroup's start) at which its walks in flight
first numbered <= 2^k, mean and max over the workgroups (counted build, flx_get_tail_diag; GPU box).  argv: scene [bounces]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(sys.argv[1] if len(sys.argv) > 1 else "dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0)
if len(sys.argv) > 2: p.max_reflections = int(sys.argv[2])
_, cnt, _ = ctx.render(p, counters=True)
t = ctx.get_tail_diag()
d = ctx.get_diag()
r = int(os.environ.get("ROUND", "0")); print("round %d wave lifetimes: mean %.0f max %.0f cycles" % (r, d[16 + 3 * r] / max(1, d[17 + 3 * r]), d[18 + 3 * r]))
print("queue dry:            workgroups %4d  mean %9.0f  max %9.0f" % (t[37], t[36] / max(1, t[37]), t[38]))
for k in range(11, -1, -1):
    if t[3 * k + 1]:
        print("walks in flight <= %4d: workgroups %4d  mean %9.0f  max %9.0f" % (1 << k, t[3 * k + 1], t[3 * k] / t[3 * k + 1], t[3 * k + 2]))
