import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.set_frame_chain(3)
ctx.update_scene(sc)
ctx.set_frame_lanes(2)
def moving(f, **kw):
    p = sc.frame_params(use_filter=0, **kw)
    p.camera[0] += 0.05 * f; p.camera[2] -= 0.03 * f; p.random_seed = float(f % 4)
    return p
a = [moving(f, width=480, height=272) for f in range(3)]
b = [moving(f, width=320, height=200) for f in range(3)]
def step(name, fn):
    t0 = time.time()
    try:
        r = fn()
        print("%-40s ok %.3f s" % (name, time.time() - t0), flush=True)
        return r
    except Exception as e:
        print("%-40s FAILED after %.3f s: %s" % (name, time.time() - t0, e), flush=True)
        print(ctx.server_stats())
        raise SystemExit(1)
want = step("renders", lambda: [ctx.render(p)[0] for p in a + b])
for i, p in enumerate(a + b):
    if ctx.frames_in_flight() == 2:
        step("end", ctx.frame_end)
    step("begin %d" % i, lambda: ctx.frame_begin(p))
while ctx.frames_in_flight():
    step("end", ctx.frame_end)
step("begin a0", lambda: ctx.frame_begin(a[0]))
step("lights", lambda: ctx.update_primary_light_sources(sc.arrays["lights"]))
step("begin a1", lambda: ctx.frame_begin(a[1]))
step("end", ctx.frame_end); step("end", ctx.frame_end)
step("begin a0", lambda: ctx.frame_begin(a[0]))
step("begin a1", lambda: ctx.frame_begin(a[1]))
step("render b2", lambda: ctx.render(b[2]))
step("end", ctx.frame_end); step("end", ctx.frame_end)
step("begin a2", lambda: ctx.frame_begin(a[2]))
step("end", ctx.frame_end)
