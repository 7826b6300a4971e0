import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0, width=480, height=272)
ctx.render(p)
ctx.set_frame_lanes(2)
ctx.set_frame_chain(2)
try:
    ctx.frame_begin(p, device=True)
    print("begun, kind", ctx.last_chained(), flush=True)
    ctx.frame_begin(p, device=True)
    print("begun 2", flush=True)
    print(ctx.frame_end(), flush=True)
    print(ctx.frame_end(), flush=True)
except Exception as e:
    print("ERR", e)
print(ctx.server_stats())
