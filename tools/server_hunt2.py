import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.set_frame_chain(3)
def moving(f, **kw):
    p = sc.frame_params(use_filter=0, **kw)
    p.camera[0] += 0.05 * f; p.camera[2] -= 0.03 * f; p.random_seed = float(f % 4)
    return p
names = {0: "alive0", 1: "alive1", 2: "alive2", 30: "tiledry0", 31: "tiledry1", 32: "tiledry2", 33: "savail0", 34: "savail1", 35: "savail2", 36: "seq0", 37: "seq1", 38: "seq2", 39: "slotP", 40: "rotlock", 41: "exit", 42: "stopped", 43: "stopafter", 44: "lastwork", 45: "ntiles0", 46: "ntiles1", 47: "ntiles2"}
cases = [(dict(width=640, height=360), 2), (dict(width=640, height=360), 3), (dict(width=1920, height=1080, tile=(8, 5, 8)), 2), (dict(width=1920, height=1080, tile=(8, 5, 8)), 3)]
for shape, lanes in cases:
    try:
        ctx.update_scene(sc)
        ctx.set_frame_lanes(lanes)
        ps = [moving(f, **shape) for f in range(8)]
        want = [ctx.render(p)[0] for p in ps]
        got = []
        for p in ps:
            if ctx.frames_in_flight() == lanes:
                got.append(ctx.frame_end()[0])
            ctx.frame_begin(p)
        while ctx.frames_in_flight():
            got.append(ctx.frame_end()[0])
        print(shape, lanes, "ok", all(np.array_equal(g.view(np.uint32), w.view(np.uint32)) for g, w in zip(got, want)), flush=True)
    except Exception as e:
        print(shape, lanes, "FAILED:", e)
        print(ctx.server_stats())
        for d in ctx.server_dump():
            if d[2:66].sum() == 0:
                continue
            print("workgroup %d wave %d (host mailbox as the device reads it: %d %d):" % (int(d[0]) & 0xffffffff, int(d[1]) & 0xffffffff, int(d[0]) >> 32, int(d[1]) >> 32), {names.get(i, "ring%d" % (i - 3)): int(d[2 + i]) for i in range(48) if d[2 + i] != 0 or i in (36, 37, 38, 39)}, "relay posted", d[66:69].tolist(), "tileNext", d[69:72].tolist())
        break
