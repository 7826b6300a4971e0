#!/usr/bin/env python3
"""Frame time of the dragon 1080p frame (flx_last_frame_ms: min / median of N synchronous renders) for every library in build/variants (tools/build_variants.sh),
each in its own process (FLX_LIB).  usage: variant_time.py [name ...]   env FLX_TILES=8: a rank's eighth"""
import os, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("FLX_VARIANT_CHILD"):
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
    from flexlight_hip import capi
    from flexlight_hip.scene_io import Scene
    sc = Scene.golden(os.environ.get("FLX_SCENE", "dragon"))
    ctx = capi.Context(0); ctx.update_scene(sc); ctx.set_frame_chain(0)
    if os.environ.get("FLX_ADAPTIVE") and hasattr(capi.LIB, "flx_debug_set_adaptive_order"): ctx.set_adaptive_order(int(os.environ["FLX_ADAPTIVE"]))
    p = sc.frame_params(width=int(os.environ.get("W", "1920")), height=int(os.environ.get("H", "1080")), samples=8, max_reflections=4, use_filter=0)
    if os.environ.get("FLX_TILES"): p.tile_rows, p.tile_count, p.tile_index = 8, int(os.environ["FLX_TILES"]), 0
    for _ in range(4): ctx.render(p)
    ms = []
    for _ in range(int(os.environ.get("N", "30"))):
        ctx.render(p); ms.append(ctx.last_frame_ms())
    print("%-8s%s frame min %.3f median %.3f ms   kernel min %.3f" % (os.environ["FLX_VARIANT_CHILD"], " adaptive " + os.environ["FLX_ADAPTIVE"] if os.environ.get("FLX_ADAPTIVE") else "", min(m[0] for m in ms), float(np.median([m[0] for m in ms])), min(m[1] for m in ms)), flush=True)
    sys.exit(0)
names = sys.argv[1:] or sorted(os.path.basename(f)[4:-3] for f in glob.glob(os.path.join(ROOT, "build/variants/lib_*.so")))
for rep in range(int(os.environ.get("REPS", "1"))):
    for n in names:
        env = dict(os.environ, FLX_VARIANT_CHILD=n, FLX_LIB=os.path.join(ROOT, "build/variants/lib_%s.so" % n))
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, timeout=120)
