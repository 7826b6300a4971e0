#!/usr/bin/env python3
"""Soak test of the frame kernel's hand-over (walk waves -> LDS-counted rings -> shade waves, workgroup-scope release / acquire): the same frame
rendered many times must hash to the same value every time — and to the committed oracle hash where there is one.  GPU box.
usage: soak_framekernel.py [frames]"""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests", "analysis"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
import make_fullsize_hashes as fs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")))
ctx = capi.Context(0)
for name, kw, key, reps in [("dragon", {}, "configs[2] dragon 1080p 8spp 4b", n), ("dragon", dict(width=3840, height=2160), "configs[3] dragon 4K 8spp 4b", max(4, n // 20)),
                            ("theater", dict(width=960, height=540, samples=8, max_reflections=6), None, max(4, n // 10)), ("dragon", dict(width=333, height=187, samples=3, max_reflections=6), None, n)]:
    sc = Scene.golden(name)
    ctx.update_scene(sc)
    ctx.set_pipeline(3); ctx.set_wavefront_organisation(2)
    p = sc.frame_params(use_filter=0, **kw)
    seen = {}
    for i in range(reps):
        ctx.set_frame_front(0 if i % 4 == 3 else 2)        # three frames with the front of the frame inside the launch (the third ring), one with the kernels in front
        h = fs.sha(ctx.render(p)[0])
        seen[h] = seen.get(h, 0) + 1
    ok = len(seen) == 1 and (key is None or list(seen)[0] == want[key]["frame"])
    print("%-8s %-48s %4d frames, %d distinct hash(es)%s  %s" % (name, kw or "1080p 8 spp 4 bounces", reps, len(seen), "" if key is None else ", oracle hash " + ("matches" if list(seen)[0] == want[key]["frame"] else "DIFFERS"), "OK" if ok else "FAIL"), flush=True)
    if not ok:
        sys.exit(1)
