#!/usr/bin/env python3
"""Soak of the adaptive tile order: N frames of random shapes, shares, cameras and sample counts (thin frames of the dragon scene: the frame kernel with its front in its own
kernel), each rendered with the adaptive order on (orders made from whatever frame came before) and compared bit for bit with the same frame in screen order."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
a = capi.Context(0); a.update_scene(sc); a.set_frame_chain(0); a.set_adaptive_order(1)
b = capi.Context(0); b.update_scene(sc); b.set_frame_chain(0); b.set_adaptive_order(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
N = int(os.environ.get("N", "600"))
shapes = [(640, 360), (512, 288), (1920, 1080), (960, 544), (333, 222)]
differ = used = 0
t0 = time.time()
shape, share, spp, b_ = shapes[0], None, 2, 3
for f in range(N):
    if rng.random() < 0.15:                                  # now and then another frame shape / share / sample count
        shape = shapes[rng.integers(len(shapes))]
        share = None if rng.random() < 0.4 else (8, int(rng.integers(0, 4)), 4) if shape[1] >= 256 else None
        if shape == (1920, 1080): share = (8, int(rng.integers(0, 8)), 8)
        spp, b_ = int(rng.choice([1, 2, 3, 4])), int(rng.choice([2, 3, 4]))
    p = sc.frame_params(width=shape[0], height=shape[1], samples=spp, max_reflections=b_, use_filter=0)
    p.camera[0] += 0.02 * (f % 50); p.camera[2] -= 0.01 * (f % 30)
    if share: p.tile_rows, p.tile_index, p.tile_count = share
    x = a.render(p)[0]
    used += a.last_organisation() >= 2
    y = b.render(p)[0]
    if not np.array_equal(x.view(np.uint32), y.view(np.uint32)): differ += 1
print("%d frames (%d through the frame kernel) in %.1f s: %d differ from their render in screen order" % (N, used, time.time() - t0, differ))
sys.exit(1 if differ else 0)
