#!/usr/bin/env python3
"""Random frames (odd sizes, 1 - 19 samples, 1 - 8 bounces, whole frames and strips) of three scenes rendered as rounds with two kernels in front and as every other
organisation / front combination: frames and counters must be identical.  GPU box; a longer version of tests/test_parity_gpu.py::test_random_frames_agree_between_the_organisations."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
bad = 0
for name in ("dragon", "theater", "cornell_obj"):
    sc = Scene.golden(name)
    hip = capi.Context(0)
    hip.update_scene(sc)
    hip.set_pipeline(3)
    for seed in range(12):
        rng = np.random.default_rng(1000 + seed)
        for case in range(60):
            w, h = int(rng.integers(1, 900)), int(rng.integers(1, 500))
            spp, bounces = int(rng.integers(1, 20)), int(rng.integers(1, 9))
            tile = (0, 0, 0)
            if rng.random() < 0.5:
                count = int(rng.integers(2, 9))
                tile = (int(rng.choice([1, 3, 8, 16])), int(rng.integers(0, count)), count)
            p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0, tile=tile)
            p.random_seed = float(case)
            hip.set_wavefront_organisation(1); hip.set_frame_front(0)
            want, want_cnt, _ = hip.render(p, counters=True)
            for organisation, front in ((2, 0), (2, 2), (2, 3), (1, 3), (0, 1)):
                hip.set_wavefront_organisation(organisation); hip.set_frame_front(front)
                got, cnt, _ = hip.render(p, counters=True)
                got2 = hip.render(p)[0]
                if not (np.array_equal(got, want, equal_nan=True) and cnt == want_cnt and np.array_equal(got2, want, equal_nan=True)):
                    bad += 1
                    print("MISMATCH", name, seed, case, w, h, spp, bounces, tile, organisation, front, flush=True)
    print(name, "done", flush=True)
print("mismatches:", bad)
