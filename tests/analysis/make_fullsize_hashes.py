#!/usr/bin/env python3
"""sha256 of the CPU oracle's frames of the five BASELINE.json configurations AT FULL SIZE, with their work counters
(tests/golden/oracle_fullsize.json; SURVEY.md 8c item 5: "whole-frame oracle images at reduced size and hashes at full size").

Oracle and kernels are written by one hand and share include/flx_math.h: bit equality between them (which every GPU test asserts)
would survive a change that moved BOTH.  These hashes are the fixed point neither may leave: tests/test_oracle_kat.py recomputes the
oracle's full-size frames on the CPU and requires them, tests/test_parity_gpu.py requires them of the GPU frames at the sizes
bench.py times.  Re-generate only when the definition of a frame changes on purpose, and say so in DESIGN.md.

    python tests/analysis/make_fullsize_hashes.py            # ~1 minute on 8 cores"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))

# key: (scene fixture, width, height, spp, bounces, filter)   — BASELINE.json configs[0..4]
CONFIGS = {
    "configs[0] cornell 256x256 1spp 1b": ("cornell", 256, 256, 1, 1, 0),
    "configs[1] cornell_obj 1080p 4spp 3b filter": ("cornell_obj", 1920, 1080, 4, 3, 1),
    "configs[2] dragon 1080p 8spp 4b": ("dragon", 1920, 1080, 8, 4, 0),
    "configs[3] dragon 4K 8spp 4b": ("dragon", 3840, 2160, 8, 4, 0),
    "configs[4] theater 1080p 16spp 6b": ("theater", 1920, 1080, 16, 6, 0),
}


CORES = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)      # asked for by number: a test process's
# OpenMP default may have been turned down to one thread by whatever ran before (torch.distributed does)


def sha(a):
    """sha256 of a float32 array's bytes, every NaN first set to the one canonical quiet NaN (a NaN's payload is not part of a frame)"""
    a = np.array(a, np.float32, copy=True)
    a[np.isnan(a)] = np.float32(np.nan)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def oracle_record(key):
    import flx_oracle
    from flexlight_hip.scene_io import Scene
    name, w, h, spp, b, filt = CONFIGS[key]
    sc = Scene.golden(name)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=b, use_filter=filt)
    if filt:
        img, cnt, gb = flx_oracle.render(sc, p, gbuffers=True, threads=CORES)
        return {"frame": sha(img), "counters": cnt, "gbuffers": {k: sha(v) for k, v in sorted(gb.items())}}
    img, cnt, _ = flx_oracle.render(sc, p, threads=CORES)
    return {"frame": sha(img), "counters": cnt}


def main():
    out = {}
    for key in CONFIGS:
        t0 = time.time()
        out[key] = oracle_record(key)
        print("%-46s %s  %.1f s" % (key, out[key]["frame"][:16], time.time() - t0), flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
