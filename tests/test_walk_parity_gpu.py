"""The GPU's walks, as its kernels run them (flx_debug_walk), against literal answers computed from the shader text (tests/golden/walk_kat.json.gz,
tests/analysis/make_walk_kat.py: rayTracer and shadowTest, fragment:172-279, transcribed statement by statement over the reference's own arrays — SURVEY.md 8a T1, T2).
None of the device walks runs the shader's loop as written: the wavefront pipeline walks a threaded, hot-first copy of the skip list with explicit successors,
stored edges, rays pre-transformed into every object space and the reciprocal box test; small scenes are walked by the wave in lockstep.  Here each of them has to
find the same hit to the last bit, the same triangle and the same shadow answer — and fetch the same number of entries — as the text."""
import gzip
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = {0: "wavefront lane walk (threaded copy)", 1: "per-pixel / persistent lane walk", 2: "lockstep walk"}


@pytest.mark.parametrize("name", ["cornell", "cornell_obj", "theater", "dragon"])
def test_walks_literal(hip, scenes, name):
    from flexlight_hip import capi
    rows = json.load(gzip.open(os.path.join(HERE, "golden", "walk_kat.json.gz"), "rt"))[name]
    rays = np.array([r[0:7] for r in rows], np.uint32).view(np.float32)
    want_suv = np.array([r[7:10] for r in rows], np.uint32)
    want = np.array([r[10:15] for r in rows], np.int64)           # 2 x transform, entry, entries fetched, shadowed, entries fetched
    hip.update_scene(scenes(name))
    ran = []
    for variant in VARIANTS:
        try:
            got = hip.debug_walk(variant, rays)
        except capi.FlexLightHipError as e:
            assert variant == 2 and "lockstep" in str(e), (variant, e)      # only the lockstep walk may be missing (large or multi-space scenes)
            continue
        ran.append(variant)
        hit = want[:, 1] != -1
        suv = got[:, 0:3].view(np.uint32)
        same = (suv == want_suv) | (np.isnan(got[:, 0:3]) & np.isnan(want_suv.view(np.float32)))
        assert same.all(), (VARIANTS[variant], np.flatnonzero(~same.all(axis=1))[:10])
        assert np.array_equal(got[:, 4].astype(np.int64), want[:, 1]), VARIANTS[variant]
        assert np.array_equal(got[hit, 3].astype(np.int64), want[hit, 0]), VARIANTS[variant]
        assert np.array_equal(got[:, 5].astype(np.int64), want[:, 2]), VARIANTS[variant]
        assert np.array_equal(got[:, 6].astype(np.int64), want[:, 3]), VARIANTS[variant]
        assert np.array_equal(got[:, 7].astype(np.int64), want[:, 4]), VARIANTS[variant]
    assert 0 in ran and 1 in ran
    if name == "cornell":
        assert 2 in ran
