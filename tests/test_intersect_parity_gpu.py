"""The GPU's intersection routines, as its kernels call them (flx_debug_intersect), against literal answers computed from the shader text one float32
operation at a time (tests/golden/intersect_kat.json.gz, tests/analysis/make_intersect_kat.py: SURVEY.md 8a I1 - I3).  The walk kernels do not run the
shader's text: 1 / det comes from v_rcp_f32 with a correction step, the acceptance rule is branch-free, the box test is an interval test over
reciprocal products with exact quotients (Markstein) and IEEE division as fallbacks — this is where "the same bits as the shader" is held directly."""
import gzip
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    return json.load(gzip.open(os.path.join(HERE, "golden", "intersect_kat.json.gz"), "rt"))


def _f32(rows, a, b):
    return np.array([r[a:b] for r in rows], np.uint32).view(np.float32)


@pytest.mark.parametrize("fn", [0, 3], ids=["walk_kernels", "per_pixel_kernel"])
def test_moeller_trumbore_literal(hip, kat, fn):
    rows = kat["moeller_trumbore"]
    got = hip.debug_intersect(fn, _f32(rows, 0, 16))
    want = _f32(rows, 16, 19)
    assert np.count_nonzero(want[:, 0]) >= 200
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "rows %s" % np.flatnonzero(~same.all(axis=1))[:10]


@pytest.mark.parametrize("fn", [1, 4], ids=["walk_kernels", "per_pixel_kernel"])
def test_moeller_trumbore_cull_literal(hip, kat, fn):
    rows = kat["moeller_trumbore_cull"]
    got = hip.debug_intersect(fn, _f32(rows, 0, 16))
    want = np.array([r[16] for r in rows], np.float32)
    assert want.sum() >= 100
    assert np.array_equal(got, want), "rows %s" % np.flatnonzero(got != want)[:10]


@pytest.mark.parametrize("fn", [2, 5], ids=["walk_kernels", "per_pixel_kernel"])
def test_ray_cuboid_literal(hip, kat, fn):
    rows = kat["ray_cuboid"]
    got = hip.debug_intersect(fn, _f32(rows, 0, 13))
    want = np.array([r[13] for r in rows], np.float32)
    assert 300 <= want.sum() <= len(rows) - 300
    assert np.array_equal(got, want), "rows %s" % np.flatnonzero(got != want)[:10]
