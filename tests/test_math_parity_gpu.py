"""include/flx_math.h must give the same bits under gcc/x86-64 and hipcc/gfx950 (it pins the
reference's implementation-defined GLSL built-ins; fragment:119-121's RNG depends on it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FNS = {0: "sin", 1: "cos", 2: "tan", 3: "acos", 4: "atan2", 5: "exp", 6: "pow", 7: "tanh", 8: "floor", 9: "sqrt", 10: "div"}


def _inputs(fn, rng, n):
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-38, -1e-38, 1e-45, 3.4e38, 0.5, 2.0 ** -16,
                        1.0000001, 0.99999994, 1048576.0, 1048577.0], np.float32)
    if fn in (0, 1):
        a = np.concatenate([rng.uniform(-700, 700, n), rng.uniform(-1e6, 1e6, n // 4)])
    elif fn == 2:
        a = rng.uniform(-3.2, 3.2, n)
    elif fn == 3:
        a = rng.uniform(-1.01, 1.01, n)
    elif fn in (4, 10):
        a = rng.normal(0, 10, n)
    elif fn == 5:
        a = rng.uniform(-100, 100, n)
    elif fn == 6:
        a = rng.uniform(0, 8, n)
    elif fn == 7:
        a = rng.uniform(-25, 25, n)
    elif fn == 8:       # floor: wide range, around zero, integers and their neighbours, the 2^23 / 2^24 edges, denormals
        k = rng.integers(-2 ** 24 - 4, 2 ** 24 + 4, n // 8).astype(np.float32)
        a = np.concatenate([rng.uniform(-1e7, 1e7, n // 2), rng.uniform(-2, 2, n // 4), k, np.nextafter(k, np.float32(np.inf)),
                            np.nextafter(k, np.float32(-np.inf)), rng.uniform(-1e-39, 1e-39, 64), rng.uniform(-3e9, 3e9, 64)])
    else:
        a = rng.uniform(0, 1e6, n)
    a = np.concatenate([a.astype(np.float32), special])
    b = None
    if fn in (4, 10):
        b = np.concatenate([rng.normal(0, 10, a.size - special.size).astype(np.float32), special[::-1]])
    if fn == 6:
        b = np.concatenate([rng.uniform(-4, 6, a.size - special.size).astype(np.float32), special[::-1]])
    return a, b


@pytest.mark.parametrize("fn", sorted(FNS))
def test_bit_equal(hip, oracle, fn):
    import ctypes as C
    rng = np.random.default_rng(1234 + fn)
    a, b = _inputs(fn, rng, 200000)
    got = hip.debug_math(fn, a, b)
    want = np.empty_like(a)
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float)) if x is not None else None
    oracle.lib().flx_oracle_math.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32]
    oracle.lib().flx_oracle_math(fn, fp(a), fp(b), fp(want), a.size)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    bad = np.flatnonzero(~same)
    assert bad.size == 0, "%s: %d mismatches, first a=%r b=%r gpu=%r cpu=%r" % (
        FNS[fn], bad.size, a[bad[0]], None if b is None else b[bad[0]], got[bad[0]], want[bad[0]])


def test_gpu_math_against_the_literal_table(hip):
    """the same literal answers (tests/golden/math_kat.json, computed in 60-digit decimal arithmetic) on the GPU"""
    import json
    import os
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "math_kat.json")))
    for name, sel in (("sin", 0), ("cos", 1), ("tan", 2), ("acos", 3), ("exp", 5), ("tanh", 7)):
        rows = np.array(table[name], np.uint32)
        got = hip.debug_math(sel, rows[:, 0].copy().view(np.float32))
        assert np.array_equal(got.view(np.uint32), rows[:, 1]), name
    rows = np.array(table["atan2"], np.uint32)
    got = hip.debug_math(4, rows[:, 0].copy().view(np.float32), rows[:, 1].copy().view(np.float32))
    assert np.array_equal(got.view(np.uint32), rows[:, 2])
