"""Known-answer tests that pin the CPU oracle (oracle/): hand-computable cases for the intersection
routines, the RNG and the BRDF, behaviour at NaN / zero-direction edge cases the path relies on
(SURVEY.md §7 R2), committed whole-frame fixtures, and include/flx_math.h against libm.  CPU only."""
import ctypes as C
import glob
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIAS = 2.0 ** -16
F3 = C.c_float * 3
F9 = C.c_float * 9


@pytest.fixture(scope="module")
def lib(oracle):
    l = oracle.lib()
    l.flx_oracle_moeller_trumbore.argtypes = [F9, F3, F3, C.c_float, F3]
    l.flx_oracle_moeller_trumbore_cull.argtypes = [F9, F3, F3, C.c_float]
    l.flx_oracle_moeller_trumbore_cull.restype = C.c_int
    l.flx_oracle_ray_cuboid.argtypes = [C.c_float, F3, F3, F3, F3]
    l.flx_oracle_ray_cuboid.restype = C.c_int
    l.flx_oracle_noise.argtypes = [C.c_float] * 4 + [C.c_float * 4]
    l.flx_oracle_forward_trace.argtypes = [F9, F3, C.c_float, F3, F3, F3]
    l.flx_oracle_math.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32]
    return l


UNIT_TRI = F9(0, 0, 0, 1, 0, 0, 0, 1, 0)          # a, b, c in the z = 0 plane, normal (e1 x e2) = +z


def mt(lib, origin, direction, l=1e9, tri=UNIT_TRI):
    out = F3()
    lib.flx_oracle_moeller_trumbore(tri, F3(*origin), F3(*direction), l, out)
    return tuple(out)


def test_moeller_trumbore_known_hit(lib):
    s, u, v = mt(lib, (0.25, 0.25, 1.0), (0, 0, -1))
    assert (s, u, v) == (1.0, 0.25, 0.25)
    # two-sided: from below as well (fragment:128 tests abs(det))
    assert mt(lib, (0.25, 0.5, -2.0), (0, 0, 1)) == (2.0, 0.25, 0.5)


def test_moeller_trumbore_rejections(lib):
    assert mt(lib, (0.25, 0.25, 1.0), (1, 0, 0)) == (0, 0, 0)              # parallel: |det| < BIAS
    assert mt(lib, (2.0, 0.25, 1.0), (0, 0, -1)) == (0, 0, 0)             # u > 1
    assert mt(lib, (0.75, 0.75, 1.0), (0, 0, -1)) == (0, 0, 0)            # u + v > 1
    assert mt(lib, (0.25, 0.25, 1.0), (0, 0, -1), l=0.5) == (0, 0, 0)     # s > l
    assert mt(lib, (0.25, 0.25, 1.0), (0, 0, -1), l=1.0)[0] == 1.0        # s == l is kept (ties: later entry wins)
    assert mt(lib, (0.25, 0.25, -1.0), (0, 0, -1)) == (0, 0, 0)           # behind: s <= BIAS
    # BIAS-wide cracks along the edges (fragment:132,136): u = BIAS/2 misses, u = 2 BIAS hits
    assert mt(lib, (BIAS / 2, 0.25, 1.0), (0, 0, -1)) == (0, 0, 0)
    assert mt(lib, (2 * BIAS, 0.25, 1.0), (0, 0, -1))[0] == 1.0


def test_moeller_trumbore_nan_direction_misses(lib):
    nan = float("nan")
    # NaN never satisfies the rejections of fragment:128-138 -> the shader returns vec3(NaN...) as a hit;
    # that is the reference's behaviour and what rayTracer's `intersection.x != 0.0` then accepts.
    s, u, v = mt(lib, (0.25, 0.25, 1.0), (nan, nan, nan))
    assert math.isnan(s)
    # the culling variant ends on (s <= l && s > BIAS), false for NaN
    assert lib.flx_oracle_moeller_trumbore_cull(UNIT_TRI, F3(0.25, 0.25, 1.0), F3(nan, nan, nan), 1e9) == 0


def test_moeller_trumbore_cull_is_one_sided(lib):
    hit = lambda o, d, l=1e9: lib.flx_oracle_moeller_trumbore_cull(UNIT_TRI, F3(*o), F3(*d), l)
    # det = e1 . (d x e2) = -d . (e1 x e2): positive only when the ray runs against e1 x e2 = +z
    assert hit((0.25, 0.25, 1.0), (0, 0, -1)) == 1
    assert hit((0.25, 0.25, -1.0), (0, 0, 1)) == 0        # det = -1 < BIAS: culled (fragment:149)
    assert hit((0.25, 0.25, 1.0), (0, 0, -1), l=0.5) == 0
    assert hit((0.25, 0.25, 1.0), (0, 0, -1), l=1.0) == 1


def test_ray_cuboid(lib):
    box = lambda l, o, d: lib.flx_oracle_ray_cuboid(l, F3(*o), F3(*d), F3(-1, -1, -1), F3(1, 1, 1))
    assert box(1e9, (0, 0, -5), (0, 0, 1)) == 1
    assert box(3.9, (0, 0, -5), (0, 0, 1)) == 0           # tmin = 4 is not < l
    assert box(4.1, (0, 0, -5), (0, 0, 1)) == 1
    assert box(1e9, (0, 0, 5), (0, 0, 1)) == 0            # behind: tmax < BIAS
    assert box(1e9, (0, 0, 0), (0, 0, 1)) == 1            # inside
    assert box(1e9, (3, 0, -5), (0, 0, 1)) == 0           # zero direction components: (+-inf) slabs miss in x
    # origin exactly on the slab plane x = 1 with d.x = 0: v0.x = -inf, v1.x = 0/0 = NaN; GLSL min/max return their
    # first argument when the second is NaN, so both the near and the far x-slab are -inf and the box is missed
    assert box(1e9, (1, 0, -5), (0, 0, 1)) == 0
    nan = float("nan")
    assert box(1e9, (0, 0, -5), (nan, nan, nan)) == 0     # NaN direction skips every AABB


def test_noise_is_the_pinned_formula(lib):
    out = (C.c_float * 4)()
    lib.flx_oracle_noise(0.3, -0.2, 1.5, 0.0, out)
    f = np.float32
    d = f(f(0.3) * f(12.9898)) + f(f(-0.2) * f(78.233))
    for k, expect in zip((53.0, 59.0, 61.0, 67.0), out):
        arg = f(d + f(f(k) * f(1.5)))
        sv = np.zeros(1, np.float32)
        lib.flx_oracle_math(0, np.array([arg], np.float32).ctypes.data_as(C.POINTER(C.c_float)), None,
                            sv.ctypes.data_as(C.POINTER(C.c_float)), 1)
        x = f(sv[0] * f(43758.5453))
        want = f(f(x - np.floor(x)) * f(2.0)) - f(1.0)
        assert f(expect) == want
        assert -1.0 <= expect <= 1.0
    # randomSeed enters as seed + randomSeed * PHI (fragment:120)
    a, b = (C.c_float * 4)(), (C.c_float * 4)()
    lib.flx_oracle_noise(0.3, -0.2, 1.5, 2.0, a)
    lib.flx_oracle_noise(0.3, -0.2, float(np.float32(1.5) + np.float32(2.0) * np.float32(1.61803398874989484820459)), 0.0, b)
    assert tuple(a) == tuple(b)


def test_math_literal_known_answers(lib):
    """the pinned sin / cos / tan / acos / atan / exp / tanh and the shader's noise() against a committed
    table of LITERAL answers (tests/golden/math_kat.json: bit patterns computed by tests/analysis/make_math_kat.py in 60-digit
    decimal arithmetic and rounded once — not by the oracle, not through include/flx_math.h).  A change to flx_math.h has to
    keep every one of them."""
    import json
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "math_kat.json")))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    for name, sel in (("sin", 0), ("cos", 1), ("tan", 2), ("acos", 3), ("exp", 5), ("tanh", 7)):
        rows = np.array(table[name], np.uint32)
        assert rows.shape[0] >= 100
        x = rows[:, 0].copy().view(np.float32)
        got = np.empty_like(x)
        lib.flx_oracle_math(sel, fp(x), None, fp(got), x.size)
        bad = np.nonzero(got.view(np.uint32) != rows[:, 1])[0]
        assert bad.size == 0, "%s(%r) = %r, literal answer %r" % (name, x[bad[0]], got[bad[0]], rows[bad[0], 1:2].copy().view(np.float32)[0])
    rows = np.array(table["atan2"], np.uint32)
    y, x = rows[:, 0].copy().view(np.float32), rows[:, 1].copy().view(np.float32)
    got = np.empty_like(x)
    lib.flx_oracle_math(4, fp(y), fp(x), fp(got), x.size)
    assert np.array_equal(got.view(np.uint32), rows[:, 2])
    rows = np.array(table["noise"], np.uint32)
    assert rows.shape == (64, 8)
    out = (C.c_float * 4)()
    for r in rows:
        nx, ny, seed, rs = [float(v) for v in r[:4].copy().view(np.float32)]
        lib.flx_oracle_noise(nx, ny, seed, rs, out)
        assert np.array_equal(np.array(list(out), np.float32).view(np.uint32), r[4:]), (nx, ny, seed, rs)


def test_forward_trace_lambert_limit(lib):
    """Fully rough dielectric lit head-on: Cook-Torrance by hand (fragment:304-334)."""
    mat = F9(0.8, 0.6, 0.4, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0)
    out = F3()
    lib.flx_oracle_forward_trace(mat, F3(0, 0, 3), 16.0, F3(0, 0, 1), F3(0, 0, 1), out)
    brightness = 16.0 / (1 + 3) ** 2
    # N = V = L = H: VdotH = NdotL = NdotH = NdotV = 1, alpha = 1, fresnel(F0, 1) = F0 = albedo
    D = 1.0 / math.pi                    # alpha^2 / (pi * (1*(1-1)+1)^2)
    G = 1.0                              # schlickBeckmann(1, 1) = 1 / (0.5 + 0.5)
    for c, albedo in zip(out, (0.8, 0.6, 0.4)):
        Ks = albedo
        want = ((1 - Ks) * albedo / math.pi + Ks * D * G / 4.0) * brightness
        assert c == pytest.approx(want, rel=2e-6)
    # light behind the surface contributes nothing (NdotL clamps to 0)
    lib.flx_oracle_forward_trace(mat, F3(1, 0, -3), 16.0, F3(0, 0, 1), F3(0, 0, 1), out)
    assert tuple(out) == (0.0, 0.0, 0.0)
    # exactly opposite light and view: H = normalize(0) = NaN, as in the shader (fragment:310)
    lib.flx_oracle_forward_trace(mat, F3(0, 0, -3), 16.0, F3(0, 0, 1), F3(0, 0, 1), out)
    assert all(math.isnan(c) for c in out)


def test_flx_math_against_libm(lib):
    rng = np.random.default_rng(7)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    cases = {0: (rng.uniform(-600, 600, 50000), np.sin), 1: (rng.uniform(-600, 600, 50000), np.cos),
             3: (rng.uniform(-1, 1, 50000), np.arccos), 5: (rng.uniform(-30, 30, 50000), np.exp), 7: (rng.uniform(-6, 6, 50000), np.tanh)}
    for fn, (x, ref) in cases.items():
        x = x.astype(np.float32)
        got = np.empty_like(x)
        lib.flx_oracle_math(fn, fp(x), None, fp(got), x.size)
        want = ref(x.astype(np.float64))
        ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(got - want) / ulp) <= 0.5000001, fn       # correctly rounded (to double-libm accuracy)
    specials = np.array([np.inf, -np.inf, np.nan, 2.0 ** 21], np.float32)
    got = np.empty_like(specials)
    lib.flx_oracle_math(0, fp(specials), None, fp(got), specials.size)
    assert np.isnan(got).all()           # sin is pinned to NaN outside |x| <= 2^20
    one = np.array([1.0000001, 1.0, -1.0, -1.5], np.float32)
    got = np.empty_like(one)
    lib.flx_oracle_math(3, fp(one), None, fp(got), one.size)
    assert list(got) == [0.0, 0.0, np.float32(math.pi), np.float32(math.pi)]   # acos clamps (fragment:516 can exceed 1 by rounding)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "oracle_*.npz"))))
def test_oracle_reproduces_committed_frames(oracle, scenes, path):
    name, size, s, b, f = os.path.basename(path)[len("oracle_"):-len(".npz")].rsplit("_", 4)
    w, h = map(int, size.split("x"))
    fix = np.load(path)
    sc = scenes(name)
    p = sc.frame_params(width=w, height=h, samples=int(s[1:]), max_reflections=int(b[1:]), use_filter=int(f[1:]))
    img, cnt, _ = oracle.render(sc, p)
    assert np.array_equal(img, fix["frame"], equal_nan=True)
    assert [cnt[k] for k in sorted(cnt)] == list(fix["counters"])
    single, cnt1, _ = oracle.render(sc, p, threads=1)           # OpenMP row split must not matter
    assert np.array_equal(single, img, equal_nan=True) and cnt1 == cnt


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "oracle_*.npz"))))
def test_as_written_bounce_loop_gives_the_same_frames(oracle, scenes, path):
    """The oracle (and the kernels) skip the rayTracer call of fragment:591 when the loop guard of :475 is about to discard
    its hit.  That this changes no output is held here as a test, not as an argument: with the bounce loop as the shader is
    written — every iteration ends with that walk — the five committed frames come out bit for bit the same, G-buffers
    included; only the closest-hit work grows, to one walk per bounce iteration."""
    name, size, s, b, f = os.path.basename(path)[len("oracle_"):-len(".npz")].rsplit("_", 4)
    w, h = map(int, size.split("x"))
    fix = np.load(path)
    sc = scenes(name)
    p = sc.frame_params(width=w, height=h, samples=int(s[1:]), max_reflections=int(b[1:]), use_filter=int(f[1:]))
    pruned, cnt, gb = oracle.render(sc, p, gbuffers=bool(p.use_filter))
    oracle.set_as_written(True)
    try:
        full, cnt_full, gb_full = oracle.render(sc, p, gbuffers=bool(p.use_filter))
    finally:
        oracle.set_as_written(False)
    assert np.array_equal(full, fix["frame"], equal_nan=True) and np.array_equal(full, pruned, equal_nan=True)
    if gb:
        for key in gb:
            assert np.array_equal(gb[key], gb_full[key], equal_nan=True), key
    assert cnt_full["closest_walks"] == cnt_full["shades"]            # the shader traces once per bounce iteration
    assert cnt_full["closest_walks"] >= cnt["closest_walks"] and cnt_full["closest_visits"] >= cnt["closest_visits"]
    for key in cnt:
        if not key.startswith("closest_"):
            assert cnt_full[key] == cnt[key], key
    if int(b[1:]) > 0 and name != "cornell_obj":
        assert cnt_full["closest_walks"] > cnt["closest_walks"]


def test_oracle_primary_ray_geometry(oracle, scenes):
    """Centre pixel of the cornell frame looks down +z from (0,0,-20) and hits the back wall at z = 5."""
    l = oracle.lib()
    from flexlight_hip.scene_io import FrameParams, SceneView
    l.flx_oracle_primary.argtypes = [C.POINTER(SceneView), C.POINTER(FrameParams), C.c_uint32, C.c_uint32, F3, C.POINTER(C.c_int), C.POINTER(C.c_int), F3]
    sc = scenes("cornell")
    p = sc.frame_params(width=255, height=255)
    view = sc.view()
    suv, d = F3(), F3()
    tid, tri = C.c_int(), C.c_int()
    l.flx_oracle_primary(C.byref(view), C.byref(p), 127, 127, suv, C.byref(tid), C.byref(tri), d)
    assert tuple(d) == (0.0, 0.0, 1.0)
    assert suv[0] == 25.0 and tid.value == 0
    g = sc.arrays["geometry"].reshape(-1, 12)[tri.value]
    assert g[10] == 2 and g[2] == 5 and g[5] == 5 and g[8] == 5      # a triangle of the back plane (z = 5)


def test_empty_and_degenerate_inputs(oracle, scenes):
    sc = scenes("cornell")
    p = sc.frame_params(width=8, height=8, samples=1, max_reflections=0, use_filter=0)
    img, cnt, _ = oracle.render(sc, p)                      # zero bounces: ambient only, times originalColor (1,1,1)
    hit = img[..., 3] == 1
    assert cnt["shades"] == 0 and cnt["closest_walks"] == 0
    assert np.allclose(img[hit][:, :3], np.array(sc.meta["ambient"], np.float32))
    bad = sc.frame_params(width=8, height=8)
    bad.samples = 0
    with pytest.raises(RuntimeError):
        oracle.render(sc, bad)


def test_fxaa_and_taa_known_answers(oracle):
    """N4 oracle pins: a flat frame passes through FXAA unchanged (low contrast everywhere away from the zero border); TAA of
    nine equal frames returns the frame in the interior, and so does TAA of one frame: the eight zero textures are clamped
    to the 3x3 neighbourhood's [min, max] before they are averaged in."""
    flat = np.full((16, 20, 4), 0.5, np.float32)
    q = np.float32(128) / np.float32(255)                   # 0.5 stored as RGBA8 and read back
    out = oracle.fxaa(flat)
    assert np.all(out[2:-2, 2:-2] == q)
    t9 = oracle.taa([flat] * 9)
    assert np.allclose(t9[1:-1, 1:-1], q, atol=1e-6)
    t1 = oracle.taa([flat])
    assert np.allclose(t1[1:-1, 1:-1], q, atol=1e-6)
    assert np.allclose(t1[0, 0], q / 9, atol=1e-6)            # at the border the neighbourhood's minimum is the zero outside


def test_unorm8_recipe_is_exact(tmp_path):
    """the filters' division-free k / 255 and their byte comparisons (tools/unorm_check.c) hold for all 256 bytes"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "unorm_check"
    subprocess.check_call(["gcc", "-O0", "-ffp-contract=off", "-o", str(exe), os.path.join(root, "tools", "unorm_check.c"), "-lm"])
    out = subprocess.check_output([str(exe)]).decode()
    assert "one Markstein step mismatches 0" in out and "!= k: 0" in out and "> 0.1f: 26" in out and out.strip().endswith(": 0")


def test_present_kat():
    """floor(clamp(x) * 255 + 0.5): known answers"""
    import flx_oracle
    f = np.zeros((1, 8, 4), np.float32)
    f[0, :, 0] = [0.0, 1.0, 0.5, -3.0, 7.0, np.nan, 1.0 / 255.0, 0.999]
    got = flx_oracle.present(f)[0, :, 0].tolist()
    assert got[:5] == [0, 255, 128, 0, 255] and got[6] == 1 and got[7] == 255


def test_intersection_literal_known_answers(lib):
    """tests/golden/intersect_kat.json.gz: moellerTrumbore, moellerTrumboreCull and rayCuboid evaluated from the shader text one float32 operation at a
    time (tests/analysis/make_intersect_kat.py: not through oracle/ nor include/flx_math.h) on 1 500 random and edge cases each — the oracle's routines
    give the same bits"""
    import gzip
    import json
    import struct
    data = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "intersect_kat.json.gz"), "rt"))
    fl = lambda words: [struct.unpack("<f", struct.pack("<I", w))[0] for w in words]
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    assert len(data["moeller_trumbore"]) >= 1000 and sum(1 for r in data["moeller_trumbore"] if r[16:] != [0, 0, 0]) >= 200
    for r in data["moeller_trumbore"]:
        out = F3()
        lib.flx_oracle_moeller_trumbore(F9(*fl(r[0:9])), F3(*fl(r[9:12])), F3(*fl(r[12:15])), fl(r[15:16])[0], out)
        assert [bits(x) for x in out] == r[16:19], r
    for r in data["moeller_trumbore_cull"]:
        assert lib.flx_oracle_moeller_trumbore_cull(F9(*fl(r[0:9])), F3(*fl(r[9:12])), F3(*fl(r[12:15])), fl(r[15:16])[0]) == r[16], r
    for r in data["ray_cuboid"]:
        assert lib.flx_oracle_ray_cuboid(fl(r[0:1])[0], F3(*fl(r[1:4])), F3(*fl(r[4:7])), F3(*fl(r[7:10])), F3(*fl(r[10:13]))) == r[13], r


@pytest.mark.parametrize("name", ["cornell", "cornell_obj", "theater", "dragon"])
def test_walk_literal_known_answers(oracle, scenes, name):
    """tests/golden/walk_kat.json.gz: rayTracer and shadowTest (fragment:172-279) transcribed statement by statement over the arrays the reference's scene.js emits
    (tests/analysis/make_walk_kat.py, with make_intersect_kat.py's float32 intersection routines: not through oracle/) — camera rays, rays from the surface points they
    find, zero direction components, the dragon's 73 694 entries: the oracle's walks find the same hit to the last bit, the same triangle, the same shadow answer and
    fetch the same number of entries"""
    import gzip
    import json
    import struct
    rows = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "walk_kat.json.gz"), "rt"))[name]
    fl = lambda words: [struct.unpack("<f", struct.pack("<I", w))[0] for w in words]
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    sc = scenes(name)
    view = sc.view()
    L = oracle.lib()
    L.flx_oracle_ray_tracer.argtypes = [C.c_void_p, F3, F3, F3, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.flx_oracle_ray_tracer.restype = None
    L.flx_oracle_shadow_test.argtypes = [C.c_void_p, F3, F3, C.c_float, C.POINTER(C.c_uint64)]
    L.flx_oracle_shadow_test.restype = C.c_int
    assert len(rows) >= 100 and sum(1 for r in rows if r[11] != -1) >= 60
    for r in rows:
        origin, d, l = F3(*fl(r[0:3])), F3(*fl(r[3:6])), fl(r[6:7])[0]
        suv, ti, tri, visits = F3(), C.c_int(), C.c_int(), C.c_uint64(0)
        L.flx_oracle_ray_tracer(C.byref(view), origin, d, suv, C.byref(ti), C.byref(tri), C.byref(visits))
        assert ([bits(x) for x in suv], ti.value, tri.value, visits.value) == (r[7:10], r[10], r[11], r[12]), r
        visits = C.c_uint64(0)
        assert (L.flx_oracle_shadow_test(C.byref(view), origin, d, l, C.byref(visits)), visits.value) == (r[13], r[14]), r


def _filter_kat_cases():
    import gzip
    import json
    return json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "filter_kat.json.gz"), "rt"))


def filter_kat_inputs(case):
    """-> (FrameParams, five float planes [H, W, 4] whose RGBA8 store is the case's bytes, expected float frame [H, W, 4])"""
    from flexlight_hip.capi import FrameParams
    W, H = case["width"], case["height"]
    p = FrameParams()
    p.width, p.height, p.samples, p.max_reflections, p.use_filter, p.hdr, p.texture_width = W, H, 1, 1, 1, case["hdr"], 1
    planes = [(np.array(pl, np.uint8).reshape(H, W, 4).astype(np.float32) / np.float32(255.0)).astype(np.float32) for pl in case["planes"]]
    want = np.array([v for row in case["out"] for v in row], np.uint32).view(np.float32).reshape(H, W, 4)
    return p, planes, want


def assert_filter_kat(got, want, hdr, what):
    if hdr == 0:
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "%s: %d of %d floats differ, first at %s" % (what, (~same).sum(), same.size, np.argwhere(~same)[0])
    else:                                    # pow(): include/flx_math.h's is exactly defined, not correctly rounded — within 2 ulp of the literal answer
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert (ulp <= 2).all() and (ulp == 0).mean() > 0.9, "%s: up to %d ulp, %.3f exact" % (what, ulp.max(), (ulp == 0).mean())


@pytest.mark.parametrize("k", range(5))
def test_filter_chain_literal_known_answers(oracle, k):
    """tests/golden/filter_kat.json.gz: the three filter shaders and the host's pass schedule (pathtracerWGL2.js:462-550: which texture is bound where in which of
    the 3 + 3 + 1 passes, which outputs land nowhere) transcribed statement by statement (tests/analysis/make_filter_kat.py: not through oracle/) on five small
    frames of four "objects" with translucent, rough and background pixels — the oracle's chain gives the same frame bit for bit"""
    from flexlight_hip.capi import GBuffers
    case = _filter_kat_cases()[k]
    p, planes, want = filter_kat_inputs(case)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    gb = GBuffers(fp(planes[0]), fp(planes[1]), fp(planes[2]), fp(planes[3]), fp(planes[4]), None)
    got = np.zeros_like(want)
    assert oracle.lib().flx_oracle_filter(C.byref(p), C.byref(gb), fp(got), 1) == 0
    assert_filter_kat(got, want, case["hdr"], "oracle, case %d" % k)


def _pixel_kat_cases():
    import gzip
    import json
    return json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pixel_kat.json.gz"), "rt"))


def pixel_kat_expectations(case, scenes):
    """-> (scene, params without filter, rows [n, 2] pixel (px, py_gl), literal colour [n, 4], literal G-buffers {name: [n, 4]} for use_filter = 1)"""
    sc = scenes(case["scene"])
    p = sc.frame_params(width=case["width"], height=case["height"], samples=case["samples"], max_reflections=case["bounces"], use_filter=0)
    p.random_seed = case["random_seed"]
    rows = np.array(case["rows"], np.int64)
    f = rows[:, 10:].astype(np.uint32).view(np.float32)
    names = ["color", "color_ip", "original_color", "id", "original_id"]
    return sc, p, rows, f[:, 0:4], {n: f[:, 4 + 4 * k: 8 + 4 * k] for k, n in enumerate(names)}


def assert_pixel_kat(case, rows, want_color, want_gb, frame, gbs, what):
    H = case["height"]
    px, py = rows[:, 0], H - 1 - rows[:, 1]
    same = lambda a, b: ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(axis=1)
    ok = same(np.ascontiguousarray(frame[py, px]), np.ascontiguousarray(want_color))
    assert ok.all(), "%s: colour of %d of %d pixels differs from the literal answer, first (px, py_gl) = %s" % (what, (~ok).sum(), ok.size, rows[~ok][0, :2])
    for n, want in want_gb.items():
        ok = same(np.ascontiguousarray(gbs[n][py, px]), np.ascontiguousarray(want))
        assert ok.all(), "%s: %s of %d of %d pixels differs, first (px, py_gl) = %s" % (what, n, (~ok).sum(), ok.size, rows[~ok][0, :2])


@pytest.mark.parametrize("k", range(5))
def test_whole_pixels_literal_known_answers(oracle, scenes, k):
    """tests/golden/pixel_kat.json.gz: lightTrace with all its bounces and main() (fragment:464-646) run from the shader text over the reference's arrays
    (tests/analysis/make_pixel_kat.py: every sample, bounce and walk, the variables that live across samples, the six outputs; only the primary hit — the
    rasteriser's part, SURVEY.md 8a P0 — is an input) on cornell.obj and on the dragon with its metals, rough surfaces and glass: the oracle's frame without
    filter and its five G-buffers with it equal the literal pixels bit for bit"""
    case = _pixel_kat_cases()[k]
    sc, p, rows, want_color, want_gb = pixel_kat_expectations(case, scenes)
    assert len(rows) >= 90
    # the primary hits the table was made from are the oracle's own, still
    L = oracle.lib()
    L.flx_oracle_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, F3, C.POINTER(C.c_int), C.POINTER(C.c_int), F3]
    L.flx_oracle_primary.restype = None
    import struct
    view = sc.view()
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    for r in rows[:: max(1, len(rows) // 40)]:
        suv, d, ti, tri = F3(), F3(), C.c_int(), C.c_int()
        L.flx_oracle_primary(C.byref(view), C.byref(p), int(r[0]), int(r[1]), suv, C.byref(ti), C.byref(tri), d)
        assert (ti.value, tri.value, [bits(x) for x in suv], [bits(x) for x in d]) == (r[2], r[3], list(r[4:7]), list(r[7:10]))
    frame, _, _ = oracle.render(sc, p)
    p.use_filter = 1
    _, _, gbs = oracle.render(sc, p, gbuffers=True)
    assert_pixel_kat(case, rows, want_color, want_gb, frame, gbs, "oracle, case %d" % k)


@pytest.mark.parametrize("name,w,h", [("cornell", 48, 48), ("cornell_obj", 64, 36), ("theater", 64, 36), ("dragon", 48, 27)])
def test_primary_visibility_agrees_with_a_rasteriser(oracle, scenes, name, w, h):
    """the reference finds the primary hit by DRAWING the triangles (pathtracer_vertex.glsl:40-72: gl_Position = (clip.xy, -1 / (1 + exp(clip.z / 65535)), clip.z);
    depth test, back faces culled, pathtracerWGL2.js:372,713-716) and hands the fragment shader interpolated varyings; the oracle and the kernels cast a ray instead (no text fixes
    the rasteriser's arithmetic).  Here the draw is emulated in float64 — homogeneous coverage of the pixel centre (every triangle, clipped or not), back-face culling by the
    sign of the clip-space determinant, nearest by z / w with the earlier instance winning ties, perspective-correct uv and position — and the ray cast has to name the same
    triangle on every pixel whose centre is not within 1e-6 of an edge or of a second surface, with hit.suv = (distance(absolutePosition, camera), uv.y, uv.z): the distance
    within 1e-5, the barycentrics within 1e-4"""
    import struct
    sc = scenes(name)
    p = sc.frame_params(width=w, height=h, samples=1, max_reflections=1, use_filter=0)
    a = sc.arrays
    g = a["geometry"].astype(np.float64).reshape(-1, 12)
    rot = a["rotation"].astype(np.float64).reshape(-1, 3, 4)[:, :, :3]                      # [matrix][column][row]
    shift = a["shift"].astype(np.float64).reshape(-1, 4)[:, :3]
    ids = a["ids"][: sc.meta["bufferLength"]].astype(np.int64)
    tI = g[ids, 9].astype(np.int64) << 1
    verts = np.stack([g[ids, 0:3], g[ids, 3:6], g[ids, 6:9]], axis=1)                      # [tri][vertex][xyz], object space
    R = np.transpose(rot[tI], (0, 2, 1))                                                   # [tri][row][column]
    world = np.einsum("trc,tvc->tvr", R, verts) + shift[tI][:, None, :]
    cam = np.array(list(p.camera), np.float64)
    V = np.array(list(p.view_matrix), np.float64).reshape(3, 3)                            # uploaded with transpose = true (pathtracerWGL2.js:333): the list's rows are the mat3's rows
    clip = np.einsum("rc,tvc->tvr", V, world - cam)                                       # viewMatrix * (absolutePosition - cameraPosition)
    M = np.transpose(clip, (0, 2, 1))                                                      # [tri]: columns = the vertices' (x, y, w)
    det = np.linalg.det(M)
    ok = np.abs(det) > 1e-300
    Minv = np.zeros_like(M)
    Minv[ok] = np.linalg.inv(M[ok])
    L = oracle.lib()
    L.flx_oracle_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, F3, C.POINTER(C.c_int), C.POINTER(C.c_int), F3]
    L.flx_oracle_primary.restype = None
    view = sc.view()
    compared = skipped = 0
    for py in range(h):
        for px in range(w):
            pix = np.array([(px + 0.5) / w * 2 - 1, (py + 0.5) / h * 2 - 1, 1.0])
            lam = Minv @ pix                                                               # pixel = sum lam_i (x, y, w)_i
            with np.errstate(all="ignore"):
                wdepth = 1.0 / lam.sum(axis=1)
                margin = lam.min(axis=1) * np.abs(wdepth)
                beta = lam * wdepth[:, None]                                               # perspective-correct barycentrics
                zndc = (-1.0 / (1.0 + np.exp(wdepth / 65535.0))) / wdepth
            front = det > 0                                                                # counter-clockwise in the window (gl.frontFace's default) = a positive determinant of the (x, y, w) columns
            cover = ok & (lam.min(axis=1) >= 0) & (wdepth > 0) & (zndc >= -1.0)
            suv, d, ti, tri = F3(), F3(), C.c_int(), C.c_int()
            L.flx_oracle_primary(C.byref(view), C.byref(p), px, py, suv, C.byref(ti), C.byref(tri), d)
            c = np.flatnonzero(cover & front)                                              # gl.enable(CULL_FACE): back faces are not drawn
            cand = (c[np.argmin(zndc[c])], c) if c.size else None
            want_tri = -1 if cand is None else int(ids[cand[0]])
            # pixels too close to call: the centre within 1e-6 of an edge of a covering / nearly covering triangle, or two surfaces within 1e-6 in depth
            near_edge = ok & (np.abs(margin) < 1e-6) & (wdepth > 0) & front
            close = False
            if cand is not None and cand[1].size > 1:
                z = np.sort(wdepth[cand[1]])
                close = (z[1] - z[0]) < 1e-6 * z[0]
            if near_edge.any() or close:
                skipped += 1
                continue
            assert tri.value == want_tri, (px, py, tri.value, want_tri)
            compared += 1
            if want_tri != -1:
                k = cand[0]
                pos = (beta[k][:, None] * world[k]).sum(axis=0)
                want = np.array([np.linalg.norm(pos - cam), beta[k][1], beta[k][2]])
                got = np.array(list(suv), np.float64)
                # (the distance to 1e-5; the barycentrics of a small, distant triangle to 1e-4 of the triangle: float32 Moeller-Trumbore's own conditioning)
                assert abs(got[0] - want[0]) <= 1e-5 * want[0] and np.abs(got[1:] - want[1:]).max() <= 1e-4, (px, py, got, want)
    assert compared > 0.97 * w * h and skipped < 0.03 * w * h, (compared, skipped)


def _temporal_kat_cases():
    import gzip
    import json
    return json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "temporal_kat.json.gz"), "rt"))


def temporal_kat_expectations(case, scenes):
    sc = scenes(case["scene"])
    p = sc.frame_params(width=case["width"], height=case["height"], samples=case["samples"], max_reflections=case["bounces"], use_filter=0, hdr=case["hdr"])
    p.is_temporal = 1
    want = np.array(case["frames"], np.uint32).view(np.float32).reshape(len(case["frames"]), case["height"], case["width"], 4)
    return sc, p, want


@pytest.mark.parametrize("k", range(2))
def test_temporal_sequence_literal_known_answers(oracle, scenes, k):
    """tests/golden/temporal_kat.json.gz: runs of temporal frames — every frame traced from the shader text into the rotating history rings, averaged by the temporal shader
    the host generates (pathtracerWGL2.js:389-402, 571-662; tests/analysis/make_temporal_kat.py) — against the oracle's flx_oracle_render_sequence: the first frames over an
    empty history, the ring wrapping, the zero-padded mat4 groups a background pixel's id matches; bit for bit without the tone mapping, within 2 ulp with it"""
    case = _temporal_kat_cases()[k]
    sc, p, want = temporal_kat_expectations(case, scenes)
    got = oracle.render_sequence(sc, p, want.shape[0])
    for f in range(want.shape[0]):
        assert_filter_kat(np.ascontiguousarray(got[f]), np.ascontiguousarray(want[f]), case["hdr"], "oracle, sequence %d frame %d" % (k, f))


def _aa_kat():
    import gzip
    import json
    return json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aa_kat.json.gz"), "rt"))


def aa_kat_frame(words, W, H):
    return np.array(words, np.uint32).view(np.float32).reshape(H, W, 4)


def assert_aa_kat(got, want, what):
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "%s: %d of %d floats differ, first at %s" % (what, (~same).sum(), same.size, np.argwhere(~same)[0])


def test_fxaa_and_taa_literal_known_answers(oracle):
    """tests/golden/aa_kat.json.gz: the FXAA and TAA shaders (modules/fxaa.js:7-137 with its macros expanded as the preprocessor does, modules/taa.js:11-59 over the nine
    textures renderFrame keeps) transcribed one float32 operation at a time (tests/analysis/make_aa_kat.py) over frames with edges, gradients, noise, a hole of background
    and out-of-range values; TAA with one, four, nine and eleven frames of history — the oracle's passes give the same bits"""
    kat = _aa_kat()
    for k, c in enumerate(kat["fxaa"]):
        W, H = c["width"], c["height"]
        assert_aa_kat(oracle.fxaa(aa_kat_frame(c["frame"], W, H)), aa_kat_frame(c["out"], W, H), "oracle FXAA, frame %d" % k)
    for k, c in enumerate(kat["taa"]):
        W, H = c["width"], c["height"]
        assert_aa_kat(oracle.taa([aa_kat_frame(f, W, H) for f in c["frames_newest_first"]]), aa_kat_frame(c["out"], W, H), "oracle TAA, state %d" % k)


def test_shading_literal_known_answers(oracle):
    """tests/golden/shading_kat.json: forwardTrace (with its GGX / Smith / Schlick helpers, fragment:282-334) and reservoirSample
    (fragment:400-461, incl. its two noise() chains, the showColor / showShadow exits and renderId.w) evaluated from the shader's
    text in float32 arithmetic with a 70-digit sine (tests/analysis/make_shading_kat.py) — not through flx_math.h or the oracle.
    The oracle must return every stored bit; the GPU frames equal the oracle's, so the table holds the kernels too."""
    import json
    from shading_kat_util import oracle_forward_trace, oracle_reservoir
    table = json.load(open(os.path.join(ROOT, "tests", "golden", "shading_kat.json")))
    assert len(table["forward_trace"]) >= 160 and len(table["reservoir"]) >= 96
    for k, row in enumerate(table["forward_trace"]):
        assert oracle_forward_trace(row[:16]) == row[16:], "forwardTrace row %d" % k
    exits = set()
    for k, row in enumerate(table["reservoir"]):
        assert oracle_reservoir(row) == row["out"], "reservoirSample row %d" % k
        exits.add(len(row["lights"]))
    assert {0, 1, 2, 9} <= exits                                 # no light at all, one, two, many
    # one whole iteration of lightTrace's bounce loop (fragment:464-599) on a single triangle under a rotated, shifted transform with two
    # lights: hit point, normals (acos / tan of their deviation), material, noise, Fresnel choice, the G-buffer accumulators (atan in the
    # normal's 4-bit code), reservoirSample with its shadow test against the triangle, the returned colour — 17 outputs per row
    from shading_kat_util import oracle_bounce
    assert len(table["bounce"]) >= 72
    for k, row in enumerate(table["bounce"]):
        assert oracle_bounce(row) == row["out"], "lightTrace bounce row %d" % k


@pytest.mark.parametrize("key", ["configs[0] cornell 256x256 1spp 1b", "configs[1] cornell_obj 1080p 4spp 3b filter", "configs[2] dragon 1080p 8spp 4b",
                                 "configs[3] dragon 4K 8spp 4b", "configs[4] theater 1080p 16spp 6b"])
def test_fullsize_frames_keep_their_hashes(oracle, key):
    """the oracle's frame of every BASELINE.json configuration at FULL size hashes to tests/golden/oracle_fullsize.json, counters
    included: the fixed point that oracle and kernels — one hand, one flx_math.h — cannot leave together unnoticed (the GPU suite
    asserts the same hashes of the GPU frames)"""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "analysis"))
    import make_fullsize_hashes as fs
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")))[key]
    got = fs.oracle_record(key)
    assert got["counters"] == want["counters"]
    assert got["frame"] == want["frame"]
    assert got.get("gbuffers") == want.get("gbuffers")
