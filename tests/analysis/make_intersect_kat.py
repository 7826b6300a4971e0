#!/usr/bin/env python3
"""Literal known answers for the intersection routines, written from the shader text — not through oracle/ or include/flx_math.h.

    moellerTrumbore      shaders/pathtracer_fragment.glsl:123-140
    moellerTrumboreCull  shaders/pathtracer_fragment.glsl:143-158
    rayCuboid            shaders/pathtracer_fragment.glsl:161-167

Every operation of the GLSL text is one float32 operation here (numpy float32 scalars: each + - * / rounds once, no contraction), in the order the text
gives; dot(a, b) = a.x b.x + a.y b.y + a.z b.z summed left to right and cross() by its textbook components are the two places where GLSL leaves the order to
the implementation — the same pins as oracle/flx_oracle.c states for itself.  Rows whose min / max would see a NaN are left out (GLSL does not define them).
Writes tests/golden/intersect_kat.json.gz: per routine a list of rows, inputs then outputs, as float32 bit patterns
(moeller_trumbore: tri 9, origin 3, dir 3, l, suv 3; moeller_trumbore_cull: tri 9, origin 3, dir 3, l, hit; ray_cuboid: l, origin 3, dir 3, min 3, max 3, hit).   usage: make_intersect_kat.py [--check]"""
import gzip, json, os, struct, sys
import numpy as np

f32 = np.float32
BIAS = f32(0.0000152587890625)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "intersect_kat.json.gz")


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def sub(a, b): return [f32(a[k] - b[k]) for k in range(3)]
def dot(a, b): return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))
def cross(a, b): return [f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])), f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))]


def moeller_trumbore(t, origin, d, l):
    zero = (f32(0), f32(0), f32(0))
    edge1, edge2 = sub(t[1], t[0]), sub(t[2], t[0])
    pvec = cross(d, edge2)
    det = dot(edge1, pvec)
    if abs(det) < BIAS: return zero
    with np.errstate(all="ignore"):
        inv_det = f32(f32(1) / det)
        tvec = sub(origin, t[0])
        u = f32(dot(tvec, pvec) * inv_det)
        if u < BIAS or u > f32(1): return zero
        qvec = cross(tvec, edge1)
        v = f32(dot(d, qvec) * inv_det)
        uv = f32(u + v)
        if v < BIAS or uv > f32(1): return zero
        s = f32(dot(edge2, qvec) * inv_det)
        if s > l or s <= BIAS: return zero
    return (s, u, v)


def moeller_trumbore_cull(t, origin, d, l):
    edge1, edge2 = sub(t[1], t[0]), sub(t[2], t[0])
    pvec = cross(d, edge2)
    det = dot(edge1, pvec)
    with np.errstate(all="ignore"):
        inv_det = f32(f32(1) / det)
        if det < BIAS: return 0
        tvec = sub(origin, t[0])
        u = f32(dot(tvec, pvec) * inv_det)
        if u < BIAS or u > f32(1): return 0
        qvec = cross(tvec, edge1)
        v = f32(dot(d, qvec) * inv_det)
        if v < BIAS or f32(u + v) > f32(1): return 0
        s = f32(dot(edge2, qvec) * inv_det)
    return int(s <= l and s > BIAS)


def gmin(x, y): return y if y < x else x                # GLSL ES 3.00: min(x, y) = y if y < x, otherwise x
def gmax(x, y): return y if x < y else x                # max(x, y) = y if x < y, otherwise x


def ray_cuboid(l, origin, d, mn, mx, pinned_nan=False):
    """-> 0 / 1, or None when a NaN reaches min / max (GLSL leaves that open) — unless pinned_nan: then min / max are exactly their defining comparisons, as the oracle pins them"""
    with np.errstate(all="ignore"):
        v0 = [f32(f32(mn[k] - origin[k]) / d[k]) for k in range(3)]
        v1 = [f32(f32(mx[k] - origin[k]) / d[k]) for k in range(3)]
    if not pinned_nan and any(np.isnan(x) for x in v0 + v1): return None
    tmin = gmax(gmax(gmin(v0[0], v1[0]), gmin(v0[1], v1[1])), gmin(v0[2], v1[2]))
    tmax = gmin(gmin(gmax(v0[0], v1[0]), gmax(v0[1], v1[1])), gmax(v0[2], v1[2]))
    return int(tmax >= gmax(tmin, BIAS) and tmin < l)


def rows():
    rng = np.random.default_rng(20261004)
    tri_rows, cull_rows, box_rows = [], [], []
    v = lambda scale=1.0: [f32(x) for x in rng.normal(0.0, scale, 3)]
    for case in range(1500):
        t = [v(2.0), v(2.0), v(2.0)]
        kind = case % 6
        if kind < 3:                                  # aimed at a point of the triangle's plane near (or in) the triangle
            a, b = rng.uniform(-0.1, 1.1, 2)
            if kind == 0: a, b = (lambda w: (w[0], w[1]))(rng.dirichlet([1.0, 1.0, 1.0]))      # inside the triangle
            if kind == 2: a, b = rng.choice([0.0, 1.0, 2.0 ** -16, 1.0 - 2.0 ** -16, 0.5]), rng.choice([0.0, 2.0 ** -16, 0.5, 1.0])     # on the edges and the BIAS-wide cracks
            target = [f32(t[0][k] + f32(a) * (t[1][k] - t[0][k]) + f32(b) * (t[2][k] - t[0][k])) for k in range(3)]
            origin = v(4.0)
            d = np.array([target[k] - origin[k] for k in range(3)], np.float64)
            n = np.linalg.norm(d)
            d = [f32(x) for x in (d / n if n > 0 else [0, 0, 1])]
            l = f32(rng.choice([1e9, float(n) * 0.9, float(n) * 1.1, float(f32(n))]))
        elif kind == 3:                               # anywhere
            origin, d, l = v(4.0), v(1.0), f32(1e9)
        elif kind == 4:                               # (nearly) parallel to the plane: |det| around BIAS
            e1 = np.array(t[1], np.float64) - np.array(t[0], np.float64)
            e2 = np.array(t[2], np.float64) - np.array(t[0], np.float64)
            nrm = np.cross(e1, e2)
            dd = e1 * rng.normal() + e2 * rng.normal() + nrm * rng.choice([0.0, 1e-6, 1e-5, 1e-4])
            d = [f32(x) for x in dd / max(np.linalg.norm(dd), 1e-30)]
            origin, l = v(4.0), f32(1e9)
        else:                                         # axis-aligned rays, degenerate triangles
            d = [f32(0), f32(0), f32(0)]
            d[rng.integers(0, 3)] = f32(rng.choice([-1.0, 1.0]))
            origin, l = v(3.0), f32(1e9)
            if rng.random() < 0.3: t[2] = list(t[1])
        s = moeller_trumbore(t, origin, d, l)
        inputs = [bits(x) for p in t for x in p] + [bits(x) for x in origin] + [bits(x) for x in d] + [bits(l)]
        tri_rows.append(inputs + [bits(x) for x in s])
        cull_rows.append(inputs + [moeller_trumbore_cull(t, origin, d, l)])
    for case in range(1500):
        c, h = v(3.0), [f32(abs(x) + 0.01) for x in rng.normal(0.0, 1.5, 3)]
        mn, mx = [f32(c[k] - h[k]) for k in range(3)], [f32(c[k] + h[k]) for k in range(3)]
        origin = v(5.0)
        kind = case % 5
        if kind == 0: d = v(1.0)
        elif kind == 1:                               # through the box
            p = [f32(mn[k] + f32(rng.random()) * (mx[k] - mn[k])) for k in range(3)]
            dd = np.array([p[k] - origin[k] for k in range(3)], np.float64)
            d = [f32(x) for x in dd / max(np.linalg.norm(dd), 1e-30)]
        elif kind == 2:                               # zero direction components: +-inf slabs
            d = v(1.0)
            d[rng.integers(0, 3)] = f32(0.0) if rng.random() < 0.5 else f32(-0.0)
        elif kind == 3:                               # from inside
            origin = [f32(mn[k] + f32(rng.random()) * (mx[k] - mn[k])) for k in range(3)]
            d = v(1.0)
        else:                                         # grazing a face
            d = v(1.0)
            k = rng.integers(0, 3)
            origin[k] = mx[k] if rng.random() < 0.5 else mn[k]
            d[k] = f32(rng.choice([0.0, 1e-7, -1e-7, 1.0]))
        dist = float(np.linalg.norm(np.array(c, np.float64) - np.array(origin, np.float64)))
        l = f32(rng.choice([1e9, dist, dist * 0.5, 4.2949673e9]))
        r = ray_cuboid(l, origin, d, mn, mx)
        if r is None: continue
        box_rows.append([bits(l)] + [bits(x) for x in origin] + [bits(x) for x in d] + [bits(x) for x in mn] + [bits(x) for x in mx] + [r])
    return {"moeller_trumbore": tri_rows, "moeller_trumbore_cull": cull_rows, "ray_cuboid": box_rows}


if __name__ == "__main__":
    data = rows()
    hits = sum(1 for r in data["moeller_trumbore"] if r[16:] != [0, 0, 0])
    print("moellerTrumbore %d rows (%d hits), cull %d rows (%d hits), rayCuboid %d rows (%d hits)" % (len(data["moeller_trumbore"]), hits, len(data["moeller_trumbore_cull"]),
          sum(r[-1] for r in data["moeller_trumbore_cull"]), len(data["ray_cuboid"]), sum(r[-1] for r in data["ray_cuboid"])))
    if "--check" in sys.argv:
        assert json.load(gzip.open(OUT, "rt")) == data, "tests/golden/intersect_kat.json.gz is not what this script writes"
        print("matches", OUT)
    else:
        with gzip.GzipFile(OUT, "wb", mtime=0) as fh:
            fh.write(json.dumps(data, separators=(",", ":")).encode())
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
