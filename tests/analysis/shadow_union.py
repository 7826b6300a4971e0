#!/usr/bin/env python3
"""Bounce-0 shadow rays of an 8 x 8 screen tile (one sample each): entries per ray against the union over the tile — would a
wave-wide walk pay for them as it does for primary rays?  Analysis only: float64, the light jitter drawn from numpy instead of
the shader's noise(), hit point offset ignored.  usage: shadow_union.py [tiles]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip.scene_io import Scene
n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sc = Scene.golden("dragon")
g = sc.arrays["geometry"].reshape(-1, 12).astype(np.float64)
rot = sc.arrays["rotation"].reshape(-1, 2, 3, 4)[:, :, :, :3].astype(np.float64)
shf = sc.arrays["shift"].reshape(-1, 2, 4)[:, :, :3].astype(np.float64)
light = sc.arrays["lights"].astype(np.float64)
p = sc.frame_params(use_filter=0)
W, H = p.width, p.height
cam = np.array(list(p.camera), np.float64)
iv = np.linalg.inv(np.array(list(p.view_matrix), np.float64).reshape(3, 3))
BIAS = 2.0 ** -16

def xf(t, o, d, inv=True):
    m = rot[t, 1]
    oo = o + shf[t, 1]
    return m[0] * oo[0] + m[1] * oo[1] + m[2] * oo[2], m[0] * d[0] + m[1] * d[1] + m[2] * d[2]

def walk(o, d, anyhit, lim):
    visited = []; i, n, minlen, cached = 0, g.shape[0], lim, 0
    to, td = o, d; best = None
    while i < n:
        e = g[i]; visited.append(i)
        if e[10] == 0: break
        t = int(e[9])
        if t != cached:
            cached = t; to, td = xf(t, o, d)
            if anyhit: td = td / np.linalg.norm(td)
        if e[10] == 1:
            with np.errstate(divide="ignore", invalid="ignore"):
                v0 = (e[0:3] - to) / td; v1 = (e[3:6] - to) / td
            tmin = np.max(np.minimum(v0, v1)); tmax = np.min(np.maximum(v0, v1))
            i = i + 1 if (tmax >= max(tmin, BIAS) and tmin < minlen) else i + 1 + int(e[6])
        else:
            a = e[0:3]; e1 = e[3:6] - a; e2 = e[6:9] - a
            pv = np.cross(td, e2); det = e1 @ pv
            ok = det >= BIAS if anyhit else abs(det) >= BIAS
            if ok:
                inv = 1.0 / det; tv = to - a; u = (tv @ pv) * inv
                if BIAS <= u <= 1:
                    q = np.cross(tv, e1); v = (td @ q) * inv
                    if v >= BIAS and u + v <= 1:
                        s = (e2 @ q) * inv
                        if BIAS < s <= minlen:
                            if anyhit: return visited, s
                            minlen = s; best = s
            i += 1
    return visited, best

rng = np.random.default_rng(2)
tx, ty = (W + 7) // 8, (H + 7) // 8
per_ray, union, longest = [], [], []
done = 0
while done < n_tiles:
    bx, by = rng.integers(0, tx), rng.integers(0, ty)
    u = set(); m = 0; rays = 0
    for l in range(64):
        px, py = bx * 8 + (l & 7), by * 8 + (l >> 3)
        if px >= W or py >= H: continue
        nx = (px + 0.5) / W * 2 - 1; ny = (H - 1 - py + 0.5) / H * 2 - 1
        d = iv @ np.array([nx, ny, 1.0]); d /= np.linalg.norm(d)
        _, s = walk(cam, d, False, 2.0 ** 32)
        if s is None: continue
        hit = cam + d * s
        target = light[0:3] + rng.uniform(-1, 1, 3) * light[4]
        ld = target - hit; dist = np.linalg.norm(ld)
        v, _ = walk(hit + ld / dist * 1e-3, ld / dist, True, dist)
        per_ray.append(len(v)); m = max(m, len(v)); u.update(v); rays += 1
    if rays < 32: continue
    union.append(len(u)); longest.append(m); done += 1
print("dragon %dx%d, %d tiles: shadow entries per ray %.1f, longest of a tile %.1f, union of a tile %.1f" % (W, H, n_tiles, np.mean(per_ray), np.mean(longest), np.mean(union)))
print("longest: percentiles 50/90/99 =", [int(np.percentile(longest, q)) for q in (50, 90, 99)], " union:", [int(np.percentile(union, q)) for q in (50, 90, 99)])
