#!/usr/bin/env python3
"""Primary rays of an 8 x 8 screen tile: how many entries does the tile visit as a whole (the union over its 64 rays) against
what one ray visits?  Analysis only (float64 restatement of the primary walk, statistics — not a parity tool).
usage: primary_union.py [scene] [tiles]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip.scene_io import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sc = Scene.golden(name)
g = sc.arrays["geometry"].reshape(-1, 12).astype(np.float64)
rot = sc.arrays["rotation"].reshape(-1, 2, 3, 4)[:, :, :, :3].astype(np.float64)      # [t][0 fwd | 1 inv][column][xyz]
shf = sc.arrays["shift"].reshape(-1, 2, 4)[:, :, :3].astype(np.float64)
p = sc.frame_params(use_filter=0)
W, H = p.width, p.height
cam = np.array(list(p.camera), np.float64)
iv = np.linalg.inv(np.array(list(p.view_matrix), np.float64).reshape(3, 3))
BIAS = 2.0 ** -16

def walk(o, d):
    visited = []
    i, n, minlen, cached = 0, g.shape[0], 2.0 ** 32, 0
    to, td = o, d
    while i < n:
        e = g[i]; visited.append(i)
        if e[10] == 0: break
        t = int(e[9])
        if t != cached:
            cached = t
            m = rot[t, 1]                      # inverse rotation, columns
            to = m.T @ (o + shf[t, 1]) if False else (m[0] * (o + shf[t, 1])[0] + m[1] * (o + shf[t, 1])[1] + m[2] * (o + shf[t, 1])[2])
            td = m[0] * d[0] + m[1] * d[1] + m[2] * d[2]
        if e[10] == 1:
            with np.errstate(divide="ignore", invalid="ignore"):
                v0 = (e[0:3] - to) / td; v1 = (e[3:6] - to) / td
            tmin = np.max(np.minimum(v0, v1)); tmax = np.min(np.maximum(v0, v1))
            i = i + 1 if (tmax >= max(tmin, BIAS) and tmin < minlen) else i + 1 + int(e[6])
        else:
            a, b, c = e[0:3], e[3:6], e[6:9]
            e1, e2 = b - a, c - a
            pv = np.cross(td, e2); det = e1 @ pv
            if det < 0:
                inv = 1.0 / det; tv = to - a; u = (tv @ pv) * inv
                if 0 <= u <= 1:
                    q = np.cross(tv, e1); v = (td @ q) * inv
                    if v >= 0 and u + v <= 1:
                        s = (e2 @ q) * inv
                        if s < minlen and s > 0: minlen = s
            i += 1
    return visited

rng = np.random.default_rng(1)
tx, ty = (W + 7) // 8, (H + 7) // 8
per_ray, union, maxray = [], [], []
for _ in range(n_tiles):
    bx, by = rng.integers(0, tx), rng.integers(0, ty)
    u = set(); m = 0
    for l in range(64):
        px, py = bx * 8 + (l & 7), by * 8 + (l >> 3)
        if px >= W or py >= H: continue
        nx = (px + 0.5) / W * 2 - 1; ny = (H - 1 - py + 0.5) / H * 2 - 1
        d = iv @ np.array([nx, ny, 1.0]); d /= np.linalg.norm(d)
        v = walk(cam, d)
        per_ray.append(len(v)); m = max(m, len(v)); u.update(v)
    union.append(len(u)); maxray.append(m)
print("longest ray of a tile: percentiles 50/90/99/100 =", [int(np.percentile(maxray, q)) for q in (50, 90, 99, 100)],
      " union: ", [int(np.percentile(union, q)) for q in (50, 90, 99, 100)])
print("%s %dx%d, %d tiles: entries per ray %.1f, longest ray of a tile %.1f, union of a tile %.1f (max %d)" %
      (name, W, H, n_tiles, np.mean(per_ray), np.mean(maxray), np.mean(union), max(union)))
