#!/usr/bin/env python3
"""Literal known answers for temporal accumulation: a run of frames, each traced from the shader text (make_pixel_kat.py's whole pixels, with main()'s
isTemporal outputs, fragment:620-646) into the rings of RGBA8 history textures the host rotates (modules/pathtracerWGL2.js:389-402), then averaged by the temporal shader the
host GENERATES (pathtracerWGL2.js:571-662, here for temporalSamples = 4: its mat4 groups padded with vec4(0), which a background pixel's zero id matches), with the per-frame
seed frame % temporalSamples (:291, 347).  Pins as in make_filter_kat.py (RGBA8 stores and fetches) and make_pixel_kat.py (the primary hit, from the oracle; relativePosition
as the same barycentric mix of the object-space vertices); pow correctly rounded.  Writes tests/golden/temporal_kat.json.gz.   usage: make_temporal_kat.py [--check]"""
import gzip, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_pixel_kat as P                                               # noqa: E402
from make_pixel_kat import f32, ONE, ZERO, INV_255, Arrays, Scene, bits, oracle_primary, pixel, NaNInBoxTest, g_floor, g_fract, add, scale, sub, length      # noqa: E402
from make_filter_kat import Tex, g_pow, g_mod1                          # noqa: E402

import make_walk_kat                                                     # noqa: E402
make_walk_kat.PIN_NAN = True          # a frame needs every pixel: where a box test meets 0 / 0, min / max are their defining comparisons (GLSL ES 3.00 8.3; the oracle's pin)
OUT = os.path.join(ROOT, "tests", "golden", "temporal_kat.json.gz")
INV_256 = f32(0.00390625)
N = 4                                                                    # config.temporalSamples


def trace_frame(sc, A, attrs, lights, atlases, p, W, H, spp, bounces):
    """the path-trace pass of one temporal frame -> the four textures it renders that the temporal shader reads (colour, colour integer part, location id, original id)"""
    c, ip, loc, oid = Tex(W, H), Tex(W, H), Tex(W, H), Tex(W, H)        # gl.clear: zeros where nothing is drawn
    ambient = [f32(x) for x in p.ambient]; camera = [f32(x) for x in p.camera]
    for py in range(H):
        for px in range(W):
            suv, tI, tri, d = oracle_primary(sc, p, px, py)
            if tri == -1: continue
            ndc = [f32(f32(f32(f32(f32(px) + f32(0.5)) / f32(W)) * f32(2.0)) - ONE), f32(f32(f32(f32(f32(py) + f32(0.5)) / f32(H)) * f32(2.0)) - ONE)]
            plain, _, _, _, _, renderOriginalId = pixel(A, attrs, lights, ambient, f32(p.random_seed), f32(p.min_importancy), spp, bounces, camera, (suv, tI, tri), d, ndc, atlases, p.texture_width)
            fc = plain[0:3]                                               # finalColor * originalColor (useFilter = 0)
            c.store(px, py, [g_fract(fc[0]), g_fract(fc[1]), g_fract(fc[2]), ONE])
            ip.store(px, py, [f32(g_floor(fc[0]) * INV_256), f32(g_floor(fc[1]) * INV_256), f32(g_floor(fc[2]) * INV_256), ONE])
            g = A.entries[tri]
            w0 = f32(f32(ONE - suv[1]) - suv[2])
            rel = add(add(scale(g[0:3], w0), scale(g[3:6], suv[1])), scale(g[6:9], suv[2]))
            div = f32(f32(2.0) * length(sub(rel, camera)))
            with np.errstate(all="ignore"):
                modv = [f32(f32(x - f32(div * g_floor(f32(x / div)))) / div) for x in rel]      # mod(relativePosition, div) / div
            loc.store(px, py, modv + [INV_255])
            oid.store(px, py, renderOriginalId)
    return c, ip, loc, oid


def temporal_shader(rings, x, y, hdr):
    """pathtracerWGL2.js:593-660 as generated for temporalSamples = 4, no filter"""
    cache, cacheIp, cacheId, cacheOId = rings
    ident, originalId = cacheId[0].fetch(x, y), cacheOId[0].fetch(x, y)
    counter = glassCounter = f32(1.0)
    c0, i0 = cache[0].fetch(x, y), cacheIp[0].fetch(x, y)
    centerW = c0[3]
    color = [f32(c0[k] + f32(i0[k] * f32(256.0))) for k in range(3)]
    glassFilter = i0[3]
    zero = [ZERO] * 4
    for i in range(1, N, 4):
        c = [cache[j].fetch(x, y) if j < N else zero for j in range(i, i + 4)]
        ip = [cacheIp[j].fetch(x, y) if j < N else zero for j in range(i, i + 4)]
        ids = [cacheId[j].fetch(x, y) if j < N else zero for j in range(i, i + 4)]
        oids = [cacheOId[j].fetch(x, y) if j < N else zero for j in range(i, i + 4)]
        for k in range(4):
            if ids[k] == ident:
                color = [f32(color[m] + f32(c[k][m] + f32(ip[k][m] * f32(256.0)))) for m in range(3)]
                counter = f32(counter + ONE)
        for k in range(4):
            if oids[k] == originalId:
                glassFilter = f32(glassFilter + ip[k][3])
                glassCounter = f32(glassCounter + ONE)
    color = [f32(v / counter) for v in color]
    if hdr == 1:
        with np.errstate(all="ignore"):
            color = [f32(v / f32(v + ONE)) for v in color]
            inv_gamma = f32(ONE / f32(0.8))
            color = [f32(f32(g_pow(f32(f32(4.0) * v), inv_gamma) / f32(4.0)) * f32(1.3)) for v in color]
    return color + [centerW]


def sequence(name, W, H, spp, bounces, hdr, frames):
    sc = Scene.golden(name)
    A = Arrays(sc)
    attrs = [[f32(x) for x in row] for row in sc.arrays["attributes"].astype(np.float32).reshape(-1, 28)]
    lights = [[f32(x) for x in row] for row in sc.arrays["lights"].astype(np.float32).reshape(-1, 6)]
    atlases = []
    for key, arr in (("albedo", "atlasAlbedo"), ("pbr", "atlasPbr"), ("tpo", "atlasTpo")):
        w, h = sc.meta["atlas"][key]
        atlases.append((sc.arrays[arr].reshape(h, w, 4), w, h))
    rings = [[Tex(W, H) for _ in range(N)] for _ in range(4)]           # TempTexture, TempIpTexture, TempIdTexture, TempOriginalIdTexture
    out = []
    for f in range(frames):
        p = sc.frame_params(width=W, height=H, samples=spp, max_reflections=bounces, use_filter=0, hdr=hdr)
        p.is_temporal = 1
        p.random_seed = float(f % N)
        for r in rings: r.insert(0, r.pop())                             # unshift(pop())
        c, ip, loc, oid = trace_frame(sc, A, attrs, lights, atlases, p, W, H, spp, bounces)
        rings[0][0], rings[1][0], rings[2][0], rings[3][0] = c, ip, loc, oid
        out.append([[bits(v) for x in range(W) for v in temporal_shader(rings, x, y, hdr)] for y in range(H - 1, -1, -1)])      # rows top-down
    return {"scene": name, "width": W, "height": H, "samples": spp, "bounces": bounces, "hdr": hdr, "frames": out}


if __name__ == "__main__":
    data = [sequence("cornell_obj", 40, 22, 1, 2, 0, 6), sequence("dragon", 20, 12, 1, 3, 1, 5)]
    for c in data:
        a = np.array(c["frames"], np.uint32).view(np.float32).reshape(len(c["frames"]), c["height"], c["width"], 4)
        print("%-12s %dx%d hdr %d: %d frames, mean colour per frame %s" % (c["scene"], c["width"], c["height"], c["hdr"], len(c["frames"]), np.round(a[..., :3].mean(axis=(1, 2, 3)), 4)))
    if "--check" in sys.argv:
        assert json.load(gzip.open(OUT, "rt")) == data, "tests/golden/temporal_kat.json.gz is not what this script writes"
        print("matches", OUT)
    else:
        with gzip.GzipFile(OUT, "wb", mtime=0) as fh:
            fh.write(json.dumps(data, separators=(",", ":")).encode())
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
