#!/usr/bin/env python3
"""Literal known answers for the denoise chain, written from the shader and the host text — not through oracle/.

    first filter    shaders/pathtracer_first_filter.glsl:17-123
    second filter   shaders/pathtracer_second_filter.glsl:17-79
    final filter    shaders/pathtracer_final_filter.glsl:11-71
    pass schedule   modules/pathtracerWGL2.js:462-550 (firstPasses = secondPasses = 3: config.js:9-10), run here statement by statement over a dictionary of textures

Every operation of the GLSL text is one float32 operation, in the order the text gives (a vec4 expression component by component); tanh and pow are correctly rounded
from 50-digit arithmetic.  What the texts leave open is pinned as oracle/flx_oracle_filter.c pins it, and said here: render targets are RGBA8 (a store clamps, NaN -> 0,
floor(255 x + 0.5); texelFetch gives byte / 255 in float32); texelFetch outside the texture gives zeros; the post vertex shader's clipSpace addresses the pixel's own
texel; the first filter's renderColorIp, which it does not initialise, starts as zeros; IdRenderTexture[2] and [3], which do not exist, attach nothing (the output is
dropped).  Writes tests/golden/filter_kat.json.gz: per case the five RGBA8 input planes (rows top-down) and the final filter's float output as bit patterns.
usage: make_filter_kat.py [--check]"""
import gzip, json, os, struct, sys
from decimal import Decimal, getcontext
import numpy as np

getcontext().prec = 50
f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "filter_kat.json.gz")
INV_256 = f32(0.00390625)
ZERO4 = [f32(0)] * 4


def bits(x): return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def rnd(d):                                             # Decimal -> nearest float32 (round half even on the exact value: float() of a 50-digit string is correctly rounded to double, then to float32:
    return f32(np.float64(str(d)))                     #  a double rounding that could only bite within 2^-29 ulp of a float32 tie, which the checks below rule out)


_TANH = {}
def g_tanh(x):
    x = f32(x)
    k = bits(x)
    if k not in _TANH:
        d = Decimal(float(x))
        e2 = (2 * d).exp()
        _TANH[k] = rnd((e2 - 1) / (e2 + 1))
    return _TANH[k]


def g_pow(x, y):                                        # x >= 0 here
    x = f32(x)
    if np.isnan(x): return x
    if x == 0: return f32(0)
    if np.isinf(x): return x
    return rnd((Decimal(float(x)).ln() * Decimal(float(y))).exp())


def g_floor(x): return f32(np.floor(x))
def g_sign(x): return f32(1) if x > 0 else (f32(-1) if x < 0 else f32(0))
def g_mod1(x): return f32(x - f32(f32(1) * g_floor(f32(x / f32(1)))))      # mod(x, 1.0) = x - 1.0 * floor(x / 1.0)
def g_max(x, y): return y if x < y else x
def g_min(x, y): return y if y < x else x
def itrunc(x):
    x = float(x)
    return int(x) if np.isfinite(x) else 0


class Tex:                                              # an RGBA8 texture; y counts from the bottom like gl_FragCoord
    def __init__(self, W, H, data=None):
        self.W, self.H = W, H
        self.d = np.zeros((H, W, 4), np.uint8) if data is None else np.array(data, np.uint8).reshape(H, W, 4)      # row 0 = top

    def fetch(self, x, y):
        if x < 0 or y < 0 or x >= self.W or y >= self.H: return list(ZERO4)
        q = self.d[self.H - 1 - y, x]
        return [f32(f32(int(c)) / f32(255)) for c in q]

    def store(self, x, y, v):
        out = []
        for c in v:
            if not (c > 0): out.append(0)
            elif c >= 1: out.append(255)
            else: out.append(int(f32(f32(c * f32(255)) + f32(0.5))))
        self.d[self.H - 1 - y, x] = out


STENCIL1 = [(-1, 0), (0, -1), (0, 1), (1, 0)]
STENCIL3_37 = [(-3, -1), (-3, 0), (-3, 1), (-2, -2), (-2, -1), (-2, 0), (-2, 1), (-2, 2), (-1, -3), (-1, -2), (-1, -1), (-1, 0), (-1, 1), (-1, 2), (-1, 3),
               (0, -3), (0, -2), (0, -1), (0, 0), (0, 1), (0, 2), (0, 3), (1, -3), (1, -2), (1, -1), (1, 0), (1, 1), (1, 2), (1, 3),
               (2, -2), (2, -1), (2, 0), (2, 1), (2, 2), (3, -1), (3, 0), (3, 1)]
STENCIL3_36 = [s for s in STENCIL3_37 if s != (0, 0)]


def add4(a, b): return [f32(a[k] + b[k]) for k in range(4)]
def scale4(a, s): return [f32(a[k] * s) for k in range(4)]


def first_filter(tc, tip, toc, tid, toid, x, y):
    centerColor, centerColorIp, centerOColor, centerId = tc.fetch(x, y), tip.fetch(x, y), toc.fetch(x, y), tid.fetch(x, y)
    centerIdw = itrunc(f32(centerId[3] * f32(255.0)))
    centerLightNum, centerShadow = centerIdw // 2, centerIdw % 2
    renderId = list(centerId)
    renderColorIp = list(ZERO4)                          # (not initialised by the shader: pinned)
    centerOId = toid.fetch(x, y)
    color, count = list(ZERO4), f32(0)
    if centerOId[3] != 0.0 and centerColorIp[3] != 0.0:
        ident = centerId
        ids = [tid.fetch(x + s[0], y + s[1]) for s in STENCIL1]
        oIds = [toid.fetch(x + s[0], y + s[1]) for s in STENCIL1]
        ipws = [tip.fetch(x + s[0], y + s[1])[3] for s in STENCIL1]
        vote = [0, 0, 0, 0]
        for i in range(4):
            if ipws[i] == 0.0:
                vote[i] = 1
                if ids[i][0:3] == ident[0:3] and oIds[i] == centerOId: vote[i] += 1
                for j in range(i + 1, 4):
                    if ids[i][0:3] == ids[j][0:3] and oIds[i] == oIds[j]: vote[i] += 1
        maxVote, idNumber = vote[0], 0
        for i in range(1, 4):
            if vote[i] >= maxVote: maxVote, idNumber = vote[i], i
        renderId = ids[idNumber]
        renderColorIp[3] = g_max(f32(f32(1.0) - g_sign(f32(maxVote))), f32(0.0))
    if centerOColor[3] == 0.0:
        color, count = list(centerColor), f32(1.0)
    else:
        k = f32(f32(1.0) + centerOColor[3])
        for s in STENCIL3_37:
            cx = x + itrunc(f32(f32(f32(f32(s[0]) * k) * k) * f32(3.5)))
            cy = y + itrunc(f32(f32(f32(f32(s[1]) * k) * k) * f32(3.5)))
            ident, originalId = tid.fetch(cx, cy), toid.fetch(cx, cy)
            idW = itrunc(f32(ident[3] * f32(255.0)))
            lightNum, shadow = idW // 2, idW % 2
            nextColor, nextColorIp = tc.fetch(cx, cy), tip.fetch(cx, cy)
            if centerId[0:3] == ident[0:3] and centerOId == originalId and (centerLightNum != lightNum or centerShadow == shadow):
                color = add4(color, add4(nextColor, scale4(nextColorIp, f32(256.0))))
                count = f32(count + f32(1))
    with np.errstate(all="ignore"):
        invCount = f32(f32(1.0) / count)
        sg = g_sign(centerColor[3])
        c3 = [f32(color[k] * invCount) for k in range(3)]
        renderColor = [f32(sg * g_mod1(c3[0])), f32(sg * g_mod1(c3[1])), f32(sg * g_mod1(c3[2])), f32(sg * centerColor[3])]
        outIp = [f32(sg * f32(g_floor(c3[0]) * INV_256)), f32(sg * f32(g_floor(c3[1]) * INV_256)), f32(sg * f32(g_floor(c3[2]) * INV_256)), f32(sg * renderColorIp[3])]
    return renderColor, outIp, renderId


def second_filter(tc, tip, toc, tid, toid, x, y):
    centerColor, centerColorIp, centerOColor, centerId, centerOId = tc.fetch(x, y), tip.fetch(x, y), toc.fetch(x, y), tid.fetch(x, y), toid.fetch(x, y)
    color = add4(centerColor, scale4([centerColorIp[0], centerColorIp[1], centerColorIp[2], f32(0.0)], f32(256.0)))
    oColor = list(centerOColor)
    ipw, count, oCount = centerColorIp[3], f32(1.0), f32(1.0)
    f = f32(f32(1.0) + f32(f32(2.0) * g_tanh(f32(centerOColor[3] + f32(centerOId[3] * f32(4.0))))))
    for s in STENCIL3_36:
        cx, cy = x + itrunc(f32(f32(s[0]) * f)), y + itrunc(f32(f32(s[1]) * f))
        ident, nextOId, nextColor, nextColorIp, nextOColor = tid.fetch(cx, cy), toid.fetch(cx, cy), tc.fetch(cx, cy), tip.fetch(cx, cy), toc.fetch(cx, cy)
        if centerOId[0:3] == nextOId[0:3]:
            if g_min(centerOId[3], nextOId[3]) > f32(0.1) and (ident == centerId or g_max(nextColorIp[3], centerColorIp[3]) >= f32(0.1)):
                color = add4(color, add4(nextColor, scale4([nextColorIp[0], nextColorIp[1], nextColorIp[2], f32(0.0)], f32(256.0))))
                count = f32(count + f32(1))
                ipw = f32(ipw + nextColorIp[3])
                oColor = add4(oColor, nextOColor)
                oCount = f32(oCount + f32(1))
            elif ident[0:3] == centerId[0:3]:
                color = add4(color, add4(nextColor, scale4([nextColorIp[0], nextColorIp[1], nextColorIp[2], f32(0.0)], f32(256.0))))
                count = f32(count + f32(1))
    with np.errstate(all="ignore"):
        invCount = f32(f32(1.0) / count)
        w = centerColor[3]
        c3 = [f32(color[k] * invCount) for k in range(3)]
        renderColor = [f32(w * g_mod1(c3[0])), f32(w * g_mod1(c3[1])), f32(w * g_mod1(c3[2])), f32(w * f32(color[3] * invCount))]
        renderColorIp = [f32(w * f32(g_floor(c3[0]) * INV_256)), f32(w * f32(g_floor(c3[1]) * INV_256)), f32(w * f32(g_floor(c3[2]) * INV_256)), f32(w * ipw)]
        renderOriginalColor = [f32(f32(w * oColor[k]) / oCount) for k in range(4)]
    return renderColor, renderColorIp, renderOriginalColor


def final_filter(tc, tip, toc, tid, toid, x, y, hdr):
    centerColor, centerColorIp, centerOColor, centerId, centerOId = tc.fetch(x, y), tip.fetch(x, y), toc.fetch(x, y), tid.fetch(x, y), toid.fetch(x, y)
    color, oColor, count, oCount = list(ZERO4), list(ZERO4), f32(0.0), f32(0.0)
    f = f32(f32(0.7) + f32(f32(2.0) * g_tanh(f32(centerOColor[3] + f32(centerOId[3] * f32(4.0))))))
    for s in STENCIL3_37:
        cx, cy = x + itrunc(f32(f32(s[0]) * f)), y + itrunc(f32(f32(s[1]) * f))
        ident, nextOId, nextColor, nextColorIp, nextOColor = tid.fetch(cx, cy), toid.fetch(cx, cy), tc.fetch(cx, cy), tip.fetch(cx, cy), toc.fetch(cx, cy)
        blurTranslucent = g_max(nextColorIp[3], centerColorIp[3]) != 0.0 and g_min(centerOId[3], nextOId[3]) > 0.0
        if blurTranslucent and centerOId[0:3] == nextOId[0:3]:
            oColor = add4(oColor, nextOColor)
            oCount = f32(oCount + f32(1))
        if (blurTranslucent or centerId[0:3] == ident[0:3]) and centerOId[0:3] == nextOId[0:3]:
            color = add4(color, add4(nextColor, scale4(nextColorIp, f32(255.0))))
            count = f32(count + f32(1))
    if centerColor[3] > 0.0:
        with np.errstate(all="ignore"):
            fc = [f32(color[k] / count) for k in range(3)]
            m = list(centerOColor[0:3]) if oCount == 0.0 else [f32(oColor[k] / oCount) for k in range(3)]
            fc = [f32(fc[k] * m[k]) for k in range(3)]
            if hdr == 1:
                fc = [f32(fc[k] / f32(fc[k] + f32(1.0))) for k in range(3)]
                inv_gamma = f32(f32(1.0) / f32(0.8))
                fc = [f32(f32(g_pow(f32(f32(4.0) * fc[k]), inv_gamma) / f32(4.0)) * f32(1.3)) for k in range(3)]
        return [fc[0], fc[1], fc[2], f32(1.0)]
    return list(ZERO4)


def chain(W, H, hdr, planes, firstPasses=3, secondPasses=3):
    """modules/pathtracerWGL2.js:462-550, the statements that touch textures; planes: R0, Ip0, O0, Id0, OId as Tex"""
    RenderTexture = [planes[0], Tex(W, H), Tex(W, H), Tex(W, H)]
    IpRenderTexture = [planes[1], Tex(W, H), Tex(W, H), Tex(W, H)]
    OriginalRenderTexture = [planes[2], Tex(W, H)]
    IdRenderTexture = [planes[3], Tex(W, H)]
    OriginalIdRenderTexture = planes[4]
    PostProgram = [first_filter, first_filter, second_filter, second_filter]
    n = nId = nOriginal = 0
    for i in range(firstPasses + secondPasses):
        np_ = (i % 2) ^ 1
        npOriginal = (int(np.fmod(i - firstPasses, 2))) ^ 1              # JavaScript's % keeps the sign of the dividend: (-3 % 2) = -1, (-1) ^ 1 = -2
        if firstPasses <= i: np_ += 2
        att0, att1 = RenderTexture[np_], IpRenderTexture[np_]
        if firstPasses <= i - 2: att2 = OriginalRenderTexture[npOriginal]
        else: att2 = IdRenderTexture[np_] if np_ < len(IdRenderTexture) else None       # IdRenderTexture[2], [3] are undefined: nothing attached
        for t in (att0, att1, att2):                                    # gl.clear
            if t is not None: t.d[:] = 0
        src = (RenderTexture[n], IpRenderTexture[n], OriginalRenderTexture[nOriginal], IdRenderTexture[nId], OriginalIdRenderTexture)
        outs = [[PostProgram[n](*src, x, y) for x in range(W)] for y in range(H)]          # (a pass reads what was there before it: no attachment is also bound as a source)
        assert att0 not in src and att1 not in src and (att2 is None or att2 not in src)
        for y in range(H):
            for x in range(W):
                a, b, c = outs[y][x]
                att0.store(x, y, a); att1.store(x, y, b)
                if att2 is not None: att2.store(x, y, c)
        n = np_
        if firstPasses <= i: nOriginal = npOriginal
        else: nId = np_
    index = 2 + (firstPasses + secondPasses) % 2
    indexId, indexOriginal = firstPasses % 2, secondPasses % 2
    src = (RenderTexture[index], IpRenderTexture[index], OriginalRenderTexture[indexOriginal], IdRenderTexture[indexId], OriginalIdRenderTexture)
    return [[final_filter(*src, x, y, hdr) for x in range(W)] for y in range(H - 1, -1, -1)]        # rows top-down


def make_case(seed, W, H, hdr):
    rng = np.random.default_rng(seed)
    # a few "objects": vertical / diagonal regions with their own ids, original ids, roughness-like w and translucency flag
    region = ((np.arange(W)[None, :] + np.arange(H)[:, None] // 2) // max(3, W // 4)) % 4
    if seed % 2: region = (np.arange(W)[None, :] // 5 + 2 * (np.arange(H)[:, None] // 6)) % 4
    ids = rng.integers(0, 256, (4, 3))
    oids = rng.integers(0, 256, (4, 3))
    R = np.zeros((H, W, 4), np.uint8); Ip = np.zeros_like(R); O = np.zeros_like(R); Id = np.zeros_like(R); OId = np.zeros_like(R)
    for y in range(H):
        for x in range(W):
            r = region[y, x]
            covered = not (r == 3 and seed % 3 == 0)                       # one region may be background (alpha 0)
            R[y, x] = list(rng.integers(0, 256, 3)) + [255 if covered else 0]
            Ip[y, x] = list(rng.integers(0, 3, 3)) + [int(rng.choice([0, 0, 0, 40, 255])) if r == 1 else 0]
            O[y, x] = list(rng.integers(0, 256, 3)) + [int(rng.choice([0, 0, 12, 60, 140, 255]))]
            Id[y, x] = list(ids[r]) + [int(rng.choice([0, 1, 2, 3, 5]))]
            OId[y, x] = list(oids[r if rng.random() > 0.1 else (r + 1) % 4]) + [int(rng.choice([0, 20, 30, 64, 200]))]
            if not covered: Ip[y, x] = 0; O[y, x] = 0; Id[y, x] = 0; OId[y, x] = 0
    planes = [Tex(W, H, p) for p in (R, Ip, O, Id, OId)]
    inputs = [p.d.reshape(-1).tolist() for p in planes]
    out = chain(W, H, hdr, planes)
    return {"width": W, "height": H, "hdr": hdr, "planes": inputs, "out": [[bits(c) for px in row for c in px] for row in out]}


if __name__ == "__main__":
    cases = [make_case(1, 24, 16, 0), make_case(2, 24, 16, 1), make_case(3, 19, 13, 0), make_case(4, 33, 9, 1), make_case(6, 16, 24, 0)]
    for c in cases:
        flat = np.array([v for row in c["out"] for v in row], np.uint32).view(np.float32).reshape(c["height"], c["width"], 4)
        print("%dx%d hdr %d: %d covered pixels, mean colour %s" % (c["width"], c["height"], c["hdr"], int((flat[..., 3] > 0).sum()), np.nanmean(flat[..., :3], axis=(0, 1))))
    if "--check" in sys.argv:
        assert json.load(gzip.open(OUT, "rt")) == cases, "tests/golden/filter_kat.json.gz is not what this script writes"
        print("matches", OUT)
    else:
        with gzip.GzipFile(OUT, "wb", mtime=0) as fh:
            fh.write(json.dumps(cases, separators=(",", ":")).encode())
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
