#!/usr/bin/env python3
"""Literal known answers for the two anti-aliasing passes, from their shader texts — not through oracle/.

    FXAA   modules/fxaa.js:7-137   (macros expanded as the preprocessor does: `range_min_max.y * 1.0 / 2.0`, `1.0 / 9.0 * (...)`)
    TAA    modules/taa.js:11-59 over the nine textures renderFrame() keeps (taa.js:109-127), newest first

One float32 operation per operation of the text.  The input textures are RGBA8 (the frame the renderer drew: a store clamps, NaN -> 0, floor(255 x + 0.5); texelFetch
gives byte / 255; outside the texture zeros) — the pins of make_filter_kat.py.  Writes tests/golden/aa_kat.json.gz: float input frames and the passes' float outputs as bit
patterns.   usage: make_aa_kat.py [--check]"""
import gzip, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_filter_kat import Tex, f32, bits, g_max, g_min                 # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "aa_kat.json.gz")
ONE, ZERO = f32(1.0), f32(0.0)


def add4(a, b): return [f32(x + y) for x, y in zip(a, b)]
def mix4(a, b, t): return [f32(f32(x * f32(ONE - t)) + f32(y * t)) for x, y in zip(a, b)]


def fxaa_pixel(t, px, py):
    fetch = lambda x, y: t.fetch(px + x, py + y)
    luma_of = lambda c: f32(f32(f32(c[1] * f32(f32(0.587) / f32(0.299))) + c[0]) * c[3])
    tex_luma = lambda x, y: luma_of(fetch(x, y))

    def contrast(x, y):
        return (g_min(tex_luma(x, y), g_min(g_min(tex_luma(x, y - 1), tex_luma(x - 1, y)), g_min(tex_luma(x, y + 1), tex_luma(x + 1, y)))),
                g_max(tex_luma(x, y), g_max(g_max(tex_luma(x, y - 1), tex_luma(x - 1, y)), g_max(tex_luma(x, y + 1), tex_luma(x + 1, y)))))

    def is_low_contrast(x, y):
        lo, hi = contrast(x, y)
        rng = f32(hi - lo)
        return rng < g_max(f32(ONE / f32(32.0)), f32(f32(hi * ONE) / f32(2.0)))

    def blur_3x3(x, y):
        s = fetch(x - 1, y - 1)
        for dx, dy in ((0, -1), (1, -1), (-1, 0), (0, 0), (1, 0), (-1, 1), (0, 1), (1, 1)):
            s = add4(s, fetch(x + dx, y + dy))
        k = f32(ONE / f32(9.0))
        return [f32(k * v) for v in s]

    def sub_pixel_aliasing(x, y):
        luma_l = f32(f32(0.25) * f32(f32(f32(tex_luma(x, y - 1) + tex_luma(x - 1, y)) + tex_luma(x + 1, y)) + tex_luma(x, y + 1)))
        range_l = f32(abs(f32(luma_l - tex_luma(x, y))))
        lo, hi = contrast(x, y)
        rng = f32(hi - lo)
        with np.errstate(all="ignore"):
            blend_l = f32(g_max(ZERO, f32(f32(range_l / rng) - ZERO)) * ONE)
        return g_min(f32(f32(7.0) / f32(8.0)), blend_l)

    original = fetch(0, 0)
    luma = [[tex_luma(-1, -1), tex_luma(0, -1), tex_luma(1, -1)], [tex_luma(-1, 0), tex_luma(0, 0), tex_luma(1, 0)], [tex_luma(-1, 1), tex_luma(0, 1), tex_luma(1, 1)]]
    q, h, o = f32(0.25), f32(0.5), f32(1.0)
    e = lambda a, b, c, ka, kb, kc: f32(abs(f32(f32(f32(ka * a) + f32(kb * b)) + f32(kc * c))))
    edge_vert = f32(f32(e(luma[0][0], luma[0][1], luma[0][2], q, f32(-0.5), q) + e(luma[1][0], luma[1][1], luma[1][2], h, f32(-1.0), h)) + e(luma[2][0], luma[2][1], luma[2][2], q, f32(-0.5), q))
    edge_horz = f32(f32(e(luma[0][0], luma[1][0], luma[2][0], q, f32(-0.5), q) + e(luma[0][1], luma[1][1], luma[2][1], h, f32(-1.0), h)) + e(luma[0][2], luma[1][2], luma[2][2], q, f32(-0.5), q))
    step = (1, 0) if edge_horz >= edge_vert else (0, 1)
    if is_low_contrast(0, 0): return original
    pos_n, pos_p = (-step[0], -step[1]), step
    color, pixel_count = list(original), ONE
    done_n = done_p = False
    luma_mcn = g_max(g_max(f32(abs(f32(luma[0][1] - luma[1][1]))), f32(abs(f32(luma[1][2] - luma[1][1])))), g_max(f32(abs(f32(luma[2][1] - luma[1][1]))), f32(abs(f32(luma[1][0] - luma[1][1])))))
    gradient = f32(abs(f32(luma_mcn - luma[1][1])))
    for _ in range(6):
        if not done_n:
            blur = blur_3x3(*pos_n)
            done_n = f32(abs(f32(luma_of(blur) - luma_mcn))) >= gradient
            color = add4(color, mix4(fetch(*pos_n), blur, sub_pixel_aliasing(*pos_n)))
            pixel_count = f32(pixel_count + ONE)
            pos_n = (pos_n[0] - step[0], pos_n[1] - step[1])
        elif not done_p:
            blur = blur_3x3(*pos_p)
            done_p = f32(abs(f32(luma_of(blur) - luma_mcn))) >= gradient
            color = add4(color, mix4(fetch(*pos_p), blur, sub_pixel_aliasing(*pos_p)))
            pixel_count = f32(pixel_count + ONE)
            pos_p = (pos_p[0] + step[0], pos_p[1] + step[1])
        else:
            break
    return [f32(v / pixel_count) for v in color]


def taa_pixel(cache, px, py):
    """cache: nine textures, newest first"""
    c = [cache[k].fetch(px, py) for k in range(1, 9)]
    minRGB, maxRGB = [ONE] * 4, [ZERO] * 4
    for i in range(3):
        for j in range(3):
            if f32(np.sqrt(f32(f32((i - 1) * (i - 1)) + f32((j - 1) * (j - 1))))) > f32(2.0): continue
            p = cache[0].fetch(px + i - 1, py + j - 1)
            minRGB = [g_min(a, b) for a, b in zip(minRGB, p)]
            maxRGB = [g_max(a, b) for a, b in zip(maxRGB, p)]
    out = cache[0].fetch(px, py)
    for k in range(8):
        out = add4(out, [g_min(g_max(v, lo), hi) for v, lo, hi in zip(c[k], minRGB, maxRGB)])
    return [f32(v / f32(9.0)) for v in out]


def frame(rng, W, H, kind):
    """a float frame with edges, gradients, noise and uncovered (alpha 0) pixels; values outside [0, 1] exercise the store's clamp"""
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.zeros((H, W, 4), np.float32)
    base[..., 0] = (xx > W // 2 + (yy // 3) % 3) * 0.8 + 0.1
    base[..., 1] = ((xx + yy) % 7 < 3) * 0.6 + 0.05 * (yy / H)
    base[..., 2] = np.clip(np.sin(xx * 0.9) * 0.5 + 0.5, 0, 1) if kind else (yy > H // 2) * 0.9
    base[..., :3] += rng.normal(0, 0.04, (H, W, 3))
    base[..., 3] = 1.0
    base[(xx - 3) ** 2 + (yy - 2) ** 2 < 6] = 0.0                       # a hole of background
    base[0, 0] = [1.7, -0.3, np.nan, 1.0]
    return base.astype(np.float32)


def to_tex(fr):
    H, W, _ = fr.shape
    t = Tex(W, H)
    for y in range(H):
        for x in range(W):
            t.store(x, y, [f32(v) for v in fr[H - 1 - y, x]])
    return t


def cases():
    rng = np.random.default_rng(20261004)
    out = {"fxaa": [], "taa": []}
    for W, H, kind in ((24, 14, 0), (17, 11, 1)):
        fr = frame(rng, W, H, kind)
        t = to_tex(fr)
        res = [[fxaa_pixel(t, x, y) for x in range(W)] for y in range(H - 1, -1, -1)]
        out["fxaa"].append({"width": W, "height": H, "frame": [bits(v) for v in fr.reshape(-1)], "out": [bits(v) for row in res for px in row for v in px]})
    W, H = 20, 12
    frames = [frame(rng, W, H, k % 2) for k in range(11)]                 # oldest first
    for upto in (1, 4, 9, 11):                                            # the history fills, then rolls
        newest_first = frames[:upto][::-1][:9]
        cache = [to_tex(f) for f in newest_first] + [Tex(W, H) for _ in range(9 - len(newest_first))]
        res = [[taa_pixel(cache, x, y) for x in range(W)] for y in range(H - 1, -1, -1)]
        out["taa"].append({"width": W, "height": H, "frames_newest_first": [[bits(v) for v in f.reshape(-1)] for f in newest_first],
                           "out": [bits(v) for row in res for px in row for v in px]})
    return out


if __name__ == "__main__":
    data = cases()
    print("fxaa %d frames, taa %d states" % (len(data["fxaa"]), len(data["taa"])))
    if "--check" in sys.argv:
        assert json.load(gzip.open(OUT, "rt")) == data, "tests/golden/aa_kat.json.gz is not what this script writes"
        print("matches", OUT)
    else:
        with gzip.GzipFile(OUT, "wb", mtime=0) as fh:
            fh.write(json.dumps(data, separators=(",", ":")).encode())
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
