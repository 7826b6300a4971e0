#!/usr/bin/env python3
"""Literal known answers for WHOLE PIXELS: lightTrace with all its bounces and main() (shaders/pathtracer_fragment.glsl:464-646) written from the shader text and
run over the arrays the reference's own scene.js emits — every sample, every bounce, both walks of every bounce, the file-scope variables that live across samples
(renderId, renderOriginalId, originalRMEx, originalTPOx, glassFilter, firstRayLength, originalColor), main()'s averaging and its six outputs.

The pieces are the literal tables' own: forwardTrace / reservoirSample's arithmetic / noise / the G-buffer encoders from make_shading_kat.py, rayTracer / shadowTest
from make_walk_kat.py, the intersection routines from make_intersect_kat.py (one float32 operation per operation of the text; sin, cos, acos, tan, atan correctly
rounded from 70-digit arithmetic).  ONE input per pixel does not come from the text: the primary hit (hit.suv, transform, triangle), the primary ray's direction and
clipSpace.xy / clipSpace.z, which the shader receives from the vertex shader through the rasteriser's interpolation (implementation-defined there; pinned as a ray cast
here: SURVEY.md 8a P0) — they are taken from the oracle's primary visibility (flx_oracle_primary) and stored with the row.  fetchTexVal's arithmetic is
the text's; what texture() does with the coordinate is the sampler set-up (REPEAT, NEAREST, RGBA8: fetch_tex_val's docstring).  Uninitialised outputs start as zeros (pinned).  Writes tests/golden/pixel_kat.json.gz.   usage: make_pixel_kat.py [--check]"""
import ctypes as C
import gzip, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_shading_kat as K                                                   # noqa: E402
from make_shading_kat import (f32, ONE, ZERO, BIAS, INV_255, add, sub, mul, scale, neg, dot, length, normalize, mix, mix3, gmax, cross, matvec, distance, gsign,      # noqa: E402
                              clamp01, reflect, refract, fresnel, forward_trace, noise, f_acos, f_tan, normal_to_spherical, to4bit)
from make_math_kat import dcos, round_f32                                      # noqa: E402
from make_intersect_kat import bits                                            # noqa: E402
from make_walk_kat import Arrays, ray_tracer, shadow_test, NaNInBoxTest       # noqa: E402
from decimal import Decimal                                                    # noqa: E402
from flexlight_hip.scene_io import Scene                                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "pixel_kat.json.gz")
SQRT3 = f32(1.7320508075688772)
INV_256 = f32(0.00390625)


def gmin(x, y): return y if y < x else x
def g_floor(x): return f32(np.floor(x))
def g_fract(x): return f32(x - g_floor(x))


class Globals:                                                               # the fragment shader's file-scope variables (fragment:74-89)
    def __init__(self):
        self.firstRayLength, self.glassFilter, self.originalRMEx, self.originalTPOx = ONE, ZERO, ZERO, ZERO
        self.originalColor = [ZERO, ZERO, ZERO]                              # (uninitialised: pinned)
        self.renderId = [ZERO] * 4; self.renderOriginalId = [ZERO] * 4       # (out variables, uninitialised: pinned)


def reservoir_sample(A, G, lights, randomSeed, albedo, rme, origin, unitDirection, randomVec, N, smoothNormal, geometryOffset, dontFilter, i):
    """fragment:400-461"""
    localColor = [ZERO, ZERO, ZERO]
    reservoirLength = totalWeight = reservoirWeight = ZERO
    reservoirNum = 0
    reservoirLightDir = [ZERO, ZERO, ZERO]                                   # (uninitialised: pinned)
    lastRandom = noise(randomVec[2], randomVec[3], BIAS, randomSeed)[0:2]
    for j, lt in enumerate(lights):
        if lt[3] <= ZERO: continue
        reservoirLength = f32(reservoirLength + ONE)
        light = add(lt[0:3], scale(randomVec[0:3], lt[4]))
        dirv = sub(light, origin)
        colorForLight = forward_trace(albedo, rme, dirv, lt[3], N, neg(unitDirection))
        localColor = add(localColor, colorForLight)
        weight = length(colorForLight)
        totalWeight = f32(totalWeight + weight)
        if f32(abs(lastRandom[1]) * totalWeight) <= weight:
            reservoirNum, reservoirWeight, reservoirLightDir = j, weight, dirv
        lastRandom = noise(lastRandom[0], lastRandom[1], BIAS, randomSeed)[2:4]
    with np.errstate(all="ignore"):
        unitLightDir = normalize(reservoirLightDir)
        showColor = reservoirLength == ZERO or reservoirWeight == ZERO
        showShadow = bool(dot(smoothNormal, unitLightDir) <= BIAS)
    baseLuminance = [rme[2]] * 3
    if dontFilter or i == 0: G.renderId[3] = f32(f32((reservoirNum % 128) << 1) * INV_255)
    if showColor: return add(localColor, baseLuminance)
    if showShadow:
        if dontFilter or i == 0: G.renderId[3] = f32(G.renderId[3] + INV_255)
        return baseLuminance
    offsetTarget = add(origin, scale(smoothNormal, geometryOffset))
    shadowed, _ = shadow_test(A, offsetTarget, unitLightDir, length(reservoirLightDir))
    if shadowed:
        if dontFilter or i == 0: G.renderId[3] = f32(G.renderId[3] + INV_255)
        return baseLuminance
    return add(localColor, baseLuminance)


def fetch_tex_val(atlas, textureWidth, uv, texNum, defaultVal):
    """fragment:108-117.  texture() on the atlases is pinned as the reference sets its samplers up and as GL defines them where it does: REPEAT (the coordinate's
    fract), NEAREST (the texel floor(coordinate x size), the last one where the product rounds up to size), an RGBA8 texel as byte / 255"""
    if texNum == f32(-1.0): return list(defaultVal)
    tex, W, H = atlas
    tw = f32(textureWidth)
    atlasHeightFactor = f32(f32(W) / f32(H))
    mod_ = f32(texNum - f32(tw * f32(np.floor(f32(texNum / tw)))))
    cx = f32(f32(uv[0] + mod_) / tw)
    cy = f32(f32(f32(uv[1] + f32(np.floor(f32(texNum / tw)))) * atlasHeightFactor) / tw)
    fx, fy = f32(g_fract(cx) * f32(W)), f32(g_fract(cy) * f32(H))
    ix, iy = min(int(fx), W - 1), min(int(fy), H - 1)
    return [f32(f32(int(c)) / f32(255)) for c in tex[iy, ix, 0:3]]


def light_trace(A, attrs, lights, ambient, randomSeed, minImportancy, G, hit, dir0, camera, ndc, cosSampleN, bounces, atlases, textureWidth):
    """fragment:464-599; hit = (suv, transformId, triangleId)"""
    dontFilter = True
    finalColor = [ZERO, ZERO, ZERO]
    importancyFactor = [ONE, ONE, ONE]
    G.originalColor = [ONE, ONE, ONE]
    origin, unitDirection = list(camera), list(dir0)                         # Ray(camera, normalize(target - camera)): the primary ray (P0)
    lastHitPoint = list(camera)
    i = 0
    while i < bounces and length(mul(importancyFactor, G.originalColor)) >= f32(minImportancy * SQRT3):
        suv, tI, triId = hit
        fi = f32(i)
        rTI, sTI = A.rotation[tI], A.shift[tI]
        origin = add(scale(unitDirection, suv[0]), origin)
        uvw = [f32(f32(ONE - suv[1]) - suv[2]), suv[1], suv[2]]
        g = A.entries[triId]
        triangle = [matvec(rTI, g[0:3]), matvec(rTI, g[3:6]), matvec(rTI, g[6:9])]
        offsetRayTarget = sub(origin, sTI)
        geometryNormal = normalize(cross(sub(triangle[0], triangle[1]), sub(triangle[0], triangle[2])))
        diffs = [distance(offsetRayTarget, triangle[0]), distance(offsetRayTarget, triangle[1]), distance(offsetRayTarget, triangle[2])]
        t = attrs[triId]
        normals = [matvec(rTI, t[0:3]), matvec(rTI, t[3:6]), matvec(rTI, t[6:9])]
        smoothNormal = normalize(matvec(normals, uvw))
        angles = [f_acos(abs(dot(geometryNormal, n))) for n in normals]
        angleTan = [clamp01(f_tan(a)) for a in angles]
        geometryOffset = dot(mul(diffs, angleTan), uvw)
        # barycentric = mat3x2(t2.yzw, t3.xyz) * uvw: columns (t2.y, t2.z), (t2.w, t3.x), (t3.y, t3.z)
        bary = [f32(f32(f32(t[9] * uvw[0]) + f32(t[11] * uvw[1])) + f32(t[13] * uvw[2])), f32(f32(f32(t[10] * uvw[0]) + f32(t[12] * uvw[1])) + f32(t[14] * uvw[2]))]
        albedo = fetch_tex_val(atlases[0], textureWidth, bary, t[15], t[18:21])
        rme = fetch_tex_val(atlases[1], textureWidth, bary, t[16], t[21:24])
        tpo = fetch_tex_val(atlases[2], textureWidth, bary, t[17], t[24:27])
        unitDirection = normalize(sub(origin, lastHitPoint))
        signDir = gsign(dot(unitDirection, smoothNormal))
        smoothNormal = scale(smoothNormal, f32(-signDir))
        randomVec = noise(ndc[0], ndc[1], f32(fi + cosSampleN), randomSeed)
        randomSpheareVec = normalize(add(smoothNormal, normalize(randomVec[0:3])))
        BRDF = mix(ONE, f32(abs(dot(smoothNormal, unitDirection))), rme[1])
        roughnessBRDF = f32(rme[0] * BRDF)
        roughNormal = normalize(mix3(smoothNormal, randomSpheareVec, roughnessBRDF))
        H = normalize(sub(roughNormal, unitDirection))
        VdotH = gmax(dot(neg(unitDirection), H), ZERO)
        F0 = scale(albedo, BRDF)
        fr = fresnel(F0, VdotH)
        fresnelReflect = gmax(fr[0], gmax(fr[1], fr[2]))
        isSolid = bool(f32(tpo[0] * fresnelReflect) <= f32(abs(randomVec[3])))
        if dontFilter:
            G.originalTPOx = tpo[0]
            G.originalColor = mul(G.originalColor, albedo)
            G.originalRMEx = f32(G.originalRMEx + rme[0])
            sc = f32(2.0 ** -i)                                               # pow(2.0, -fi): a power of two
            cn = [normal_to_spherical(smoothNormal), rme[0], to4bit(rme[1], rme[2])]
            upd = [f32(sc * cn[0]), f32(sc * cn[1]), f32(sc * cn[2]), f32(sc * ZERO)]
            G.renderId = [f32(a + b) for a, b in zip(G.renderId, upd)]
            if i == 0: G.renderOriginalId = [f32(a + b) for a, b in zip(G.renderOriginalId, upd)]
            dontFilter = (rme[0] < f32(0.01) and isSolid) or not isSolid
            if isSolid and tpo[0] > f32(0.01):
                G.glassFilter = f32(G.glassFilter + ONE)
                dontFilter = False
        else:
            importancyFactor = mul(importancyFactor, albedo)
        if i == 1:
            with np.errstate(all="ignore"):
                G.firstRayLength = gmin(f32(length(sub(origin, lastHitPoint)) / length(sub(lastHitPoint, camera))), G.firstRayLength)
        localColor = reservoir_sample(A, G, lights, randomSeed, albedo, rme, origin, unitDirection, randomVec, scale(roughNormal, f32(-signDir)),
                                      scale(smoothNormal, f32(-signDir)), geometryOffset, dontFilter, i)
        finalColor = add(finalColor, mul(localColor, importancyFactor))
        if isSolid:
            unitDirection = normalize(mix3(reflect(unitDirection, smoothNormal), randomSpheareVec, roughnessBRDF))
        else:
            with np.errstate(all="ignore"):
                eta = mix(f32(ONE / tpo[2]), tpo[2], gmax(signDir, ZERO))
            unitDirection = normalize(mix3(refract(unitDirection, smoothNormal, eta), randomSpheareVec, roughnessBRDF))
        (nsuv, ntI, ntri), _ = ray_tracer(A, origin, unitDirection)
        hit = (nsuv, ntI, ntri)
        if ntri == -1: break
        lastHitPoint = list(origin)
        i += 1
    return add(finalColor, mul(importancyFactor, ambient))


def pixel(A, attrs, lights, ambient, randomSeed, minImportancy, samples, bounces, camera, hit, dir0, ndc, atlases, textureWidth):
    """main(), fragment:601-646: -> (colour without filter, then the five outputs with useFilter = 1)"""
    G = Globals()
    finalColor = [ZERO, ZERO, ZERO]
    for i in range(samples):
        cosSampleN = round_f32(dcos(Decimal(float(i))))
        finalColor = add(finalColor, light_trace(A, attrs, lights, ambient, randomSeed, minImportancy, G, hit, dir0, camera, ndc, cosSampleN, bounces, atlases, textureWidth))
    invSamples = f32(ONE / f32(samples))
    finalColor = scale(finalColor, invSamples)
    plain = mul(finalColor, G.originalColor) + [ONE]
    renderColor = [g_fract(finalColor[0]), g_fract(finalColor[1]), g_fract(finalColor[2]), ONE]
    renderColorIp = [f32(g_floor(finalColor[0]) * INV_256), f32(g_floor(finalColor[1]) * INV_256), f32(g_floor(finalColor[2]) * INV_256), G.glassFilter]
    renderOriginalColor = G.originalColor + [f32(gmin(G.originalRMEx, G.firstRayLength) + INV_255)]
    renderId = [G.renderId[0], G.renderId[1], G.renderId[2], f32(G.renderId[3] + INV_255)]
    renderOriginalId = [ZERO, ZERO, ZERO, f32(G.originalTPOx + INV_255)]
    return plain, renderColor, renderColorIp, renderOriginalColor, renderId, renderOriginalId


def oracle_primary(sc, params, px, py_gl):
    import flx_oracle
    L = flx_oracle.lib()
    F3 = C.c_float * 3
    L.flx_oracle_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, F3, C.POINTER(C.c_int), C.POINTER(C.c_int), F3]
    L.flx_oracle_primary.restype = None
    suv, d, ti, tri = F3(), F3(), C.c_int(), C.c_int()
    view = sc.view()
    L.flx_oracle_primary(C.byref(view), C.byref(params), px, py_gl, suv, C.byref(ti), C.byref(tri), d)
    return [f32(x) for x in suv], ti.value, tri.value, [f32(x) for x in d]


CASES = [("cornell_obj", 48, 27, 2, 3, 0.0), ("dragon", 24, 14, 2, 4, 1.0), ("dragon", 12, 8, 3, 6, 2.0), ("theater", 32, 18, 2, 3, 3.0), ("cornell", 20, 20, 2, 2, 1.0)]


def rows():
    data = []
    for name, W, H, spp, bounces, seed in CASES:
        sc = Scene.golden(name)
        A = Arrays(sc)
        attrs = [[f32(x) for x in row] for row in sc.arrays["attributes"].astype(np.float32).reshape(-1, 28)]
        lights = [[f32(x) for x in row] for row in sc.arrays["lights"].astype(np.float32).reshape(-1, 6)]
        p = sc.frame_params(width=W, height=H, samples=spp, max_reflections=bounces, use_filter=0)
        p.random_seed = seed
        atlases = []
        for key, arr in (("albedo", "atlasAlbedo"), ("pbr", "atlasPbr"), ("tpo", "atlasTpo")):
            w, h = sc.meta["atlas"][key]
            atlases.append((sc.arrays[arr].reshape(h, w, 4), w, h))
        ambient = [f32(x) for x in p.ambient]
        camera = [f32(x) for x in p.camera]
        out = []
        for py_gl in range(H):
            for px in range(W):
                suv, tI, tri, d = oracle_primary(sc, p, px, py_gl)
                if tri == -1: continue
                ndc = [f32(f32(f32(f32(f32(px) + f32(0.5)) / f32(W)) * f32(2.0)) - ONE), f32(f32(f32(f32(f32(py_gl) + f32(0.5)) / f32(H)) * f32(2.0)) - ONE)]
                try:
                    res = pixel(A, attrs, lights, ambient, f32(p.random_seed), f32(p.min_importancy), spp, bounces, camera, (suv, tI, tri), d, ndc, atlases, p.texture_width)
                except NaNInBoxTest:
                    continue
                out.append([px, py_gl, tI, tri] + [bits(x) for x in suv] + [bits(x) for x in d] + [bits(x) for part in res for x in part])
        data.append({"scene": name, "width": W, "height": H, "samples": spp, "bounces": bounces, "random_seed": seed, "rows": out})
    return data


if __name__ == "__main__":
    data = rows()
    for c in data:
        r = np.array([x[10:14] for x in c["rows"]], np.uint32).view(np.float32)
        print("%-12s %dx%d %d spp %d bounces: %d pixels, mean colour %s" % (c["scene"], c["width"], c["height"], c["samples"], c["bounces"], len(c["rows"]), r[:, :3].mean(axis=0)))
    if "--check" in sys.argv:
        assert json.load(gzip.open(OUT, "rt")) == data, "tests/golden/pixel_kat.json.gz is not what this script writes"
        print("matches", OUT)
    else:
        with gzip.GzipFile(OUT, "wb", mtime=0) as fh:
            fh.write(json.dumps(data, separators=(",", ":")).encode())
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
