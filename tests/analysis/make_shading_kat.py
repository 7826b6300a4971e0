#!/usr/bin/env python3
"""Literal known-answer tables for the SHADING half of the path (round 3): forwardTrace with its GGX / Smith /
Schlick helpers (shaders/pathtracer_fragment.glsl:282-334) and reservoirSample (:400-461), evaluated here from the SHADER TEXT —
not through include/flx_math.h and not through the oracle's C — and stored as bit patterns in tests/golden/shading_kat.json.

How a GLSL expression is evaluated here: every arithmetic operation of the shader is one IEEE float32 operation (numpy float32 scalars:
+ - * / and sqrt are correctly rounded), in the order the shader's text gives; dot(a, b) is (a.x b.x + a.y b.y) + a.z b.z,
length = sqrt(dot), normalize = a / length, mix(x, y, a) = x (1 - a) + y a (the GLSL ES 3.00 definitions); the ONE transcendental of
this code, the sine inside noise(), is the correctly rounded sine computed in 70-digit decimal arithmetic (make_math_kat.dsin);
pow(1 - theta, 5.0) of fresnel() is the product ((x x)(x x)) x, the pin DESIGN.md documents (GLSL leaves pow's precision open).

tests/test_oracle_kat.py::test_shading_literal_known_answers feeds the stored inputs to the oracle (flx_oracle_forward_trace,
flx_oracle_reservoir_sample) and requires the stored bits.  Since every GPU frame equals the oracle's bit for bit, the table holds
the kernels as well.  Re-run only to add inputs.

    python tests/analysis/make_shading_kat.py        # rewrites tests/golden/shading_kat.json, reports disagreements with today's oracle"""
import json
import os
import sys
from decimal import Decimal

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_math_kat import ROOT, bits, dsin, round_f32      # noqa: E402  (70-digit sine, one rounding to float32)

f32 = np.float32
PI = f32(3.141592653589793)
PHI = f32(1.61803398874989484820459)
BIAS = f32(0.0000152587890625)
INV_PI = f32(0.3183098861837907)
INV_255 = f32(0.00392156862745098)
ONE, ZERO = f32(1.0), f32(0.0)


# ---- GLSL built-ins on float32 scalars / 3-vectors (lists of float32) ----
def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def length(a):
    return f32(np.sqrt(dot(a, a)))


def normalize(a):
    n = length(a)
    with np.errstate(divide="ignore", invalid="ignore"):
        return [f32(x / n) for x in a]


def gmax(x, y):                      # GLSL max(x, y) = y if x < y else x
    return y if x < y else x


def mix(x, y, a):
    return f32(f32(x * f32(ONE - a)) + f32(y * a))


def add(a, b): return [f32(x + y) for x, y in zip(a, b)]
def sub(a, b): return [f32(x - y) for x, y in zip(a, b)]
def mul(a, b): return [f32(x * y) for x, y in zip(a, b)]
def scale(a, s): return [f32(x * s) for x in a]
def neg(a): return [f32(-x) for x in a]


# ---- fragment:282-300 ----
def trowbridge_reitz(alpha, NdotH):
    numerator = f32(alpha * alpha)
    denom = f32(f32(f32(NdotH * NdotH) * f32(numerator - ONE)) + ONE)
    return f32(numerator / gmax(f32(f32(PI * denom) * denom), BIAS))


def schlick_beckmann(alpha, NdotX):
    k = f32(alpha * f32(0.5))
    denominator = f32(f32(NdotX * f32(ONE - k)) + k)
    denominator = gmax(denominator, BIAS)
    return f32(NdotX / denominator)


def smith(alpha, NdotV, NdotL):
    return f32(schlick_beckmann(alpha, NdotV) * schlick_beckmann(alpha, NdotL))


def fresnel(F0, theta):
    x = f32(ONE - theta)
    x2 = f32(x * x)
    p = f32(f32(x2 * x2) * x)          # pow(1 - theta, 5.0)
    return [f32(f + f32(f32(ONE - f) * p)) for f in F0]


# ---- fragment:304-334 ----
def forward_trace(albedo, rme, lightDir, strength, N, V):
    lenP1 = f32(ONE + length(lightDir))
    brightness = f32(strength / f32(lenP1 * lenP1))
    L = normalize(lightDir)
    H = normalize(add(V, L))
    VdotH = gmax(dot(V, H), ZERO)
    NdotL = gmax(dot(N, L), ZERO)
    NdotH = gmax(dot(N, H), ZERO)
    NdotV = gmax(dot(N, V), ZERO)
    alpha = f32(rme[0] * rme[0])
    BRDF = mix(ONE, NdotV, rme[1])
    F0 = scale(albedo, BRDF)
    Ks = fresnel(F0, VdotH)
    Kd = [f32(f32(ONE - k) * f32(ONE - rme[1])) for k in Ks]
    lambert = scale(albedo, INV_PI)
    tr, sm = trowbridge_reitz(alpha, NdotH), smith(alpha, NdotV, NdotL)
    cookTorranceNumerator = scale(scale(Ks, tr), sm)
    cookTorranceDenominator = f32(f32(f32(4.0) * NdotV) * NdotL)
    cookTorranceDenominator = gmax(cookTorranceDenominator, BIAS)
    cookTorrance = [f32(x / cookTorranceDenominator) for x in cookTorranceNumerator]
    radiance = add(mul(Kd, lambert), cookTorrance)
    return scale(scale(radiance, NdotL), brightness)


# ---- fragment:119-121 ----
def noise(nx, ny, seed, randomSeed):
    d = f32(f32(nx * f32(12.9898)) + f32(ny * f32(78.233)))
    k = f32(seed + f32(randomSeed * PHI))
    out = []
    for c in (53.0, 59.0, 61.0, 67.0):
        arg = f32(d + f32(f32(c) * k))
        sv = round_f32(dsin(Decimal(float(arg))))
        x = f32(sv * f32(43758.5453))
        fr = f32(x - f32(np.floor(x)))
        out.append(f32(f32(fr * f32(2.0)) - ONE))
    return out


# ---- fragment:400-461 with an empty scene behind the lights (shadowTest finds nothing) ----
def reservoir_sample(lights, randomSeed, albedo, rme, origin, unitDirection, randomVec, N, smoothNormal, geometryOffset, dontFilter, i):
    localColor = [ZERO, ZERO, ZERO]
    reservoirLength = ZERO
    totalWeight = ZERO
    reservoirNum = 0
    reservoirWeight = ZERO
    reservoirLightDir = [ZERO, ZERO, ZERO]                # (uninitialised in the shader; pinned to 0, as in oracle and kernels)
    lastRandom = noise(randomVec[2], randomVec[3], BIAS, randomSeed)[0:2]
    for j, lt in enumerate(lights):
        strength, variation = lt[3], lt[4]
        if strength <= ZERO:
            continue
        reservoirLength = f32(reservoirLength + ONE)
        light = add(lt[0:3], scale(randomVec[0:3], variation))
        dirv = sub(light, origin)
        colorForLight = forward_trace(albedo, rme, dirv, strength, N, neg(unitDirection))
        localColor = add(localColor, colorForLight)
        weight = length(colorForLight)
        totalWeight = f32(totalWeight + weight)
        if f32(abs(lastRandom[1]) * totalWeight) <= weight:
            reservoirNum = j
            reservoirWeight = weight
            reservoirLightDir = dirv
        lastRandom = noise(lastRandom[0], lastRandom[1], BIAS, randomSeed)[2:4]
    unitLightDir = normalize(reservoirLightDir)
    showColor = reservoirLength == ZERO or reservoirWeight == ZERO
    with np.errstate(invalid="ignore"):
        showShadow = bool(dot(smoothNormal, unitLightDir) <= BIAS)
    baseLuminance = [rme[2], rme[2], rme[2]]
    renderIdW = ZERO
    if dontFilter or i == 0:
        renderIdW = f32(f32((reservoirNum % 128) << 1) * INV_255)
    if showColor:
        return add(localColor, baseLuminance) + [renderIdW]
    if showShadow:
        if dontFilter or i == 0:
            renderIdW = f32(renderIdW + INV_255)
        return baseLuminance + [renderIdW]
    return add(localColor, baseLuminance) + [renderIdW]      # the shadow ray meets nothing


def unit(rng):
    v = rng.normal(size=3)
    return [f32(x) for x in v / np.linalg.norm(v)]


def main():
    rng = np.random.default_rng(20261005)
    table = {"forward_trace": [], "reservoir": []}
    for k in range(160):
        albedo = [f32(x) for x in rng.uniform(0, 1, 3)]
        rme = [f32(rng.uniform(0, 1)), f32(rng.choice([0.0, 1.0, rng.uniform(0, 1)])), f32(rng.uniform(0, 0.3))]
        N = unit(rng)
        V = unit(rng)
        if k % 3 == 0:                                    # the common case: V and the light on N's side
            V = normalize(add(V, scale(N, f32(1.5))))
        lightDir = [f32(x) for x in rng.normal(size=3) * rng.choice([0.05, 1.0, 7.0, 300.0])]
        if k % 4 == 0:
            lightDir = add(lightDir, scale(N, f32(3.0)))
        strength = f32(rng.choice([0.5, 3.0, 80.0, 1000.0]))
        if k == 7:
            rme[0] = ZERO                                  # mirror: alpha = 0
        if k == 11:
            lightDir = scale(N, f32(2.0))                  # light along the normal
        if k == 13:
            V = neg(N)                                     # grazing from behind: NdotV clamps to 0, the denominators hit BIAS
        out = forward_trace(albedo, rme, lightDir, strength, N, V)
        table["forward_trace"].append([bits(x) for x in albedo + rme + lightDir + [strength] + N + V + out])
    for k in range(96):
        n_lights = int(rng.choice([0, 1, 2, 2, 3, 9]))
        lights = []
        for _ in range(n_lights):
            pos = rng.uniform(-8, 8, 3)
            lights.append([f32(pos[0]), f32(pos[1]), f32(pos[2]), f32(rng.choice([0.0, -1.0, 40.0, 120.0, 900.0])), f32(rng.choice([0.0, 0.1, 0.7])), ZERO])
        albedo = [f32(x) for x in rng.uniform(0, 1, 3)]
        rme = [f32(rng.uniform(0.02, 1)), f32(rng.uniform(0, 1)), f32(rng.choice([0.0, 0.05, 0.5]))]
        origin = [f32(x) for x in rng.uniform(-2, 2, 3)]
        N = unit(rng)
        smoothNormal = normalize(add(N, [f32(x) for x in rng.normal(size=3) * 0.1]))
        unitDirection = neg(normalize(add(unit(rng), scale(N, f32(1.2)))))       # arriving against N, mostly
        randomVec = [f32(x) for x in rng.uniform(-1, 1, 4)]
        randomSeed = f32(rng.integers(0, 4))
        geometryOffset = f32(rng.choice([0.0, 2.0 ** -10, 0.01]))
        dontFilter, i = int(rng.integers(0, 2)), int(rng.integers(0, 3))
        out = reservoir_sample(lights, randomSeed, albedo, rme, origin, unitDirection, randomVec, N, smoothNormal, geometryOffset, dontFilter, i)
        table["reservoir"].append({
            "lights": [[bits(x) for x in lt] for lt in lights], "random_seed": bits(randomSeed),
            "in": [bits(x) for x in albedo + rme + origin + unitDirection + randomVec + N + smoothNormal + [geometryOffset]],
            "dont_filter": dontFilter, "i": i, "out": [bits(x) for x in out]})
    with open(os.path.join(ROOT, "tests", "golden", "shading_kat.json"), "w") as fh:
        json.dump(table, fh, separators=(",", ":"))
    # how does today's oracle compare?  (informational; the test is what binds)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from shading_kat_util import oracle_forward_trace, oracle_reservoir
    bad_ft = sum(1 for row in table["forward_trace"] if oracle_forward_trace(row[:16]) != row[16:])
    bad_rs = sum(1 for row in table["reservoir"] if oracle_reservoir(row) != row["out"])
    print({k: len(v) for k, v in table.items()}, "oracle disagrees on: forward_trace %d, reservoir %d" % (bad_ft, bad_rs))


if __name__ == "__main__":
    main()
