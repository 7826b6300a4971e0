#!/usr/bin/env python3
"""Literal known-answer tables for include/flx_math.h and for the shader's noise(), computed WITHOUT the
oracle and without flx_math.h: 60-digit decimal arithmetic (Python's decimal module: Taylor series after argument reduction with
a 70-digit pi), one correct rounding to float32 at the end.  The output, tests/golden/math_kat.json, holds bit patterns only:
    {"sin": [[x_bits, y_bits], ...], "atan2": [[y_bits, x_bits, r_bits], ...], "noise": [[nx, ny, seed, random_seed, r0, r1, r2, r3], ...]}
tests/test_oracle_kat.py::test_math_literal_known_answers feeds the inputs to the oracle's routines and compares with the stored
outputs: a pin that does not pass through the code it pins.  Re-run only to add inputs; a change of flx_math.h must keep every
stored answer.

    python tests/analysis/make_math_kat.py            # rewrites tests/golden/math_kat.json and reports disagreements with the oracle"""
import json
import os
import sys
from decimal import Decimal, getcontext

import numpy as np

getcontext().prec = 70
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899")
f32 = np.float32


def bits(x):
    return int(np.array([x], np.float32).view(np.uint32)[0])


def from_bits(u):
    return np.array([u], np.uint32).view(np.float32)[0]


def round_f32(d):
    """Decimal -> the nearest float32 (ties do not occur for these transcendental values)"""
    c = f32(float(d))
    best = None
    for cand in (np.nextafter(c, f32(-np.inf)), c, np.nextafter(c, f32(np.inf))):
        if not np.isfinite(cand):
            continue
        err = abs(Decimal(float(cand)) - d)
        if best is None or err < best[0]:
            best = (err, cand)
    return best[1]


def dsin(x):
    x = x % (2 * PI)
    if x > PI:
        x -= 2 * PI
    term, total, n = x, x, 1
    while abs(term) > Decimal(10) ** -68:
        term = -term * x * x / ((2 * n) * (2 * n + 1))
        total += term
        n += 1
    return total


def dcos(x):
    return dsin(x + PI / 2)


def datan(x):
    if x < 0:
        return -datan(-x)
    if x > 1:
        return PI / 2 - datan(1 / x)
    k = 0
    while x > Decimal("0.1"):                       # atan(x) = 2 atan(x / (1 + sqrt(1 + x^2)))
        x = x / (1 + (1 + x * x).sqrt())
        k += 1
    term, total, n = x, x, 0
    while abs(term) > Decimal(10) ** -68:
        n += 1
        term = -term * x * x
        total += term / (2 * n + 1)
    return total * (2 ** k)


def datan2(y, x):
    if x > 0:
        return datan(y / x)
    if x < 0:
        return datan(y / x) + (PI if y >= 0 else -PI)
    return PI / 2 if y > 0 else (-PI / 2 if y < 0 else Decimal(0))


def dacos(x):
    return datan2((1 - x * x).sqrt(), x)


FUNCS = {
    "sin": (0, lambda x: dsin(x)), "cos": (1, lambda x: dcos(x)), "tan": (2, lambda x: dsin(x) / dcos(x)),
    "acos": (3, lambda x: dacos(x)), "exp": (5, lambda x: x.exp()), "tanh": (7, lambda x: ((2 * x).exp() - 1) / ((2 * x).exp() + 1)),
}


def inputs(name, rng):
    if name in ("sin", "cos", "tan"):
        # the RNG's arguments (d + 53..67 * k: up to a few hundred), small angles, multiples of pi/2 rounded to float, large ones
        xs = list(rng.uniform(-700, 700, 120)) + list(rng.uniform(-4, 4, 60)) + [0.0, 1e-8, -3e-5, 0.5, 1.0, 1.5707964, 3.1415927, -4.712389, 6.2831855, 100.0, 51471.0, 1000000.0, -1048576.0]
        if name == "tan":
            xs = [x for x in xs if abs(abs(float(f32(x))) % 3.141592653589793 - 1.5707963267948966) > 1e-3]
        return xs
    if name == "acos":
        return list(rng.uniform(-1, 1, 120)) + [0.0, 1.0, -1.0, 0.5, -0.5, 0.99999994, -0.99999994, 1e-7, 0.70710677]
    if name == "exp":
        return list(rng.uniform(-30, 30, 120)) + [0.0, 1.0, -1.0, 0.5, -10.5, 65535.0 / 65535.0, 11.0899, -0.00001]
    if name == "tanh":
        return list(rng.uniform(-6, 6, 120)) + [0.0, 0.5, -0.5, 1.0, 4.0, 8.5, -8.5, 1e-4]
    raise KeyError(name)


def main():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
    rng = np.random.default_rng(20261004)
    table, report = {}, {}
    for name, (_, fn) in FUNCS.items():
        rows = []
        for x in inputs(name, rng):
            xf = f32(x)
            y = round_f32(fn(Decimal(float(xf))))
            rows.append([bits(xf), bits(y)])
        table[name] = rows
    rows = []
    pairs = list(zip(rng.uniform(-3, 3, 100), rng.uniform(-3, 3, 100))) + [(0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (1.0, 1.0), (-1.0, -1.0), (1e-8, 1.0), (1.0, -1e-8), (0.5, -0.5)]
    for y, x in pairs:
        yf, xf = f32(y), f32(x)
        rows.append([bits(yf), bits(xf), bits(round_f32(datan2(Decimal(float(yf)), Decimal(float(xf)))))])
    table["atan2"] = rows
    # noise(n, seed) of fragment:119-121 with the uniform randomSeed: fract(sin(dot(n, (12.9898, 78.233)) + (53, 59, 61, 67) * (seed + randomSeed * PHI)) * 43758.5453) * 2 - 1,
    # every operation a float32 operation in the order the oracle and the kernels evaluate it, sin = the correctly rounded sine
    PHI = f32(1.61803398874989484820459)
    rows = []
    for _ in range(64):
        nx, ny = f32(rng.uniform(-1, 1)), f32(rng.uniform(-1, 1))
        seed = f32(rng.choice([0.0, 1.0, 2.0, 3.0]) + np.cos(float(rng.integers(0, 16))))
        rs = f32(rng.integers(0, 4))
        d = f32(nx * f32(12.9898)) + f32(ny * f32(78.233))
        k = f32(seed + f32(rs * PHI))
        out = []
        for c in (53.0, 59.0, 61.0, 67.0):
            arg = f32(d + f32(f32(c) * k))
            sv = round_f32(dsin(Decimal(float(arg))))
            x = f32(sv * f32(43758.5453))
            fr = f32(x - f32(np.floor(x)))
            out.append(bits(f32(f32(fr * f32(2.0)) - f32(1.0))))
        rows.append([bits(nx), bits(ny), bits(seed), bits(rs)] + out)
    table["noise"] = rows
    with open(os.path.join(ROOT, "tests", "golden", "math_kat.json"), "w") as fh:
        json.dump(table, fh, separators=(",", ":"))
    # how does today's oracle compare?  (informational; the test is what binds)
    import ctypes as C
    import flx_oracle
    lib = flx_oracle.lib()
    lib.flx_oracle_math.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32]
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    for name, (sel, _) in FUNCS.items():
        x = np.array([r[0] for r in table[name]], np.uint32).view(np.float32)
        got = np.empty_like(x)
        lib.flx_oracle_math(sel, fp(x), None, fp(got), x.size)
        want = np.array([r[1] for r in table[name]], np.uint32)
        report[name] = int((got.view(np.uint32) != want).sum())
    y = np.array([r[0] for r in table["atan2"]], np.uint32).view(np.float32)
    x = np.array([r[1] for r in table["atan2"]], np.uint32).view(np.float32)
    got = np.empty_like(x)
    lib.flx_oracle_math(4, fp(y), fp(x), fp(got), x.size)
    report["atan2"] = int((got.view(np.uint32) != np.array([r[2] for r in table["atan2"]], np.uint32)).sum())
    print({k: len(v) for k, v in table.items()}, "oracle disagrees on:", report)


if __name__ == "__main__":
    main()
