'use strict';
/*
 * flx_oracle_js.js — the CPU oracle's per-pixel function in plain JavaScript (single thread, Node).
 * TEST INFRASTRUCTURE ONLY, like oracle/flx_oracle.c: nothing under web-ray-tracer_amd/ may use it.  It exists for
 * one number — BASELINE.json configs[0] names "the reference JS/CPU headless path", and the reference has no such
 * renderer (its only CPU ray code is modules/math.js:113-137) — so bench.py's `cpu_baseline` carries, beside the C
 * oracle on all host cores, THIS REPOSITORY'S restatement of the shader timed under Node on one thread
 * (`cpu_baseline.js`): what a JavaScript fallback of the reference's fragment program would cost.
 *
 * It follows oracle/flx_oracle.c function by function (which cites shaders/pathtracer_fragment.glsl at every
 * definition) for frames without filter and without temporal accumulation, and include/flx_math.h for the
 * transcendental routines.  Arithmetic: binary32 = Math.fround after every +, -, *, / and sqrt (a double operation
 * on two floats rounded once to float is the correctly rounded float operation), binary64 = plain JavaScript numbers;
 * no Math.sin / cos / ... — so the frame equals the C oracle's BIT FOR BIT (tests/test_oracle_js_cpu.py).
 *
 *   node oracle/js/flx_oracle_js.js scene.flxs[.gz] params.json [--out frame.f32] [--repeat N]
 * params.json: { width, height, camera[3], view_matrix[9], samples, max_reflections, min_importancy, ambient[3],
 *               random_seed, texture_width }   (use_filter = 0, is_temporal = 0)
 * prints one JSON line: { ms_per_frame, frames, width, height, samples, bounces, mray_per_s, counters }
 */
const fs = require('fs');
const zlib = require('zlib');

const f = Math.fround;
const PHI = f(1.61803398874989484820459), SQRT3 = f(1.7320508075688772), INV_PI = f(0.3183098861837907), INV_255 = f(0.00392156862745098);
const PI_F = f(3.141592653589793), BIAS = f(0.0000152587890625), POW32 = 4294967296.0, NEAR_VIEW_DEPTH = 0.5;
const NAN_F = NaN;

/* ---- include/flx_math.h -------------------------------------------------------------------------------------------- */
const fmin = (x, y) => (y < x) ? y : x;                    /* GLSL min: "y if y < x, otherwise x" */
const fmax = (x, y) => (x < y) ? y : x;
const fabs = x => Math.abs(x);
const fclamp = (x, lo, hi) => fmin(fmax(x, lo), hi);
const fsign = x => (x > 0) ? 1 : ((x < 0) ? -1 : 0);
const fmix = (a, b, t) => f(f(a * f(1 - t)) + f(b * t));
const fsqrt = x => f(Math.sqrt(x));
const ffloor = x => Math.floor(x);                         /* exact; NaN / inf / -0 as they are */
const ffract = x => f(x - ffloor(x));
const fmod = (x, y) => f(x - f(y * ffloor(f(x / y))));
function ksincos (r, useCos) {
  const z = r * r;
  let p = useCos ? 4.7794773323873852974e-14 : 0.0;
  p = p * z + (useCos ? -1.1470745597729724714e-11 : -7.6471637318198164759e-13);
  p = p * z + (useCos ? 2.0876756987868098979e-09 : 1.6059043836821614599e-10);
  p = p * z + (useCos ? -2.7557319223985890653e-07 : -2.5052108385441718775e-08);
  p = p * z + (useCos ? 2.4801587301587301587e-05 : 2.7557319223985890653e-06);
  p = p * z + (useCos ? -1.3888888888888888889e-03 : -1.9841269841269841270e-04);
  p = p * z + (useCos ? 4.1666666666666666667e-02 : 8.3333333333333333333e-03);
  p = p * z + (useCos ? -0.5 : -1.6666666666666666667e-01);
  const t = z * p;
  return useCos ? (1.0 + t) : (r + r * t);
}
function ksin (r) {
  const z = r * r;
  let p = -7.6471637318198164759e-13;
  p = p * z + 1.6059043836821614599e-10; p = p * z + -2.5052108385441718775e-08; p = p * z + 2.7557319223985890653e-06;
  p = p * z + -1.9841269841269841270e-04; p = p * z + 8.3333333333333333333e-03; p = p * z + -1.6666666666666666667e-01;
  return r + r * (z * p);
}
function kcos (r) {
  const z = r * r;
  let p = 4.7794773323873852974e-14;
  p = p * z + -1.1470745597729724714e-11; p = p * z + 2.0876756987868098979e-09; p = p * z + -2.7557319223985890653e-07;
  p = p * z + 2.4801587301587301587e-05; p = p * z + -1.3888888888888888889e-03; p = p * z + 4.1666666666666666667e-02;
  p = p * z + -0.5;
  return 1.0 + z * p;
}
let remR = 0.0;                                             /* flx_rem_pio2's reduced argument */
function remPio2 (x) {                                      /* -> quadrant, or -1 for NaN / inf / |x| > 2^20 */
  if (!(fabs(x) <= 1048576.0)) { remR = 0.0; return -1; }
  const kd = Math.floor(x * 0.63661977236758134308 + 0.5);
  remR = (x - kd * 1.5707963267341256142) - kd * 6.0771005065061922045e-11;
  return kd & 3;
}
function fsin (x) {
  const q = remPio2(x);
  if (q < 0) return NAN_F;
  let v = ksincos(remR, q & 1);
  if (q & 2) v = -v;
  return f(v);
}
function fcos (x) {
  const q = remPio2(x);
  if (q < 0) return NAN_F;
  let v = ksincos(remR, (q & 1) ^ 1);
  if ((q + 1) & 2) v = -v;
  return f(v);
}
function ftan (x) {
  const q = remPio2(x);
  if (q < 0) return NAN_F;
  const s = ksin(remR), c = kcos(remR);
  return f((q & 1) ? (-c / s) : (s / c));
}
function katan (z) {
  const w = z * z;
  let p = 1.0 / 29.0;
  p = -p * w + 1.0 / 27.0; p = -p * w + 1.0 / 25.0; p = -p * w + 1.0 / 23.0;
  p = -p * w + 1.0 / 21.0; p = -p * w + 1.0 / 19.0; p = -p * w + 1.0 / 17.0;
  p = -p * w + 1.0 / 15.0; p = -p * w + 1.0 / 13.0; p = -p * w + 1.0 / 11.0;
  p = -p * w + 1.0 / 9.0; p = -p * w + 1.0 / 7.0; p = -p * w + 1.0 / 5.0;
  p = -p * w + 1.0 / 3.0;
  return z - z * (w * p);
}
function atanPos (t) {
  const PIO2 = 1.57079632679489661923, PIO4 = 0.78539816339744830962;
  let inv = false;
  if (t > 1.0) { t = 1.0 / t; inv = true; }
  const a = (t > 0.41421356237309504880) ? PIO4 + katan((t - 1.0) / (t + 1.0)) : katan(t);
  return inv ? (PIO2 - a) : a;
}
function fatan2 (y, x) {
  if (x !== x || y !== y) return NAN_F;
  const ay = y < 0.0 ? -y : y, ax = x < 0.0 ? -x : x;
  let a;
  if (ax === 0.0 && ay === 0.0) a = 0.0;
  else if (ax >= ay) a = atanPos(ay / ax);
  else a = 1.57079632679489661923 - atanPos(ax / ay);
  if (a !== a) a = 0.78539816339744830962;
  if (x < 0.0) a = 3.14159265358979323846 - a;
  if (y < 0.0) a = -a;
  return f(a);
}
function facos (x) {
  if (x !== x) return NAN_F;
  if (x >= 1.0) return 0.0;
  if (x <= -1.0) return PI_F;
  return f(2.0 * atanPos(Math.sqrt((1.0 - x) / (1.0 + x))));
}
const fpow5 = x => { const x2 = f(x * x); const x4 = f(x2 * x2); return f(x4 * x); };
const exp2NegInt = i => (i < 126) ? Math.pow(2, -i) : 0.0;     /* exact powers of two */
function f2uint (x) {
  if (!(x > 0.0)) return 0;
  if (x >= 4294967296.0) return 0xffffffff;
  return Math.trunc(x) >>> 0;
}
function invert3x3 (m) {
  const a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], ff = m[5], g = m[6], h = m[7], i = m[8];
  const A = e * i - ff * h, B = -(d * i - ff * g), C = d * h - e * g;
  const det = a * A + b * B + c * C;
  const r = 1.0 / det;
  return [f(A * r), f(-(b * i - c * h) * r), f((b * ff - c * e) * r),
    f(B * r), f((a * i - c * g) * r), f(-(a * ff - c * d) * r),
    f(C * r), f(-(a * h - b * g) * r), f((a * e - b * d) * r)];
}

/* ---- vectors: plain arrays [x, y, z]; every component operation rounded to binary32 -------------------------------- */
const add3 = (a, b) => [f(a[0] + b[0]), f(a[1] + b[1]), f(a[2] + b[2])];
const sub3 = (a, b) => [f(a[0] - b[0]), f(a[1] - b[1]), f(a[2] - b[2])];
const mul3 = (a, b) => [f(a[0] * b[0]), f(a[1] * b[1]), f(a[2] * b[2])];
const div3 = (a, b) => [f(a[0] / b[0]), f(a[1] / b[1]), f(a[2] / b[2])];
const scale3 = (a, s) => [f(a[0] * s), f(a[1] * s), f(a[2] * s)];
const divs3 = (a, s) => [f(a[0] / s), f(a[1] / s), f(a[2] / s)];
const neg3 = a => [-a[0], -a[1], -a[2]];
const dot3 = (a, b) => f(f(f(a[0] * b[0]) + f(a[1] * b[1])) + f(a[2] * b[2]));
const cross3 = (a, b) => [f(f(a[1] * b[2]) - f(b[1] * a[2])), f(f(a[2] * b[0]) - f(b[2] * a[0])), f(f(a[0] * b[1]) - f(b[0] * a[1]))];
const length3 = a => fsqrt(dot3(a, a));
const normalize3 = a => divs3(a, length3(a));
const distance3 = (a, b) => length3(sub3(a, b));
const mix3 = (a, b, t) => [fmix(a[0], b[0], t), fmix(a[1], b[1], t), fmix(a[2], b[2], t)];
const m3mul = (m, v) => [                                  /* m: 9 numbers, columns c0 c1 c2 */
  f(f(f(m[0] * v[0]) + f(m[3] * v[1])) + f(m[6] * v[2])),
  f(f(f(m[1] * v[0]) + f(m[4] * v[1])) + f(m[7] * v[2])),
  f(f(f(m[2] * v[0]) + f(m[5] * v[1])) + f(m[8] * v[2]))];
const reflect3 = (I, N) => sub3(I, scale3(N, f(2.0 * dot3(N, I))));
function refract3 (I, N, eta) {
  const d = dot3(N, I);
  const k = f(1.0 - f(f(eta * eta) * f(1.0 - f(d * d))));
  if (k < 0.0) return [0, 0, 0];
  return sub3(scale3(I, eta), scale3(N, f(f(eta * d) + fsqrt(k))));
}

/* ---- the renderer ---------------------------------------------------------------------------------------------------- */
function Renderer (scene, fp) {
  const geometry = scene.arrays.geometry, attributes = scene.arrays.attributes, rotation = scene.arrays.rotation, shift = scene.arrays.shift;
  const lights = scene.arrays.lights;
  const nEntries = geometry.length / 12, nLights = lights.length / 6;
  const atlas = [scene.arrays.atlasAlbedo, scene.arrays.atlasPbr, scene.arrays.atlasTpo];
  const atlasDim = [scene.meta.atlas.albedo, scene.meta.atlas.pbr, scene.meta.atlas.tpo];
  const cnt = { primary_visits: 0, closest_visits: 0, shadow_visits: 0, closest_walks: 0, shadow_walks: 0, shades: 0, primary_hits: 0, atlas_texels: 0 };
  const rs = f(fp.random_seed), tw = f(fp.texture_width), ambient = fp.ambient.map(f), minImp = f(fp.min_importancy);
  /* shader globals (fragment:83-89) */
  let firstRayLength = 1, glassFilter = 0, originalRMEx = 0, originalTPOx = 0, originalColor = [0, 0, 0];
  let renderIdW = 0;                                        /* (the other G-buffer words reach no output of a frame without filter) */
  let ndcX = 0, ndcY = 0;

  const rotationAt = i => { const o = i * 12; return [rotation[o], rotation[o + 1], rotation[o + 2], rotation[o + 4], rotation[o + 5], rotation[o + 6], rotation[o + 8], rotation[o + 9], rotation[o + 10]]; };
  const shiftAt = i => [shift[i * 4], shift[i * 4 + 1], shift[i * 4 + 2]];

  function fetchTexVal (which, u, v, texNum, def) {         /* fragment:108-117 */
    if (texNum === -1.0) return def;
    const W = atlas[which] && atlas[which].length ? atlasDim[which][0] : 1, H = atlas[which] && atlas[which].length ? atlasDim[which][1] : 1;
    const atlasHeightFactor = f(W / H);
    const cx = f(f(u + fmod(texNum, tw)) / tw);
    const cy = f(f(f(v + ffloor(f(texNum / tw))) * atlasHeightFactor) / tw);
    cnt.atlas_texels++;
    if (!(atlas[which] && atlas[which].length)) return [0, 0, 0];
    const fx = f(ffract(cx) * W), fy = f(ffract(cy) * H);
    let ix = f2uint(fx), iy = f2uint(fy);
    if (ix >= W) ix = W - 1;
    if (iy >= H) iy = H - 1;
    const o = (iy * W + ix) * 4;
    const t = atlas[which];
    return [f(t[o] / 255.0), f(t[o + 1] / 255.0), f(t[o + 2] / 255.0)];
  }
  function noise (nx, ny, seed) {                           /* fragment:119-121 */
    const d = f(f(nx * f(12.9898)) + f(ny * f(78.233)));
    const k = f(seed + f(rs * PHI));
    const c = f(43758.5453);
    return [
      f(f(ffract(f(fsin(f(d + f(53.0 * k))) * c)) * 2.0) - 1.0),
      f(f(ffract(f(fsin(f(d + f(59.0 * k))) * c)) * 2.0) - 1.0),
      f(f(ffract(f(fsin(f(d + f(61.0 * k))) * c)) * 2.0) - 1.0),
      f(f(ffract(f(fsin(f(d + f(67.0 * k))) * c)) * 2.0) - 1.0)];
  }
  const ZERO = [0, 0, 0];
  function moellerTrumbore (a, b, c, o, dir, l) {           /* fragment:123-140 */
    const edge1 = sub3(b, a), edge2 = sub3(c, a);
    const pvec = cross3(dir, edge2);
    const det = dot3(edge1, pvec);
    if (fabs(det) < BIAS) return ZERO;
    const invDet = f(1.0 / det);
    const tvec = sub3(o, a);
    const u = f(dot3(tvec, pvec) * invDet);
    if (u < BIAS || u > 1.0) return ZERO;
    const qvec = cross3(tvec, edge1);
    const v = f(dot3(dir, qvec) * invDet);
    const uvSum = f(u + v);
    if (v < BIAS || uvSum > 1.0) return ZERO;
    const s = f(dot3(edge2, qvec) * invDet);
    if (s > l || s <= BIAS) return ZERO;
    return [s, u, v];
  }
  function moellerTrumboreCull (a, b, c, o, dir, l) {       /* fragment:143-158 */
    const edge1 = sub3(b, a), edge2 = sub3(c, a);
    const pvec = cross3(dir, edge2);
    const det = dot3(edge1, pvec);
    const invDet = f(1.0 / det);
    if (det < BIAS) return false;
    const tvec = sub3(o, a);
    const u = f(dot3(tvec, pvec) * invDet);
    if (u < BIAS || u > 1.0) return false;
    const qvec = cross3(tvec, edge1);
    const v = f(dot3(dir, qvec) * invDet);
    if (v < BIAS || f(u + v) > 1.0) return false;
    const s = f(dot3(edge2, qvec) * invDet);
    return (s <= l && s > BIAS);
  }
  function moellerTrumborePrimary (a, b, c, o, dir, l, viewDepthPerS) {      /* oracle/flx_oracle.c: primary visibility */
    const edge1 = sub3(b, a), edge2 = sub3(c, a);
    const pvec = cross3(dir, edge2);
    const det = dot3(edge1, pvec);
    if (!(det < 0.0)) return ZERO;
    const invDet = f(1.0 / det);
    const tvec = sub3(o, a);
    const u = f(dot3(tvec, pvec) * invDet);
    if (!(u >= 0.0 && u <= 1.0)) return ZERO;
    const qvec = cross3(tvec, edge1);
    const v = f(dot3(dir, qvec) * invDet);
    if (!(v >= 0.0 && f(u + v) <= 1.0)) return ZERO;
    const s = f(dot3(edge2, qvec) * invDet);
    if (!(s < l) || !(f(s * viewDepthPerS) >= NEAR_VIEW_DEPTH)) return ZERO;
    return [s, u, v];
  }
  function rayCuboid (l, o, dir, mn, mx) {                  /* fragment:161-167 */
    const v0 = div3(sub3(mn, o), dir), v1 = div3(sub3(mx, o), dir);
    const tmin = fmax(fmax(fmin(v0[0], v1[0]), fmin(v0[1], v1[1])), fmin(v0[2], v1[2]));
    const tmax = fmin(fmin(fmax(v0[0], v1[0]), fmax(v0[1], v1[1])), fmax(v0[2], v1[2]));
    return tmax >= fmax(tmin, BIAS) && tmin < l;
  }
  /* fragment:172-227; mode 1 = primary visibility.  -> { suv, transformId, triangleId } */
  function rayTracer (o, dir, mode, viewDepthPerS, counter) {
    let tO = o, tD = dir, cachedTI = 0;
    let suv = ZERO, hitTI = 0, hitTri = -1;
    let minLen = POW32;
    for (let i = 0; i < nEntries; i++) {
      const e = i * 12;
      cnt[counter]++;
      const tI = (geometry[e + 9] | 0) << 1;
      if (tI !== cachedTI) {
        const iI = tI + 1;
        const R = rotationAt(iI);
        cachedTI = tI;
        tO = m3mul(R, add3(o, shiftAt(iI)));
        tD = m3mul(R, dir);
      }
      const kind = geometry[e + 10];
      if (kind === 0.0) return { suv, transformId: hitTI, triangleId: hitTri };
      if (kind === 1.0) {
        if (!rayCuboid(minLen, tO, tD, [geometry[e], geometry[e + 1], geometry[e + 2]], [geometry[e + 3], geometry[e + 4], geometry[e + 5]])) i += geometry[e + 6] | 0;
      } else {
        const a = [geometry[e], geometry[e + 1], geometry[e + 2]], b = [geometry[e + 3], geometry[e + 4], geometry[e + 5]], c = [geometry[e + 6], geometry[e + 7], geometry[e + 8]];
        const hit = mode ? moellerTrumborePrimary(a, b, c, tO, tD, minLen, viewDepthPerS) : moellerTrumbore(a, b, c, tO, tD, minLen);
        if (hit[0] !== 0.0) { suv = hit; hitTI = tI; hitTri = i; minLen = hit[0]; }
      }
    }
    return { suv, transformId: hitTI, triangleId: hitTri };
  }
  function shadowTest (o, dir, l) {                          /* fragment:231-280 */
    let tO = o, tD = dir, cachedTI = 0;
    for (let i = 0; i < nEntries; i++) {
      const e = i * 12;
      cnt.shadow_visits++;
      const tI = (geometry[e + 9] | 0) << 1;
      if (tI !== cachedTI) {
        const iI = tI + 1;
        const R = rotationAt(iI);
        cachedTI = tI;
        tO = m3mul(R, add3(o, shiftAt(iI)));
        tD = normalize3(m3mul(R, dir));
      }
      const kind = geometry[e + 10];
      if (kind === 0.0) return false;
      if (kind === 1.0) {
        if (!rayCuboid(l, tO, tD, [geometry[e], geometry[e + 1], geometry[e + 2]], [geometry[e + 3], geometry[e + 4], geometry[e + 5]])) i += geometry[e + 6] | 0;
      } else if (moellerTrumboreCull([geometry[e], geometry[e + 1], geometry[e + 2]], [geometry[e + 3], geometry[e + 4], geometry[e + 5]], [geometry[e + 6], geometry[e + 7], geometry[e + 8]], tO, tD, l)) return true;
    }
    return false;
  }
  /* fragment:282-334 */
  function trowbridgeReitz (alpha, NdotH) {
    const numerator = f(alpha * alpha);
    const denom = f(f(f(NdotH * NdotH) * f(numerator - 1.0)) + 1.0);
    return f(numerator / fmax(f(f(PI_F * denom) * denom), BIAS));
  }
  function schlickBeckmann (alpha, NdotX) {
    const k = f(alpha * 0.5);
    let denominator = f(f(NdotX * f(1.0 - k)) + k);
    denominator = fmax(denominator, BIAS);
    return f(NdotX / denominator);
  }
  const smith = (alpha, NdotV, NdotL) => f(schlickBeckmann(alpha, NdotV) * schlickBeckmann(alpha, NdotL));
  function fresnel (F0, theta) {
    const p = fpow5(f(1.0 - theta));
    return [f(F0[0] + f(f(1.0 - F0[0]) * p)), f(F0[1] + f(f(1.0 - F0[1]) * p)), f(F0[2] + f(f(1.0 - F0[2]) * p))];
  }
  function forwardTrace (material, lightDir, strength, N, V) {
    const lenP1 = f(1.0 + length3(lightDir));
    const brightness = f(strength / f(lenP1 * lenP1));
    const L = normalize3(lightDir);
    const H = normalize3(add3(V, L));
    const VdotH = fmax(dot3(V, H), 0.0);
    const NdotL = fmax(dot3(N, L), 0.0);
    const NdotH = fmax(dot3(N, H), 0.0);
    const NdotV = fmax(dot3(N, V), 0.0);
    const alpha = f(material.rme[0] * material.rme[0]);
    const BRDF = fmix(1.0, NdotV, material.rme[1]);
    const F0 = scale3(material.albedo, BRDF);
    const Ks = fresnel(F0, VdotH);
    const oneMinusMetal = f(1.0 - material.rme[1]);
    const Kd = [f(f(1.0 - Ks[0]) * oneMinusMetal), f(f(1.0 - Ks[1]) * oneMinusMetal), f(f(1.0 - Ks[2]) * oneMinusMetal)];
    const lambert = scale3(material.albedo, INV_PI);
    const tr = trowbridgeReitz(alpha, NdotH);
    const sm = smith(alpha, NdotV, NdotL);
    const cookTorranceNumerator = scale3(scale3(Ks, tr), sm);
    let cookTorranceDenominator = f(f(4.0 * NdotV) * NdotL);
    cookTorranceDenominator = fmax(cookTorranceDenominator, BIAS);
    const cookTorrance = divs3(cookTorranceNumerator, cookTorranceDenominator);
    const radiance = add3(mul3(Kd, lambert), cookTorrance);
    return scale3(scale3(radiance, NdotL), brightness);
  }
  /* fragment:400-461 */
  function reservoirSample (material, rayO, rayD, randomVec, N, smoothNormal, geometryOffset, dontFilter, i) {
    let localColor = [0, 0, 0];
    let reservoirLength = 0, totalWeight = 0, reservoirNum = 0, reservoirWeight = 0;
    let reservoirLightDir = [0, 0, 0];
    const n0 = noise(randomVec[2], randomVec[3], BIAS);
    let lastRandomX = n0[0], lastRandomY = n0[1];
    for (let j = 0; j < nLights; j++) {
      const lo = j * 6;
      const strength = lights[lo + 3], variation = lights[lo + 4];
      if (strength <= 0.0) continue;
      reservoirLength = f(reservoirLength + 1.0);
      const light = add3([lights[lo], lights[lo + 1], lights[lo + 2]], scale3([randomVec[0], randomVec[1], randomVec[2]], variation));
      const dir = sub3(light, rayO);
      const colorForLight = forwardTrace(material, dir, strength, N, neg3(rayD));
      localColor = add3(localColor, colorForLight);
      const weight = length3(colorForLight);
      totalWeight = f(totalWeight + weight);
      if (f(fabs(lastRandomY) * totalWeight) <= weight) { reservoirNum = j; reservoirWeight = weight; reservoirLightDir = dir; }
      const n1 = noise(lastRandomX, lastRandomY, BIAS);
      lastRandomX = n1[2]; lastRandomY = n1[3];
    }
    const unitLightDir = normalize3(reservoirLightDir);
    const showColor = reservoirLength === 0.0 || reservoirWeight === 0.0;
    const showShadow = dot3(smoothNormal, unitLightDir) <= BIAS;
    const baseLuminance = [material.rme[2], material.rme[2], material.rme[2]];
    if (dontFilter || i === 0) renderIdW = f(((reservoirNum % 128) << 1) * INV_255);
    if (showColor) return add3(localColor, baseLuminance);
    if (showShadow) {
      if (dontFilter || i === 0) renderIdW = f(renderIdW + INV_255);
      return baseLuminance;
    }
    const offsetTarget = add3(rayO, scale3(smoothNormal, geometryOffset));
    cnt.shadow_walks++;
    if (shadowTest(offsetTarget, unitLightDir, length3(reservoirLightDir))) {
      if (dontFilter || i === 0) renderIdW = f(renderIdW + INV_255);
      return baseLuminance;
    }
    return add3(localColor, baseLuminance);
  }
  /* fragment:464-599 */
  function lightTrace (hit, dir0, camera, cosSampleN, bounces) {
    let dontFilter = true;
    let finalColor = [0, 0, 0];
    let importancyFactor = [1, 1, 1];
    originalColor = [1, 1, 1];
    let rayO = camera, rayD = dir0;
    let lastHitPoint = camera;
    const thr = f(minImp * SQRT3);
    for (let i = 0; i < bounces && length3(mul3(importancyFactor, originalColor)) >= thr; i++) {
      const fi = i;
      cnt.shades++;
      const rTI = rotationAt(hit.transformId);
      const sTI = shiftAt(hit.transformId);
      rayO = add3(scale3(rayD, hit.suv[0]), rayO);
      const uvw = [f(f(1.0 - hit.suv[1]) - hit.suv[2]), hit.suv[1], hit.suv[2]];
      const g = hit.triangleId * 12;
      const t0v = m3mul(rTI, [geometry[g], geometry[g + 1], geometry[g + 2]]);
      const t1v = m3mul(rTI, [geometry[g + 3], geometry[g + 4], geometry[g + 5]]);
      const t2v = m3mul(rTI, [geometry[g + 6], geometry[g + 7], geometry[g + 8]]);
      const offsetRayTarget = sub3(rayO, sTI);
      const geometryNormal = normalize3(cross3(sub3(t0v, t1v), sub3(t0v, t2v)));
      const diffs = [distance3(offsetRayTarget, t0v), distance3(offsetRayTarget, t1v), distance3(offsetRayTarget, t2v)];
      const t = hit.triangleId * 28;
      const A = attributes;
      const n0 = m3mul(rTI, [A[t], A[t + 1], A[t + 2]]);
      const n1 = m3mul(rTI, [A[t + 3], A[t + 4], A[t + 5]]);
      const n2 = m3mul(rTI, [A[t + 6], A[t + 7], A[t + 8]]);
      let smoothNormal = normalize3([
        f(f(f(n0[0] * uvw[0]) + f(n1[0] * uvw[1])) + f(n2[0] * uvw[2])),
        f(f(f(n0[1] * uvw[0]) + f(n1[1] * uvw[1])) + f(n2[1] * uvw[2])),
        f(f(f(n0[2] * uvw[0]) + f(n1[2] * uvw[1])) + f(n2[2] * uvw[2]))]);
      const angles = [facos(fabs(dot3(geometryNormal, n0))), facos(fabs(dot3(geometryNormal, n1))), facos(fabs(dot3(geometryNormal, n2)))];
      const angleTan = [fclamp(ftan(angles[0]), 0.0, 1.0), fclamp(ftan(angles[1]), 0.0, 1.0), fclamp(ftan(angles[2]), 0.0, 1.0)];
      const geometryOffset = dot3(mul3(diffs, angleTan), uvw);
      const bu = f(f(f(A[t + 9] * uvw[0]) + f(A[t + 11] * uvw[1])) + f(A[t + 13] * uvw[2]));
      const bv = f(f(f(A[t + 10] * uvw[0]) + f(A[t + 12] * uvw[1])) + f(A[t + 14] * uvw[2]));
      const material = {
        albedo: fetchTexVal(0, bu, bv, A[t + 15], [A[t + 18], A[t + 19], A[t + 20]]),
        rme: fetchTexVal(1, bu, bv, A[t + 16], [A[t + 21], A[t + 22], A[t + 23]]),
        tpo: fetchTexVal(2, bu, bv, A[t + 17], [A[t + 24], A[t + 25], A[t + 26]])
      };
      rayD = normalize3(sub3(rayO, lastHitPoint));
      const signDir = fsign(dot3(rayD, smoothNormal));
      smoothNormal = scale3(smoothNormal, -signDir);
      const randomVec = noise(ndcX, ndcY, f(fi + cosSampleN));
      const randomSpheareVec = normalize3(add3(smoothNormal, normalize3([randomVec[0], randomVec[1], randomVec[2]])));
      const BRDF = fmix(1.0, fabs(dot3(smoothNormal, rayD)), material.rme[1]);
      const roughnessBRDF = f(material.rme[0] * BRDF);
      const roughNormal = normalize3(mix3(smoothNormal, randomSpheareVec, roughnessBRDF));
      const H = normalize3(sub3(roughNormal, rayD));
      const VdotH = fmax(dot3(neg3(rayD), H), 0.0);
      const F0 = scale3(material.albedo, BRDF);
      const fr = fresnel(F0, VdotH);
      const fresnelReflect = fmax(fr[0], fmax(fr[1], fr[2]));
      const isSolid = f(material.tpo[0] * fresnelReflect) <= fabs(randomVec[3]);
      if (dontFilter) {
        originalTPOx = material.tpo[0];
        originalColor = mul3(originalColor, material.albedo);
        originalRMEx = f(originalRMEx + material.rme[0]);
        dontFilter = (material.rme[0] < f(0.01) && isSolid) || !isSolid;
        if (isSolid && material.tpo[0] > f(0.01)) { glassFilter = f(glassFilter + 1.0); dontFilter = false; }
      } else {
        importancyFactor = mul3(importancyFactor, material.albedo);
      }
      if (i === 1) firstRayLength = fmin(f(length3(sub3(rayO, lastHitPoint)) / length3(sub3(lastHitPoint, camera))), firstRayLength);
      const localColor = reservoirSample(material, rayO, rayD, randomVec, scale3(roughNormal, -signDir), scale3(smoothNormal, -signDir), geometryOffset, dontFilter, i);
      finalColor = add3(finalColor, mul3(localColor, importancyFactor));
      if (isSolid) {
        rayD = normalize3(mix3(reflect3(rayD, smoothNormal), randomSpheareVec, roughnessBRDF));
      } else {
        const eta = fmix(f(1.0 / material.tpo[2]), material.tpo[2], fmax(signDir, 0.0));
        rayD = normalize3(mix3(refract3(rayD, smoothNormal, eta), randomSpheareVec, roughnessBRDF));
      }
      /* (the walk of fragment:591 whose result the loop guard discards is not made: oracle/flx_oracle.c, g_as_written = 0) */
      if (!(i + 1 < bounces && length3(mul3(importancyFactor, originalColor)) >= thr)) break;
      cnt.closest_walks++;
      hit = rayTracer(rayO, rayD, 0, 0.0, 'closest_visits');
      if (hit.triangleId === -1) break;
      lastHitPoint = rayO;
    }
    return add3(finalColor, mul3(importancyFactor, ambient));
  }

  const invView = invert3x3(fp.view_matrix.map(f));
  const viewRow2 = [f(fp.view_matrix[6]), f(fp.view_matrix[7]), f(fp.view_matrix[8])];
  const camera = fp.camera.map(f);
  const W = fp.width, H = fp.height;
  /* fragment:601-646 for a frame without filter / temporal: rows top-down as flx_render returns them */
  this.render = function (out) {
    const invSamples = f(1.0 / fp.samples);
    for (let row = 0; row < H; row++) {
      const pyGl = H - 1 - row;
      for (let px = 0; px < W; px++) {
        firstRayLength = 1; glassFilter = 0; originalRMEx = 0; originalTPOx = 0; originalColor = [0, 0, 0]; renderIdW = 0;
        const nx = f(f(f(f(px + 0.5) / W) * 2.0) - 1.0), ny = f(f(f(f(pyGl + 0.5) / H) * 2.0) - 1.0);
        ndcX = nx; ndcY = ny;
        let d = [f(f(f(invView[0] * nx) + f(invView[1] * ny)) + invView[2]), f(f(f(invView[3] * nx) + f(invView[4] * ny)) + invView[5]),
          f(f(f(invView[6] * nx) + f(invView[7] * ny)) + invView[8])];
        d = normalize3(d);
        const viewDepthPerS = dot3(viewRow2, d);
        const hit = rayTracer(camera, d, 1, viewDepthPerS, 'primary_visits');
        const o = (row * W + px) * 4;
        if (hit.triangleId === -1) { out[o] = out[o + 1] = out[o + 2] = out[o + 3] = 0; continue; }
        cnt.primary_hits++;
        let finalColor = [0, 0, 0];
        for (let s = 0; s < fp.samples; s++) finalColor = add3(finalColor, lightTrace(hit, d, camera, fcos(s), fp.max_reflections));
        finalColor = mul3(scale3(finalColor, invSamples), originalColor);
        out[o] = finalColor[0]; out[o + 1] = finalColor[1]; out[o + 2] = finalColor[2]; out[o + 3] = 1.0;
      }
    }
    return cnt;
  };
}

function readFlxs (file) {
  let blob = fs.readFileSync(file);
  if (file.endsWith('.gz')) blob = zlib.gunzipSync(blob);
  if (blob.slice(0, 6).toString('latin1') !== 'FLXS1\n') throw new Error('not a .flxs file: ' + file);
  const jsonLen = blob.readUInt32LE(6);
  const desc = JSON.parse(blob.slice(10, 10 + jsonLen).toString('utf8'));
  const base = (10 + jsonLen + 15) & ~15;
  const arrays = {};
  desc.arrays.forEach(e => {
    const width = e.dtype === 'u8' ? 1 : 4;
    const copy = new Uint8Array(e.count * width);
    blob.copy(Buffer.from(copy.buffer), 0, base + e.offset, base + e.offset + e.count * width);
    arrays[e.name] = e.dtype === 'f32' ? new Float32Array(copy.buffer) : e.dtype === 'i32' ? new Int32Array(copy.buffer) : copy;
  });
  return { meta: desc.meta, arrays };
}

if (require.main === module) {
  const args = process.argv.slice(2);
  const opt = (flag, d) => { const i = args.indexOf(flag); return i >= 0 ? args[i + 1] : d; };
  const scene = readFlxs(args[0]);
  const fp = JSON.parse(fs.readFileSync(args[1], 'utf8'));
  if (fp.use_filter || fp.is_temporal) throw new Error('frames without filter and temporal accumulation only');
  const repeat = Number(opt('--repeat', 1));
  const out = new Float32Array(fp.width * fp.height * 4);
  let cnt = null;
  const t0 = process.hrtime.bigint();
  for (let r = 0; r < repeat; r++) cnt = new Renderer(scene, fp).render(out);
  const ms = Number(process.hrtime.bigint() - t0) / 1e6 / repeat;
  if (opt('--out', null)) fs.writeFileSync(opt('--out'), Buffer.from(out.buffer));
  console.log(JSON.stringify({ ms_per_frame: ms, frames: repeat, width: fp.width, height: fp.height, samples: fp.samples, bounces: fp.max_reflections,
    mray_per_s: fp.samples * fp.max_reflections * fp.width * fp.height / (ms * 1e-3) / 1e6, counters: cnt, node: process.version }));
}

module.exports = { Renderer, readFlxs };
