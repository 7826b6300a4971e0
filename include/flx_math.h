/*
 * flx_math.h — exactly-defined scalar float routines for the FlexLight path tracer.
 *
 * Why this exists: the reference's RNG is fract(sin(x) * 43758.5453)
 * (shaders/pathtracer_fragment.glsl:119-121).  One ulp of sin() moves the random number by
 * ~4e-3 and can flip a reservoir pick or the solid/translucent choice (fragment:426,550), so
 * "the same picture" is only definable against an implementation that shares ONE sin().  The
 * GLSL built-ins the path uses (sin cos tan acos atan exp pow tanh) are implementation-defined
 * in GLSL ES 3.00, so this header pins them: every routine below is built only from IEEE-754
 * +,-,*,/ and sqrt on float/double plus exact int conversions, with no libm call and no fused
 * multiply-add unless written as flx_fmaf.  Compiled with -ffp-contract=off it produces the
 * same bits under gcc on x86-64 and under hipcc on gfx950; tests/test_math_parity.py checks
 * that on the GPU.
 *
 * Shared by the CPU oracle (oracle/) and the HIP kernels (web-ray-tracer_amd/csrc/): it is a
 * numerical definition, not a renderer.  Plain C99, also valid C++ / HIP.
 */
#ifndef FLX_MATH_H
#define FLX_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define FLX_HD __host__ __device__ static inline __attribute__((always_inline))
#else
#define FLX_HD static inline
#endif

#ifndef FLX_SINCOS_TABLE
#define FLX_SINCOS_TABLE 0      /* device code: sin / cos coefficients from a table instead of selects (same value; set per translation unit) */
#endif
#define FLX_PI_F 3.141592653589793f
#define FLX_BIAS 0.0000152587890625f          /* 2^-16, fragment:8 */
#define FLX_POW32 4294967296.0f               /* fragment:7 */

/* ---- bit casts ---------------------------------------------------------------------------- */
typedef union { float f; uint32_t u; } flx_f32_bits;
typedef union { double d; uint64_t u; } flx_f64_bits;

FLX_HD uint32_t flx_f2u(float f) { flx_f32_bits b; b.f = f; return b.u; }
FLX_HD float flx_u2f(uint32_t u) { flx_f32_bits b; b.u = u; return b.f; }
FLX_HD uint64_t flx_d2u(double d) { flx_f64_bits b; b.d = d; return b.u; }
FLX_HD double flx_u2d(uint64_t u) { flx_f64_bits b; b.u = u; return b.d; }

FLX_HD float flx_nanf(void) { return flx_u2f(0x7fc00000u); }
FLX_HD float flx_inff(void) { return flx_u2f(0x7f800000u); }
FLX_HD int flx_isnanf(float x) { return x != x; }

/* ---- GLSL ES 3.00 §8.3 common functions, spelled out ---------------------------------------- */
/* min(x,y): "y if y < x, otherwise x";  max(x,y): "y if x < y, otherwise x".  A NaN in x is
 * therefore returned, a NaN in y is dropped — frozen here because rayCuboid (fragment:161-167)
 * relies on what happens for 0/0. */
FLX_HD float flx_min(float x, float y) { return (y < x) ? y : x; }
FLX_HD float flx_max(float x, float y) { return (x < y) ? y : x; }
FLX_HD float flx_abs(float x) { return flx_u2f(flx_f2u(x) & 0x7fffffffu); }
FLX_HD float flx_clamp(float x, float lo, float hi) { return flx_min(flx_max(x, lo), hi); }
FLX_HD float flx_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
FLX_HD float flx_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

FLX_HD float flx_sqrt(float x) { return __builtin_sqrtf(x); }   /* IEEE correctly rounded on both sides */

/* floor for float without libm: exact. */
FLX_HD float flx_floor(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_floorf(x);                              /* v_floor_f32: floor is exact, so this is the value computed below (NaN and
                                                            * infinities returned as they are, -0 kept; kernels run with denormals on) */
#endif
  uint32_t u = flx_f2u(x);
  int e = (int)((u >> 23) & 0xffu) - 127;
  if (e >= 23) return x;                       /* integral, inf or nan */
  if (e < 0) {                                 /* |x| < 1 */
    if ((u & 0x7fffffffu) == 0u) return x;     /* +-0 */
    return (u >> 31) ? -1.0f : 0.0f;
  }
  uint32_t mask = 0x007fffffu >> e;
  if ((u & mask) == 0u) return x;
  if (u >> 31) u += mask;                      /* negative: round magnitude up */
  return flx_u2f(u & ~mask);
}
FLX_HD float flx_fract(float x) { return x - flx_floor(x); }
FLX_HD float flx_mod(float x, float y) { return x - y * flx_floor(x / y); }

FLX_HD double flx_floord(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_floor(x);                               /* v_floor_f64: floor is exact, so this is the value computed below */
#endif
  uint64_t u = flx_d2u(x);
  int e = (int)((u >> 52) & 0x7ffu) - 1023;
  if (e >= 52) return x;
  if (e < 0) {
    if ((u & 0x7fffffffffffffffull) == 0ull) return x;
    return (u >> 63) ? -1.0 : 0.0;
  }
  uint64_t mask = 0x000fffffffffffffull >> e;
  if ((u & mask) == 0ull) return x;
  if (u >> 63) u += mask;
  return flx_u2d(u & ~mask);
}

/* ---- sin / cos / tan ---------------------------------------------------------------------- */
/* Reduction in double: k = round(x * 2/pi), r = x - k*pi/2 with pi/2 split in two parts whose
 * first has 33 significant bits (k < 2^20 keeps k*hi exact).  Kernels: Taylor to r^15 / r^16 on
 * |r| <= pi/4 (truncation < 1e-15), Horner in double, one rounding to float at the end. */
FLX_HD double flx_ksin(double r) {
  double z = r * r;
  double p = -7.6471637318198164759e-13;                 /* -1/15! */
  p = p * z + 1.6059043836821614599e-10;                 /*  1/13! */
  p = p * z + -2.5052108385441718775e-08;                /* -1/11! */
  p = p * z + 2.7557319223985890653e-06;                 /*  1/9!  */
  p = p * z + -1.9841269841269841270e-04;                /* -1/7!  */
  p = p * z + 8.3333333333333333333e-03;                 /*  1/5!  */
  p = p * z + -1.6666666666666666667e-01;                /* -1/3!  */
  return r + r * (z * p);
}
FLX_HD double flx_kcos(double r) {
  double z = r * r;
  double p = 4.7794773323873852974e-14;                  /*  1/16! */
  p = p * z + -1.1470745597729724714e-11;                /* -1/14! */
  p = p * z + 2.0876756987868098979e-09;                 /*  1/12! */
  p = p * z + -2.7557319223985890653e-07;                /* -1/10! */
  p = p * z + 2.4801587301587301587e-05;                 /*  1/8!  */
  p = p * z + -1.3888888888888888889e-03;                /* -1/6!  */
  p = p * z + 4.1666666666666666667e-02;                 /*  1/4!  */
  p = p * z + -0.5;
  return 1.0 + z * p;
}
/* flx_ksin(r) or flx_kcos(r) — the same values bit for bit — through ONE Horner chain whose coefficients are selected:
 * a GPU lane pays for one polynomial instead of both branches of the quadrant test.  The sine chain is one step
 * shorter; it starts from 0, and 0 * z + c == c exactly. */
#if defined(__HIP_DEVICE_COMPILE__) && FLX_SINCOS_TABLE
/* the two coefficient sets as a table: five loads instead of eighteen 32-bit selects per call (the same coefficients in the same
 * order: the same value) */
static __device__ const double flx_sincos_tab[2][10] = {
  { 0.0, -7.6471637318198164759e-13, 1.6059043836821614599e-10, -2.5052108385441718775e-08, 2.7557319223985890653e-06,
    -1.9841269841269841270e-04, 8.3333333333333333333e-03, -1.6666666666666666667e-01, 0.0, 0.0 },
  { 4.7794773323873852974e-14, -1.1470745597729724714e-11, 2.0876756987868098979e-09, -2.7557319223985890653e-07, 2.4801587301587301587e-05,
    -1.3888888888888888889e-03, 4.1666666666666666667e-02, -0.5, 0.0, 0.0 } };
#endif
FLX_HD double flx_ksincos(double r, int use_cos) {
  double z = r * r;
#if defined(__HIP_DEVICE_COMPILE__) && FLX_SINCOS_TABLE
  {
    const double *c = flx_sincos_tab[use_cos ? 1 : 0];
    double q = c[0];
    q = q * z + c[1]; q = q * z + c[2]; q = q * z + c[3]; q = q * z + c[4]; q = q * z + c[5]; q = q * z + c[6]; q = q * z + c[7];
    const double u = z * q;
    return use_cos ? (1.0 + u) : (r + r * u);
  }
#endif
  double p = use_cos ? 4.7794773323873852974e-14 : 0.0;
  p = p * z + (use_cos ? -1.1470745597729724714e-11 : -7.6471637318198164759e-13);
  p = p * z + (use_cos ? 2.0876756987868098979e-09 : 1.6059043836821614599e-10);
  p = p * z + (use_cos ? -2.7557319223985890653e-07 : -2.5052108385441718775e-08);
  p = p * z + (use_cos ? 2.4801587301587301587e-05 : 2.7557319223985890653e-06);
  p = p * z + (use_cos ? -1.3888888888888888889e-03 : -1.9841269841269841270e-04);
  p = p * z + (use_cos ? 4.1666666666666666667e-02 : 8.3333333333333333333e-03);
  p = p * z + (use_cos ? -0.5 : -1.6666666666666666667e-01);
  double t = z * p;
  return use_cos ? (1.0 + t) : (r + r * t);
}
/* Returns quadrant (0..3) and reduced argument; *ok = 0 for NaN/Inf/|x| > 2^20 (result NaN). */
FLX_HD int flx_rem_pio2(float x, double *r, int *ok) {
  const double INV_PIO2 = 0.63661977236758134308;
  const double PIO2_HI = 1.5707963267341256142;          /* 33 bits of pi/2 */
  const double PIO2_LO = 6.0771005065061922045e-11;      /* pi/2 - PIO2_HI */
  if (!(flx_abs(x) <= 1048576.0f)) { *ok = 0; *r = 0.0; return 0; }
  double xd = (double)x;
  double kd = flx_floord(xd * INV_PIO2 + 0.5);
  *r = (xd - kd * PIO2_HI) - kd * PIO2_LO;
  *ok = 1;
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)kd & 3;                                      /* |kd| < 2^20: the two low bits of the 32-bit conversion (one instruction) are those of the 64-bit one */
#endif
  return (int)((long long)kd & 3ll);
}
FLX_HD float flx_sin(float x) {
  double r; int ok; int q = flx_rem_pio2(x, &r, &ok);
  if (!ok) return flx_nanf();
  double v = flx_ksincos(r, q & 1);
  if (q & 2) v = -v;
  return (float)v;
}
FLX_HD float flx_cos(float x) {
  double r; int ok; int q = flx_rem_pio2(x, &r, &ok);
  if (!ok) return flx_nanf();
  double v = flx_ksincos(r, (q & 1) ^ 1);
  if ((q + 1) & 2) v = -v;
  return (float)v;
}
FLX_HD float flx_tan(float x) {
  double r; int ok; int q = flx_rem_pio2(x, &r, &ok);
  if (!ok) return flx_nanf();
  double s = flx_ksin(r), c = flx_kcos(r);
  double v = (q & 1) ? (-c / s) : (s / c);
  return (float)v;
}

/* ---- atan / atan2 / acos ------------------------------------------------------------------ */
/* atan on [0,inf): fold x>1 to 1/x, fold t>tan(pi/8) around 1, odd Taylor series to z^29 on
 * |z| <= 0.4143 (truncation < 2e-13 relative). */
FLX_HD double flx_katan(double z) {
  double w = z * z;
  double p = 1.0 / 29.0;
  p = -p * w + 1.0 / 27.0;  p = -p * w + 1.0 / 25.0;  p = -p * w + 1.0 / 23.0;
  p = -p * w + 1.0 / 21.0;  p = -p * w + 1.0 / 19.0;  p = -p * w + 1.0 / 17.0;
  p = -p * w + 1.0 / 15.0;  p = -p * w + 1.0 / 13.0;  p = -p * w + 1.0 / 11.0;
  p = -p * w + 1.0 / 9.0;   p = -p * w + 1.0 / 7.0;   p = -p * w + 1.0 / 5.0;
  p = -p * w + 1.0 / 3.0;
  return z - z * (w * p);
}
FLX_HD double flx_atan_pos(double t) {                     /* t >= 0, finite or +inf */
  const double PIO2 = 1.57079632679489661923, PIO4 = 0.78539816339744830962;
  int inv = 0;
  if (t > 1.0) { t = 1.0 / t; inv = 1; }
  double a;
  if (t > 0.41421356237309504880) a = PIO4 + flx_katan((t - 1.0) / (t + 1.0));
  else a = flx_katan(t);
  return inv ? (PIO2 - a) : a;
}
/* atan(y, x) as GLSL's two-argument atan; (0,0) -> 0, NaN in -> NaN. */
FLX_HD float flx_atan2(float y, float x) {
  const double PI = 3.14159265358979323846;
  if (flx_isnanf(x) || flx_isnanf(y)) return flx_nanf();
  double yd = (double)y, xd = (double)x;
  double ay = yd < 0.0 ? -yd : yd, ax = xd < 0.0 ? -xd : xd;
  double a;
  if (ax == 0.0 && ay == 0.0) a = 0.0;
  else if (ax >= ay) a = flx_atan_pos(ay / ax);           /* ay/ax in [0,1]; inf/inf -> NaN guarded below */
  else a = 1.57079632679489661923 - flx_atan_pos(ax / ay);
  if (a != a) a = 0.78539816339744830962;                 /* inf/inf */
  if (xd < 0.0) a = PI - a;
  if (yd < 0.0) a = -a;
  return (float)a;
}
/* acos with the argument clamped to [-1,1]: the path calls it on |n_g . n_i| of unit vectors
 * (fragment:516), which rounding can push a hair over 1; GLSL leaves that undefined, we define 0. */
FLX_HD float flx_acos(float x) {
  if (flx_isnanf(x)) return flx_nanf();
  double xd = (double)x;
  if (xd >= 1.0) return 0.0f;
  if (xd <= -1.0) return FLX_PI_F;
  double t = __builtin_sqrt((1.0 - xd) / (1.0 + xd));
  return (float)(2.0 * flx_atan_pos(t));
}

/* ---- exp / log / pow / tanh ---------------------------------------------------------------- */
FLX_HD double flx_expd(double x) {
  const double INV_LN2 = 1.44269504088896340736;
  const double LN2_HI = 6.93147180369123816490e-01;       /* fdlibm split of ln 2 */
  const double LN2_LO = 1.90821492927058770002e-10;
  if (x != x) return x;
  if (x > 709.0) return flx_u2d(0x7ff0000000000000ull);
  if (x < -708.0) return 0.0;
  double kd = flx_floord(x * INV_LN2 + 0.5);
  double r = (x - kd * LN2_HI) - kd * LN2_LO;
  double p = 1.0 / 6227020800.0;                           /* 1/13! */
  p = p * r + 1.0 / 479001600.0;  p = p * r + 1.0 / 39916800.0;  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;     p = p * r + 1.0 / 40320.0;     p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;        p = p * r + 1.0 / 120.0;       p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;          p = p * r + 0.5;               p = p * r + 1.0;
  p = p * r + 1.0;
  long long k = (long long)kd;                             /* |k| <= 1023 here */
  return p * flx_u2d((uint64_t)(k + 1023ll) << 52);
}
FLX_HD double flx_logd(double x) {                         /* x > 0, finite, normal */
  const double LN2 = 0.69314718055994530942;
  uint64_t u = flx_d2u(x);
  long long e = (long long)((u >> 52) & 0x7ffull) - 1023ll;
  double m = flx_u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);   /* [1,2) */
  if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
  double f = (m - 1.0) / (m + 1.0);
  double w = f * f;
  double p = 1.0 / 23.0;
  p = p * w + 1.0 / 21.0;  p = p * w + 1.0 / 19.0;  p = p * w + 1.0 / 17.0;
  p = p * w + 1.0 / 15.0;  p = p * w + 1.0 / 13.0;  p = p * w + 1.0 / 11.0;
  p = p * w + 1.0 / 9.0;   p = p * w + 1.0 / 7.0;   p = p * w + 1.0 / 5.0;
  p = p * w + 1.0 / 3.0;
  return (double)e * LN2 + (2.0 * f + 2.0 * f * (w * p));
}
FLX_HD float flx_exp(float x) { return (float)flx_expd((double)x); }
/* pow(x,y) = exp(y ln x); x == 0 -> 0 for y > 0 (else inf), x < 0 -> NaN (GLSL: undefined). */
FLX_HD float flx_pow(float x, float y) {
  if (flx_isnanf(x) || flx_isnanf(y)) return flx_nanf();
  if (x < 0.0f) return flx_nanf();
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : flx_inff());
  if (x == flx_inff()) return (y > 0.0f) ? flx_inff() : ((y == 0.0f) ? 1.0f : 0.0f);
  double xd = (double)x;
  if (xd < 2.2250738585072014e-308) return 0.0f;           /* unreachable for float input */
  return (float)flx_expd((double)y * flx_logd(xd));
}
/* pow(1 - theta, 5.0) of fresnel (fragment:301) as an exact product chain. */
FLX_HD float flx_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
FLX_HD float flx_tanh(float x) {
  if (flx_isnanf(x)) return x;
  if (x > 20.0f) return 1.0f;
  if (x < -20.0f) return -1.0f;
  double e = flx_expd(2.0 * (double)x);
  return (float)((e - 1.0) / (e + 1.0));
}
/* pow(2.0, -i) for small non-negative integer i (fragment:560): exact. */
FLX_HD float flx_exp2_neg_int(int i) { return (i < 126) ? flx_u2f((uint32_t)(127 - i) << 23) : 0.0f; }


/* ---- small host-side helper shared by oracle and library --------------------------------------- */
/* Inverse of a row-major 3x3 (the viewMatrix), adjugate / determinant in double, rounded to float:
 * primary rays are V^-1 * (ndc.x, ndc.y, 1)  (SURVEY.md §8a row P0). */
FLX_HD void flx_invert3x3(const float m[9], float out[9]) {
  double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * B + c * C;
  double r = 1.0 / det;
  out[0] = (float)(A * r); out[1] = (float)(-(b * i - c * h) * r); out[2] = (float)((b * f - c * e) * r);
  out[3] = (float)(B * r); out[4] = (float)((a * i - c * g) * r);  out[5] = (float)(-(a * f - c * d) * r);
  out[6] = (float)(C * r); out[7] = (float)(-(a * h - b * g) * r); out[8] = (float)((a * e - b * d) * r);
}

/* float -> uint as GLSL's uint(x) for the in-range values the path produces, with the out-of-range
 * cases (undefined in GLSL) pinned: negative / NaN -> 0, too large -> 0xffffffff. */
FLX_HD uint32_t flx_f2uint(float x) {
  if (!(x > 0.0f)) return 0u;
  if (x >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)x;
}

#endif /* FLX_MATH_H */
