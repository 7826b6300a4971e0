/*
 * flx_oracle.c — CPU oracle: plain-C restatement of the reference's path-tracing shaders.
 * TEST INFRASTRUCTURE ONLY (see flx_oracle.h).  Build: oracle/Makefile (-O2 -ffp-contract=off).
 *
 * Follows, function by function (all paths relative to the reference checkout):
 *   shaders/pathtracer_fragment.glsl   every function; cited at each definition below
 *   shaders/pathtracer_vertex.glsl:40-72 + modules/pathtracerWGL2.js:372,712-718
 *                                      primary visibility, re-expressed as a ray cast
 *   modules/pathtracerWGL2.js:312-356  uniforms
 * GLSL built-ins are expanded per GLSL ES 3.00 §8 with the transcendental ones pinned by
 * include/flx_math.h.  Arithmetic is binary32 throughout, evaluated strictly left to right as
 * written; vector ops are component-wise; dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;
 * mat3 * vec3 = (col0*v.x + col1*v.y) + col2*v.z.
 *
 * Behaviour the GLSL leaves open and this file pins (also listed in DESIGN.md):
 *   - renderId / renderOriginalId start at 0 (fragment:562-563 accumulate into unwritten outputs)
 *   - texture(): NEAREST, REPEAT, texel = floor(fract(coord) * size) (gllib.js:67-70)
 *   - acos argument clamped to [-1,1]; uint(x) of negative/NaN = 0
 *   - primary visibility by ray cast: front faces only (det < 0), inclusive barycentrics, first
 *     entry wins ties (depth LESS), near plane at view depth 0.5, pixel centre NDC
 */
#include "flx_oracle.h"
#include "flx_math.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PHI 1.61803398874989484820459f      /* fragment:5 */
#define SQRT3 1.7320508075688772f           /* fragment:6 */
#define INV_PI 0.3183098861837907f          /* fragment:10 */
#define INV_256 0.00390625f                 /* fragment:11 */
#define INV_255 0.00392156862745098f        /* fragment:12 */
#define PI_F 3.141592653589793f             /* fragment:4 */
#define BIAS FLX_BIAS
#define POW32 FLX_POW32
#define NEAR_VIEW_DEPTH 0.5f                /* vertex:68 + clip volume, SURVEY §8a P0 */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { v3 c0, c1, c2; } m3;       /* columns, like GLSL mat3 */

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 div3(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divs3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
static inline float length3(v3 a) { return flx_sqrt(dot3(a, a)); }
static inline v3 normalize3(v3 a) { return divs3(a, length3(a)); }
static inline float distance3(v3 a, v3 b) { return length3(sub3(a, b)); }
static inline v3 mix3(v3 a, v3 b, float t) { return V3(flx_mix(a.x, b.x, t), flx_mix(a.y, b.y, t), flx_mix(a.z, b.z, t)); }
static inline v3 m3mul(m3 m, v3 v) {
  return V3((m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z,
            (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z,
            (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z);
}
/* GLSL reflect / refract, ES 3.00 §8.5 */
static inline v3 reflect3(v3 I, v3 N) { return sub3(I, scale3(N, 2.0f * dot3(N, I))); }
static inline v3 refract3(v3 I, v3 N, float eta) {
  float d = dot3(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
  return sub3(scale3(I, eta), scale3(N, eta * d + flx_sqrt(k)));
}

typedef struct { v3 origin, unitDirection; } Ray;              /* fragment:19-22 */
typedef struct { v3 albedo, rme, tpo; } Material;              /* fragment:24-28 */
typedef struct { v3 suv; int transformId; int triangleId; } Hit;   /* fragment:30-34 */

/* Everything a "fragment invocation" sees: uniforms, textures and the shader's global variables. */
typedef struct {
  const flx_scene_view *sc;
  const flx_frame_params *fp;
  /* shader globals, fragment:83-89 and the MRT outputs fragment:74-79 */
  float firstRayLength, glassFilter, originalRMEx, originalTPOx;
  v3 originalColor;
  v4 renderId, renderOriginalId;
  v3 ndc;                       /* clipSpace.xy / clipSpace.z of this fragment (z unused) */
  v3 relativePosition;          /* object-space position of the primary hit (vertex:53-63 interpolated) */
  flx_counters cnt;
} Frag;

static inline m3 rotation_at(const flx_scene_view *sc, int i) {          /* std140 mat3: 3 x vec4 columns */
  const float *r = sc->rotation + (size_t)i * 12;
  m3 m = { V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), V3(r[8], r[9], r[10]) };
  return m;
}
static inline v3 shift_at(const flx_scene_view *sc, int i) {
  const float *s = sc->shift + (size_t)i * 4;
  return V3(s[0], s[1], s[2]);
}

/* fragment:91-95 */
static float to4BitRepresentation(float a, float b) {
  uint32_t aui = flx_f2uint(a * 255.0f) & 240u;
  uint32_t bui = (flx_f2uint(b * 255.0f) & 240u) >> 4;
  return (float)(aui | bui) * INV_255;
}
/* fragment:97-101 */
static float normalToSphearical4BitRepresentation(v3 n) {
  float phi = (flx_atan2(n.z, n.x) * INV_PI) * 0.5f + 0.5f;
  float theta = (flx_atan2(n.x, n.y) * INV_PI) * 0.5f + 0.5f;
  return to4BitRepresentation(phi, theta);
}
/* fragment:103-105 */
static v3 combineNormalRME(v3 n, v3 rme) {
  return V3(normalToSphearical4BitRepresentation(n), rme.x, to4BitRepresentation(rme.y, rme.z));
}

/* fragment:108-117; texture() = NEAREST + REPEAT on an RGBA8 atlas */
static v3 fetchTexVal(Frag *f, int which, float u, float v, float texNum, v3 defaultVal) {
  if (texNum == -1.0f) return defaultVal;
  const flx_scene_view *sc = f->sc;
  uint32_t W = sc->atlas[which] ? sc->atlas_w[which] : 1u, H = sc->atlas[which] ? sc->atlas_h[which] : 1u;
  float tw = (float)f->fp->texture_width;
  float atlasHeightFactor = (float)W / (float)H;
  float cx = (u + flx_mod(texNum, tw)) / tw;
  float cy = ((v + flx_floor(texNum / tw)) * atlasHeightFactor) / tw;
  f->cnt.atlas_texels++;
  if (!sc->atlas[which]) return V3(0.0f, 0.0f, 0.0f);
  float fx = flx_fract(cx) * (float)W, fy = flx_fract(cy) * (float)H;
  uint32_t ix = flx_f2uint(fx), iy = flx_f2uint(fy);
  if (ix >= W) ix = W - 1u;
  if (iy >= H) iy = H - 1u;
  const uint8_t *t = sc->atlas[which] + ((size_t)iy * W + ix) * 4;
  return V3((float)t[0] / 255.0f, (float)t[1] / 255.0f, (float)t[2] / 255.0f);
}

/* fragment:119-121 */
static v4 noise(float random_seed, float nx, float ny, float seed) {
  float d = nx * 12.9898f + ny * 78.233f;
  float k = seed + random_seed * PHI;
  v4 r;
  r.x = flx_fract(flx_sin(d + 53.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  r.y = flx_fract(flx_sin(d + 59.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  r.z = flx_fract(flx_sin(d + 61.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  r.w = flx_fract(flx_sin(d + 67.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  return r;
}

/* fragment:123-140 */
static v3 moellerTrumbore(v3 a, v3 b, v3 c, Ray ray, float l) {
  const v3 zero = { 0.0f, 0.0f, 0.0f };
  v3 edge1 = sub3(b, a);
  v3 edge2 = sub3(c, a);
  v3 pvec = cross3(ray.unitDirection, edge2);
  float det = dot3(edge1, pvec);
  if (flx_abs(det) < BIAS) return zero;
  float inv_det = 1.0f / det;
  v3 tvec = sub3(ray.origin, a);
  float u = dot3(tvec, pvec) * inv_det;
  if (u < BIAS || u > 1.0f) return zero;
  v3 qvec = cross3(tvec, edge1);
  float v = dot3(ray.unitDirection, qvec) * inv_det;
  float uvSum = u + v;
  if (v < BIAS || uvSum > 1.0f) return zero;
  float s = dot3(edge2, qvec) * inv_det;
  if (s > l || s <= BIAS) return zero;
  return V3(s, u, v);
}

/* fragment:143-158 */
static int moellerTrumboreCull(v3 a, v3 b, v3 c, Ray ray, float l) {
  v3 edge1 = sub3(b, a);
  v3 edge2 = sub3(c, a);
  v3 pvec = cross3(ray.unitDirection, edge2);
  float det = dot3(edge1, pvec);
  float invDet = 1.0f / det;
  if (det < BIAS) return 0;
  v3 tvec = sub3(ray.origin, a);
  float u = dot3(tvec, pvec) * invDet;
  if (u < BIAS || u > 1.0f) return 0;
  v3 qvec = cross3(tvec, edge1);
  float v = dot3(ray.unitDirection, qvec) * invDet;
  if (v < BIAS || u + v > 1.0f) return 0;
  float s = dot3(edge2, qvec) * invDet;
  return (s <= l && s > BIAS);
}

/* Primary-visibility triangle test: what the rasteriser does for the pixel's ray —
 * front faces only (counter-clockwise on screen <=> det < 0 here), inclusive edges (no BIAS:
 * rasterised triangles have no cracks), in front of the near plane. Returns 0-vector on miss. */
static v3 moellerTrumborePrimary(v3 a, v3 b, v3 c, Ray ray, float l, float viewDepthPerS) {
  const v3 zero = { 0.0f, 0.0f, 0.0f };
  v3 edge1 = sub3(b, a);
  v3 edge2 = sub3(c, a);
  v3 pvec = cross3(ray.unitDirection, edge2);
  float det = dot3(edge1, pvec);
  if (!(det < 0.0f)) return zero;
  float inv_det = 1.0f / det;
  v3 tvec = sub3(ray.origin, a);
  float u = dot3(tvec, pvec) * inv_det;
  if (!(u >= 0.0f && u <= 1.0f)) return zero;
  v3 qvec = cross3(tvec, edge1);
  float v = dot3(ray.unitDirection, qvec) * inv_det;
  if (!(v >= 0.0f && u + v <= 1.0f)) return zero;
  float s = dot3(edge2, qvec) * inv_det;
  if (!(s < l) || !(s * viewDepthPerS >= NEAR_VIEW_DEPTH)) return zero;
  return V3(s, u, v);
}

/* fragment:161-167 */
static int rayCuboid(float l, Ray ray, v3 minCorner, v3 maxCorner) {
  v3 v0 = div3(sub3(minCorner, ray.origin), ray.unitDirection);
  v3 v1 = div3(sub3(maxCorner, ray.origin), ray.unitDirection);
  float tmin = flx_max(flx_max(flx_min(v0.x, v1.x), flx_min(v0.y, v1.y)), flx_min(v0.z, v1.z));
  float tmax = flx_min(flx_min(flx_max(v0.x, v1.x), flx_max(v0.y, v1.y)), flx_max(v0.z, v1.z));
  return tmax >= flx_max(tmin, BIAS) && tmin < l;
}

static const Hit NO_HIT = { { 0.0f, 0.0f, 0.0f }, 0, -1 };     /* fragment:81 */

/* analysis hook (tests/analysis/visit_histogram.py): when set, every entry fetch is tallied per entry index */
static uint64_t *g_visit_hist = NULL;
/* analysis hook: g_walk_hist[k] counts walks with 2^k <= visits < 2^(k+1) (k < 31), [31] = longest walk */
static uint64_t *g_walk_hist = NULL;
/* analysis hook (tests/analysis/walk_sim.py, single-threaded runs only): byte trace of every bounce walk — an 8-byte header
 * {0xF0 | kind (0 shadow, 1 closest), bounce, sample, 0, px lo, px hi, py lo, py hi}, one byte per entry visited
 * (its type: 0 terminator, 1 box, 2 triangle; | 0x10 when the entry's transform differs from the cached one), then 0xFF */
static uint8_t *g_trace = NULL;
static size_t g_trace_cap = 0, g_trace_len = 0;
static uint32_t g_trace_px = 0, g_trace_py = 0, g_trace_sample = 0, g_trace_bounce = 0;
static void trace_byte(uint8_t b) { if (g_trace && g_trace_len < g_trace_cap) g_trace[g_trace_len++] = b; }
/* second analysis hook: the entry INDEX of every bounce-walk visit as uint32, 0xFFFFFFFF after each walk */
static uint32_t *g_itrace = NULL;
static size_t g_itrace_cap = 0, g_itrace_len = 0;
static void itrace(uint32_t v) { if (g_itrace && g_itrace_len < g_itrace_cap) g_itrace[g_itrace_len++] = v; }
static void trace_begin(int kind) {
  if (!g_trace) return;
  trace_byte((uint8_t)(0xF0 | kind)); trace_byte((uint8_t)g_trace_bounce); trace_byte((uint8_t)g_trace_sample); trace_byte(0);
  trace_byte((uint8_t)(g_trace_px & 255u)); trace_byte((uint8_t)(g_trace_px >> 8)); trace_byte((uint8_t)(g_trace_py & 255u)); trace_byte((uint8_t)(g_trace_py >> 8));
}
static void tally_walk(uint64_t v) {
  if (!g_walk_hist) return;
  int k = 0; while ((v >> (k + 1)) && k < 30) k++;
  _Pragma("omp atomic") g_walk_hist[k]++;
  _Pragma("omp critical") { if (v > g_walk_hist[31]) g_walk_hist[31] = v; }
}

/* 1: walk the ray of fragment:591 in every bounce iteration, as the shader is written (see lightTrace below); 0 (default):
 * skip the walks whose result the loop guard discards.  Same frames, different closest_* counters. */
static int g_as_written = 0;
void flx_oracle_set_as_written(int on) { g_as_written = on != 0; }

/* fragment:172-227.  mode 0 = rayTracer as written; mode 1 = primary visibility (same walk, the
 * primary triangle rule, strict "<" so the first of equal-depth triangles is kept). */
static Hit rayTracerImpl(const flx_scene_view *sc, Ray ray, int mode, float viewDepthPerS, uint64_t *visits) {
  const uint64_t visits0 = *visits;
  Ray tR = ray;
  int cachedTI = 0;
  Hit hit = NO_HIT;
  float minLen = POW32;
  int size = (int)sc->n_entries_padded;
  if (!mode) trace_begin(1);
  for (int i = 0; i < size; i++) {
    const float *e = sc->geometry + (size_t)i * 12;
    (*visits)++;
    if (g_visit_hist) { _Pragma("omp atomic") g_visit_hist[i]++; }
    int tI = (int)e[9] << 1;
    if (!mode) { trace_byte((uint8_t)((int)e[10] | (tI != cachedTI ? 0x10 : 0))); itrace((uint32_t)i); }
    if (tI != cachedTI) {
      int iI = tI + 1;
      m3 rotationII = rotation_at(sc, iI);
      cachedTI = tI;
      tR.origin = m3mul(rotationII, add3(ray.origin, shift_at(sc, iI)));
      tR.unitDirection = m3mul(rotationII, ray.unitDirection);
    }
    if (e[10] == 0.0f) { if (!mode) { tally_walk(*visits - visits0); trace_byte(0xFF); itrace(0xFFFFFFFFu); } return hit; }
    if (e[10] == 1.0f) {
      if (!rayCuboid(minLen, tR, V3(e[0], e[1], e[2]), V3(e[3], e[4], e[5]))) i += (int)e[6];
    } else {
      v3 a = V3(e[0], e[1], e[2]), b = V3(e[3], e[4], e[5]), c = V3(e[6], e[7], e[8]);
      v3 intersection = mode ? moellerTrumborePrimary(a, b, c, tR, minLen, viewDepthPerS)
                             : moellerTrumbore(a, b, c, tR, minLen);
      if (intersection.x != 0.0f) {
        hit.suv = intersection; hit.transformId = tI; hit.triangleId = i;
        minLen = intersection.x;
      }
    }
  }
  if (!mode) { tally_walk(*visits - visits0); trace_byte(0xFF); itrace(0xFFFFFFFFu); }
  return hit;
}

/* fragment:231-280 */
static int shadowTestImpl(const flx_scene_view *sc, Ray ray, float l, uint64_t *visits) {
  Ray tR = ray;
  int cachedTI = 0;
  float minLen = l;
  int size = (int)sc->n_entries_padded;
  trace_begin(0);
  for (int i = 0; i < size; i++) {
    const float *e = sc->geometry + (size_t)i * 12;
    (*visits)++;
    if (g_visit_hist) { _Pragma("omp atomic") g_visit_hist[i]++; }
    int tI = (int)e[9] << 1;
    trace_byte((uint8_t)((int)e[10] | (tI != cachedTI ? 0x10 : 0))); itrace((uint32_t)i);
    if (tI != cachedTI) {
      int iI = tI + 1;
      m3 rotationII = rotation_at(sc, iI);
      cachedTI = tI;
      tR.origin = m3mul(rotationII, add3(ray.origin, shift_at(sc, iI)));
      tR.unitDirection = normalize3(m3mul(rotationII, ray.unitDirection));
    }
    if (e[10] == 0.0f) { trace_byte(0xFF); itrace(0xFFFFFFFFu); return 0; }
    if (e[10] == 1.0f) {
      if (!rayCuboid(minLen, tR, V3(e[0], e[1], e[2]), V3(e[3], e[4], e[5]))) i += (int)e[6];
    } else {
      if (moellerTrumboreCull(V3(e[0], e[1], e[2]), V3(e[3], e[4], e[5]), V3(e[6], e[7], e[8]), tR, minLen)) { trace_byte(0xFF); itrace(0xFFFFFFFFu); return 1; }
    }
  }
  trace_byte(0xFF); itrace(0xFFFFFFFFu);
  return 0;
}

/* fragment:282-302 */
static float trowbridgeReitz(float alpha, float NdotH) {
  float numerator = alpha * alpha;
  float denom = NdotH * NdotH * (numerator - 1.0f) + 1.0f;
  return numerator / flx_max(PI_F * denom * denom, BIAS);
}
static float schlickBeckmann(float alpha, float NdotX) {
  float k = alpha * 0.5f;
  float denominator = NdotX * (1.0f - k) + k;
  denominator = flx_max(denominator, BIAS);
  return NdotX / denominator;
}
static float smith(float alpha, float NdotV, float NdotL) {
  return schlickBeckmann(alpha, NdotV) * schlickBeckmann(alpha, NdotL);
}
static v3 fresnel(v3 F0, float theta) {
  float p = flx_pow5(1.0f - theta);
  return V3(F0.x + (1.0f - F0.x) * p, F0.y + (1.0f - F0.y) * p, F0.z + (1.0f - F0.z) * p);
}

/* fragment:304-334 */
static v3 forwardTrace(Material material, v3 lightDir, float strength, v3 N, v3 V) {
  float lenP1 = 1.0f + length3(lightDir);
  float brightness = strength / (lenP1 * lenP1);
  v3 L = normalize3(lightDir);
  v3 H = normalize3(add3(V, L));
  float VdotH = flx_max(dot3(V, H), 0.0f);
  float NdotL = flx_max(dot3(N, L), 0.0f);
  float NdotH = flx_max(dot3(N, H), 0.0f);
  float NdotV = flx_max(dot3(N, V), 0.0f);
  float alpha = material.rme.x * material.rme.x;
  float BRDF = flx_mix(1.0f, NdotV, material.rme.y);
  v3 F0 = scale3(material.albedo, BRDF);
  v3 Ks = fresnel(F0, VdotH);
  float oneMinusMetal = 1.0f - material.rme.y;
  v3 Kd = V3((1.0f - Ks.x) * oneMinusMetal, (1.0f - Ks.y) * oneMinusMetal, (1.0f - Ks.z) * oneMinusMetal);
  v3 lambert = scale3(material.albedo, INV_PI);
  float tr = trowbridgeReitz(alpha, NdotH);
  float sm = smith(alpha, NdotV, NdotL);
  v3 cookTorranceNumerator = scale3(scale3(Ks, tr), sm);
  float cookTorranceDenominator = 4.0f * NdotV * NdotL;
  cookTorranceDenominator = flx_max(cookTorranceDenominator, BIAS);
  v3 cookTorrance = divs3(cookTorranceNumerator, cookTorranceDenominator);
  v3 radiance = add3(mul3(Kd, lambert), cookTorrance);
  return scale3(scale3(radiance, NdotL), brightness);
}

/* fragment:400-461 */
static v3 reservoirSample(Frag *f, Material material, Ray ray, v4 randomVec, v3 N, v3 smoothNormal,
                          float geometryOffset, int dontFilter, int i) {
  const flx_scene_view *sc = f->sc;
  float rs = f->fp->random_seed;
  v3 localColor = V3(0.0f, 0.0f, 0.0f);
  float reservoirLength = 0.0f;
  float totalWeight = 0.0f;
  int reservoirNum = 0;
  float reservoirWeight = 0.0f;
  v3 reservoirLightDir = V3(0.0f, 0.0f, 0.0f);
  v4 n0 = noise(rs, randomVec.z, randomVec.w, BIAS);
  float lastRandomX = n0.x, lastRandomY = n0.y;
  int size = (int)sc->n_lights;                 /* L = 0: the 1x1 zero texture's strength is 0 -> skipped */
  for (int j = 0; j < size; j++) {
    const float *lt = sc->lights + (size_t)j * 6;
    float strength = lt[3], variation = lt[4];
    if (strength <= 0.0f) continue;
    reservoirLength += 1.0f;
    v3 light = add3(V3(lt[0], lt[1], lt[2]), scale3(V3(randomVec.x, randomVec.y, randomVec.z), variation));
    v3 dir = sub3(light, ray.origin);
    v3 colorForLight = forwardTrace(material, dir, strength, N, neg3(ray.unitDirection));
    localColor = add3(localColor, colorForLight);
    float weight = length3(colorForLight);
    totalWeight += weight;
    if (flx_abs(lastRandomY) * totalWeight <= weight) {
      reservoirNum = j;
      reservoirWeight = weight;
      reservoirLightDir = dir;
    }
    v4 n1 = noise(rs, lastRandomX, lastRandomY, BIAS);
    lastRandomX = n1.z; lastRandomY = n1.w;
  }
  v3 unitLightDir = normalize3(reservoirLightDir);
  int showColor = reservoirLength == 0.0f || reservoirWeight == 0.0f;
  int showShadow = dot3(smoothNormal, unitLightDir) <= BIAS;
  v3 baseLuminance = V3(material.rme.z, material.rme.z, material.rme.z);
  if (dontFilter || i == 0) f->renderId.w = (float)((reservoirNum % 128) << 1) * INV_255;
  if (showColor) return add3(localColor, baseLuminance);
  if (showShadow) {
    if (dontFilter || i == 0) f->renderId.w += INV_255;
    return baseLuminance;
  }
  v3 offsetTarget = add3(ray.origin, scale3(smoothNormal, geometryOffset));
  Ray lightRay = { offsetTarget, unitLightDir };
  f->cnt.shadow_walks++;
  if (shadowTestImpl(sc, lightRay, length3(reservoirLightDir), &f->cnt.shadow_visits)) {
    if (dontFilter || i == 0) f->renderId.w += INV_255;
    return baseLuminance;
  }
  return add3(localColor, baseLuminance);
}

/* fragment:464-599.  `dir0` is the primary ray's unit direction (normalize(target - camera), :472). */
static v3 lightTrace(Frag *f, Hit hit, v3 dir0, v3 camera, float cosSampleN, int bounces) {
  const flx_scene_view *sc = f->sc;
  const flx_frame_params *fp = f->fp;
  v3 ambient = V3(fp->ambient[0], fp->ambient[1], fp->ambient[2]);
  int dontFilter = 1;
  v3 finalColor = V3(0.0f, 0.0f, 0.0f);
  v3 importancyFactor = V3(1.0f, 1.0f, 1.0f);
  f->originalColor = V3(1.0f, 1.0f, 1.0f);
  Ray ray = { camera, dir0 };
  v3 lastHitPoint = camera;
  for (int i = 0; i < bounces && length3(mul3(importancyFactor, f->originalColor)) >= fp->min_importancy * SQRT3; i++) {
    float fi = (float)i;
    if (g_trace) g_trace_bounce = (uint32_t)i;          /* analysis hook only: never written by the threads of a normal render */
    f->cnt.shades++;
    m3 rTI = rotation_at(sc, hit.transformId);
    v3 sTI = shift_at(sc, hit.transformId);
    ray.origin = add3(scale3(ray.unitDirection, hit.suv.x), ray.origin);
    v3 uvw = V3(1.0f - hit.suv.y - hit.suv.z, hit.suv.y, hit.suv.z);
    const float *g = sc->geometry + (size_t)hit.triangleId * 12;
    v3 t0v = m3mul(rTI, V3(g[0], g[1], g[2]));
    v3 t1v = m3mul(rTI, V3(g[3], g[4], g[5]));
    v3 t2v = m3mul(rTI, V3(g[6], g[7], g[8]));
    v3 offsetRayTarget = sub3(ray.origin, sTI);
    v3 geometryNormal = normalize3(cross3(sub3(t0v, t1v), sub3(t0v, t2v)));
    v3 diffs = V3(distance3(offsetRayTarget, t0v), distance3(offsetRayTarget, t1v), distance3(offsetRayTarget, t2v));
    const float *t = sc->attributes + (size_t)hit.triangleId * 28;
    v3 n0 = m3mul(rTI, V3(t[0], t[1], t[2]));
    v3 n1 = m3mul(rTI, V3(t[3], t[4], t[5]));
    v3 n2 = m3mul(rTI, V3(t[6], t[7], t[8]));
    /* normals * uvw */
    v3 smoothNormal = normalize3(V3((n0.x * uvw.x + n1.x * uvw.y) + n2.x * uvw.z,
                                    (n0.y * uvw.x + n1.y * uvw.y) + n2.y * uvw.z,
                                    (n0.z * uvw.x + n1.z * uvw.y) + n2.z * uvw.z));
    /* geometryNormal * normals = (dot(gN, n0), dot(gN, n1), dot(gN, n2)) */
    v3 angles = V3(flx_acos(flx_abs(dot3(geometryNormal, n0))), flx_acos(flx_abs(dot3(geometryNormal, n1))),
                   flx_acos(flx_abs(dot3(geometryNormal, n2))));
    v3 angleTan = V3(flx_clamp(flx_tan(angles.x), 0.0f, 1.0f), flx_clamp(flx_tan(angles.y), 0.0f, 1.0f),
                     flx_clamp(flx_tan(angles.z), 0.0f, 1.0f));
    float geometryOffset = dot3(mul3(diffs, angleTan), uvw);
    /* mat3x2(t2.yzw, t3.xyz) * uvw */
    float bu = (t[9] * uvw.x + t[11] * uvw.y) + t[13] * uvw.z;
    float bv = (t[10] * uvw.x + t[12] * uvw.y) + t[14] * uvw.z;
    Material material;
    material.albedo = fetchTexVal(f, 0, bu, bv, t[15], V3(t[18], t[19], t[20]));
    material.rme = fetchTexVal(f, 1, bu, bv, t[16], V3(t[21], t[22], t[23]));
    material.tpo = fetchTexVal(f, 2, bu, bv, t[17], V3(t[24], t[25], t[26]));

    ray.unitDirection = normalize3(sub3(ray.origin, lastHitPoint));
    float signDir = flx_sign(dot3(ray.unitDirection, smoothNormal));
    smoothNormal = scale3(smoothNormal, -signDir);

    v4 randomVec = noise(fp->random_seed, f->ndc.x, f->ndc.y, fi + cosSampleN);
    v3 randomSpheareVec = normalize3(add3(smoothNormal, normalize3(V3(randomVec.x, randomVec.y, randomVec.z))));
    float BRDF = flx_mix(1.0f, flx_abs(dot3(smoothNormal, ray.unitDirection)), material.rme.y);
    float roughnessBRDF = material.rme.x * BRDF;
    v3 roughNormal = normalize3(mix3(smoothNormal, randomSpheareVec, roughnessBRDF));
    v3 H = normalize3(sub3(roughNormal, ray.unitDirection));
    float VdotH = flx_max(dot3(neg3(ray.unitDirection), H), 0.0f);
    v3 F0 = scale3(material.albedo, BRDF);
    v3 fr = fresnel(F0, VdotH);
    float fresnelReflect = flx_max(fr.x, flx_max(fr.y, fr.z));
    int isSolid = material.tpo.x * fresnelReflect <= flx_abs(randomVec.w);

    if (dontFilter) {
      f->originalTPOx = material.tpo.x;
      f->originalColor = mul3(f->originalColor, material.albedo);
      f->originalRMEx += material.rme.x;
      float scale = flx_exp2_neg_int(i);                     /* pow(2.0, -fi) */
      v3 cn = combineNormalRME(smoothNormal, material.rme);
      v4 renderIdUpdate = { scale * cn.x, scale * cn.y, scale * cn.z, scale * 0.0f };
      f->renderId.x += renderIdUpdate.x; f->renderId.y += renderIdUpdate.y;
      f->renderId.z += renderIdUpdate.z; f->renderId.w += renderIdUpdate.w;
      if (i == 0) {
        f->renderOriginalId.x += renderIdUpdate.x; f->renderOriginalId.y += renderIdUpdate.y;
        f->renderOriginalId.z += renderIdUpdate.z; f->renderOriginalId.w += renderIdUpdate.w;
      }
      dontFilter = (material.rme.x < 0.01f && isSolid) || !isSolid;
      if (isSolid && material.tpo.x > 0.01f) {
        f->glassFilter += 1.0f;
        dontFilter = 0;
      }
    } else {
      importancyFactor = mul3(importancyFactor, material.albedo);
    }

    if (i == 1) f->firstRayLength = flx_min(length3(sub3(ray.origin, lastHitPoint)) / length3(sub3(lastHitPoint, camera)), f->firstRayLength);
    v3 localColor = reservoirSample(f, material, ray, randomVec, scale3(roughNormal, -signDir), scale3(smoothNormal, -signDir),
                                    geometryOffset, dontFilter, i);
    finalColor = add3(finalColor, mul3(localColor, importancyFactor));
    if (isSolid) {
      ray.unitDirection = normalize3(mix3(reflect3(ray.unitDirection, smoothNormal), randomSpheareVec, roughnessBRDF));
    } else {
      float eta = flx_mix(1.0f / material.tpo.z, material.tpo.z, flx_max(signDir, 0.0f));
      ray.unitDirection = normalize3(mix3(refract3(ray.unitDirection, smoothNormal, eta), randomSpheareVec, roughnessBRDF));
    }
    /* fragment:591 traces the next ray even when the loop guard (:475) is about to end the loop — in the last iteration,
     * or once the path's importancy has dropped below the threshold: a hit nobody shades ("incl. the useless
     * last one").  Its result reaches no output, so it is not walked, here and in the kernels alike — unless
     * flx_oracle_set_as_written(1) asks for the shader's statement order as it stands: then every iteration ends with that
     * walk (the counters are the shader's own work) and the loop guard at the top ends the loop; tests/test_oracle_kat.py
     * holds the two modes' frames against each other bit for bit. */
    if (!g_as_written && !(i + 1 < bounces && length3(mul3(importancyFactor, f->originalColor)) >= fp->min_importancy * SQRT3)) break;
    f->cnt.closest_walks++;
    hit = rayTracerImpl(sc, ray, 0, 0.0f, &f->cnt.closest_visits);
    if (hit.triangleId == -1) break;
    lastHitPoint = ray.origin;
  }
  return add3(finalColor, mul3(importancyFactor, ambient));
}

/* Primary visibility: replaces vertex shader + rasteriser (vertex:40-72). */
typedef struct { float inv_view[9]; v3 view_row2; } PrimarySetup;

static void primary_setup(const flx_frame_params *fp, PrimarySetup *ps) {
  flx_invert3x3(fp->view_matrix, ps->inv_view);
  ps->view_row2 = V3(fp->view_matrix[6], fp->view_matrix[7], fp->view_matrix[8]);
}
static Hit primary_hit(Frag *f, const PrimarySetup *ps, uint32_t px, uint32_t py_gl, v3 *dirOut) {
  const flx_frame_params *fp = f->fp;
  float nx = ((float)px + 0.5f) / (float)fp->width * 2.0f - 1.0f;
  float ny = ((float)py_gl + 0.5f) / (float)fp->height * 2.0f - 1.0f;
  f->ndc = V3(nx, ny, 1.0f);
  const float *iv = ps->inv_view;
  /* V^-1 * (nx, ny, 1): columns of the row-major inverse */
  v3 d = V3((iv[0] * nx + iv[1] * ny) + iv[2], (iv[3] * nx + iv[4] * ny) + iv[5], (iv[6] * nx + iv[7] * ny) + iv[8]);
  d = normalize3(d);
  *dirOut = d;
  Ray ray = { V3(fp->camera[0], fp->camera[1], fp->camera[2]), d };
  float viewDepthPerS = dot3(ps->view_row2, d);
  return rayTracerImpl(f->sc, ray, 1, viewDepthPerS, &f->cnt.primary_visits);
}

/* fragment:601-646, one pixel.  out* are 4-float slots (any may be NULL). */
static void fragment_main(Frag *f, const PrimarySetup *ps, uint32_t px, uint32_t py_gl, float *outRgba,
                          float *gColor, float *gColorIp, float *gOrigColor, float *gId, float *gOrigId, float *gLoc) {
  const flx_frame_params *fp = f->fp;
  f->firstRayLength = 1.0f; f->glassFilter = 0.0f; f->originalRMEx = 0.0f; f->originalTPOx = 0.0f;
  f->originalColor = V3(0.0f, 0.0f, 0.0f);
  f->renderId.x = f->renderId.y = f->renderId.z = f->renderId.w = 0.0f;
  f->renderOriginalId = f->renderId;
  v3 dir0;
  Hit hit = primary_hit(f, ps, px, py_gl, &dir0);
  float zero4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  if (hit.triangleId == -1) {            /* not covered: stays at the clear colour (pathtracerWGL2.js:416,718) */
    if (outRgba) memcpy(outRgba, zero4, sizeof zero4);
    if (gColor) memcpy(gColor, zero4, sizeof zero4);
    if (gColorIp) memcpy(gColorIp, zero4, sizeof zero4);
    if (gOrigColor) memcpy(gOrigColor, zero4, sizeof zero4);
    if (gId) memcpy(gId, zero4, sizeof zero4);
    if (gOrigId) memcpy(gOrigId, zero4, sizeof zero4);
    if (gLoc) memcpy(gLoc, zero4, sizeof zero4);
    return;
  }
  f->cnt.primary_hits++;
  v3 camera = V3(fp->camera[0], fp->camera[1], fp->camera[2]);
  v3 finalColor = V3(0.0f, 0.0f, 0.0f);
  if (g_trace) { g_trace_px = px; g_trace_py = py_gl; }
  for (int i = 0; i < fp->samples; i++) {
    if (g_trace) g_trace_sample = (uint32_t)i;
    float cosSampleN = flx_cos((float)i);
    finalColor = add3(finalColor, lightTrace(f, hit, dir0, camera, cosSampleN, fp->max_reflections));
  }
  float invSamples = 1.0f / (float)fp->samples;
  finalColor = scale3(finalColor, invSamples);
  float color[4] = { 0, 0, 0, 0 }, colorIp[4] = { 0, 0, 0, 0 };
  if (fp->use_filter == 1) {
    color[0] = flx_fract(finalColor.x); color[1] = flx_fract(finalColor.y); color[2] = flx_fract(finalColor.z); color[3] = 1.0f;
    colorIp[0] = flx_floor(finalColor.x) * INV_256; colorIp[1] = flx_floor(finalColor.y) * INV_256;
    colorIp[2] = flx_floor(finalColor.z) * INV_256; colorIp[3] = f->glassFilter;
  } else {
    finalColor = mul3(finalColor, f->originalColor);
    if (fp->is_temporal == 1) {
      color[0] = flx_fract(finalColor.x); color[1] = flx_fract(finalColor.y); color[2] = flx_fract(finalColor.z); color[3] = 1.0f;
      colorIp[0] = flx_floor(finalColor.x) * INV_256; colorIp[1] = flx_floor(finalColor.y) * INV_256;
      colorIp[2] = flx_floor(finalColor.z) * INV_256; colorIp[3] = 1.0f;
    } else {
      color[0] = finalColor.x; color[1] = finalColor.y; color[2] = finalColor.z; color[3] = 1.0f;
    }
  }
  if (outRgba) memcpy(outRgba, color, sizeof color);
  if (gColor) memcpy(gColor, color, sizeof color);
  if (gColorIp) memcpy(gColorIp, colorIp, sizeof colorIp);
  if (gOrigColor) {
    gOrigColor[0] = f->originalColor.x; gOrigColor[1] = f->originalColor.y; gOrigColor[2] = f->originalColor.z;
    gOrigColor[3] = flx_min(f->originalRMEx, f->firstRayLength) + INV_255;
  }
  if (gId) { gId[0] = f->renderId.x; gId[1] = f->renderId.y; gId[2] = f->renderId.z; gId[3] = f->renderId.w + INV_255; }
  if (gOrigId) { gOrigId[0] = 0.0f; gOrigId[1] = 0.0f; gOrigId[2] = 0.0f; gOrigId[3] = f->originalTPOx + INV_255; }
  if (gLoc) {                                /* fragment:640-642; relativePosition is in object space, camera in world space, as in the shader */
    const float *g = f->sc->geometry + (size_t)hit.triangleId * 12;
    float w0 = 1.0f - hit.suv.y - hit.suv.z;
    v3 rel = add3(add3(scale3(V3(g[0], g[1], g[2]), w0), scale3(V3(g[3], g[4], g[5]), hit.suv.y)), scale3(V3(g[6], g[7], g[8]), hit.suv.z));
    float div = 2.0f * distance3(rel, camera);
    gLoc[0] = flx_mod(rel.x, div) / div; gLoc[1] = flx_mod(rel.y, div) / div; gLoc[2] = flx_mod(rel.z, div) / div; gLoc[3] = INV_255;
  }
}

static int check_inputs(const flx_scene_view *sc, const flx_frame_params *fp) {
  if (!sc || !fp || !sc->geometry || !sc->attributes || !sc->rotation || !sc->shift) return 0;
  if (sc->n_entries_padded == 0 || sc->n_transforms == 0) return 0;
  if (fp->width == 0 || fp->height == 0 || fp->samples < 1 || fp->max_reflections < 0 || fp->texture_width < 1) return 0;
  if (sc->n_lights && !sc->lights) return 0;
  return 1;
}

static uint32_t tile_rows_total(const flx_frame_params *fp, uint32_t *rows, uint32_t cap) {
  uint32_t tr = fp->tile_rows, tc = fp->tile_count, ti = fp->tile_index, n = 0;
  if (tr == 0 || tc <= 1) { tr = fp->height ? fp->height : 1; tc = 1; ti = 0; }
  for (uint32_t y = 0; y < fp->height; y++) {
    if ((y / tr) % tc == ti) { if (rows && n < cap) rows[n] = y; n++; }
  }
  return n;
}

/* The path-trace pass alone: what the fragment program writes for every pixel (no temporal pass, no filter). */
int flx_oracle_trace(const flx_scene_view *scene, const flx_frame_params *params, float *out_rgba,
                     const flx_gbuffers *gb, flx_counters *counters, int threads) {
  if (!check_inputs(scene, params)) return FLX_ERR_INVALID;
  PrimarySetup ps;
  primary_setup(params, &ps);
  uint32_t W = params->width, H = params->height;
  uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * H);
  if (!rows) return FLX_ERR_INVALID;
  uint32_t nrows = tile_rows_total(params, rows, H);
  flx_counters total;
  memset(&total, 0, sizeof total);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel
  {
    Frag f;
    memset(&f, 0, sizeof f);
    f.sc = scene; f.fp = params;
#pragma omp for schedule(dynamic, 1)
    for (uint32_t k = 0; k < nrows; k++) {
      uint32_t row = rows[k];                   /* image row, 0 = top */
      uint32_t py_gl = H - 1u - row;
      for (uint32_t px = 0; px < W; px++) {
        size_t o = ((size_t)k * W + px) * 4;
        fragment_main(&f, &ps, px, py_gl, out_rgba ? out_rgba + o : NULL,
                      gb && gb->color ? gb->color + o : NULL, gb && gb->color_ip ? gb->color_ip + o : NULL,
                      gb && gb->original_color ? gb->original_color + o : NULL, gb && gb->id ? gb->id + o : NULL,
                      gb && gb->original_id ? gb->original_id + o : NULL, gb && gb->location_id ? gb->location_id + o : NULL);
      }
    }
#pragma omp critical
    {
      total.primary_visits += f.cnt.primary_visits; total.closest_visits += f.cnt.closest_visits;
      total.shadow_visits += f.cnt.shadow_visits; total.closest_walks += f.cnt.closest_walks;
      total.shadow_walks += f.cnt.shadow_walks; total.shades += f.cnt.shades;
      total.primary_hits += f.cnt.primary_hits; total.atlas_texels += f.cnt.atlas_texels;
    }
  }
  free(rows);
  if (counters) *counters = total;
  return FLX_OK;
}

/* One frame as renderFrame() produces it (pathtracerWGL2.js:375-554): trace, then the temporal pass over an
 * EMPTY history when is_temporal (= the first frame after a reset; use flx_oracle_render_sequence for more),
 * then the filter chain when use_filter. */
int flx_oracle_render(const flx_scene_view *scene, const flx_frame_params *params, float *out_rgba,
                      const flx_gbuffers *gb_in, flx_counters *counters, int threads) {
  if (!check_inputs(scene, params)) return FLX_ERR_INVALID;
  if (params->is_temporal != 1 && params->use_filter != 1) return flx_oracle_trace(scene, params, out_rgba, gb_in, counters, threads);
  if (flx_tile_rows_of(params) != params->height) return FLX_ERR_INVALID;   /* temporal / filter frames are whole frames */
  return flx_oracle_render_sequence_impl(scene, params, 1, out_rgba, gb_in, counters, threads);
}

uint32_t flx_tile_rows_of(const flx_frame_params *params) { return tile_rows_total(params, NULL, 0); }

/* ---- known-answer hooks ---------------------------------------------------------------------------- */
static Ray mkray(const float o[3], const float d[3]) { Ray r = { V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]) }; return r; }

void flx_oracle_moeller_trumbore(const float tri[9], const float origin[3], const float dir[3], float l, float suv[3]) {
  v3 r = moellerTrumbore(V3(tri[0], tri[1], tri[2]), V3(tri[3], tri[4], tri[5]), V3(tri[6], tri[7], tri[8]), mkray(origin, dir), l);
  suv[0] = r.x; suv[1] = r.y; suv[2] = r.z;
}
int flx_oracle_moeller_trumbore_cull(const float tri[9], const float origin[3], const float dir[3], float l) {
  return moellerTrumboreCull(V3(tri[0], tri[1], tri[2]), V3(tri[3], tri[4], tri[5]), V3(tri[6], tri[7], tri[8]), mkray(origin, dir), l);
}
int flx_oracle_ray_cuboid(float l, const float origin[3], const float dir[3], const float mn[3], const float mx[3]) {
  return rayCuboid(l, mkray(origin, dir), V3(mn[0], mn[1], mn[2]), V3(mx[0], mx[1], mx[2]));
}
void flx_oracle_noise(float nx, float ny, float seed, float random_seed, float out[4]) {
  v4 r = noise(random_seed, nx, ny, seed);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void flx_oracle_forward_trace(const float m[9], const float light_dir[3], float strength, const float n[3], const float v[3], float out[3]) {
  Material mat = { V3(m[0], m[1], m[2]), V3(m[3], m[4], m[5]), V3(m[6], m[7], m[8]) };
  v3 r = forwardTrace(mat, V3(light_dir[0], light_dir[1], light_dir[2]), strength, V3(n[0], n[1], n[2]), V3(v[0], v[1], v[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* reservoirSample (fragment:400-461) on its own, for the literal known-answer table (tests/golden/shading_kat.json): the lights of
 * `lights` (6 floats each), an empty scene behind them (the shadow ray finds nothing), out = the returned colour and renderId.w. */
void flx_oracle_reservoir_sample(const float *lights, uint32_t n_lights, float random_seed, const float m[9], const float origin[3], const float unit_dir[3],
                                 const float random_vec[4], const float n[3], const float smooth_normal[3], float geometry_offset, int dont_filter, int i,
                                 float out[4]) {
  static const float no_geometry[12 * 256] = { 0.0f };
  static const float identity_rot[24] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
  static const float zero_shift[8] = { 0.0f };
  flx_scene_view sc;
  memset(&sc, 0, sizeof sc);
  sc.geometry = no_geometry; sc.attributes = no_geometry; sc.n_entries_padded = 256;
  sc.rotation = identity_rot; sc.shift = zero_shift; sc.n_transforms = 1;
  sc.lights = lights; sc.n_lights = n_lights;
  flx_frame_params fp;
  memset(&fp, 0, sizeof fp);
  fp.random_seed = random_seed;
  Frag f;
  memset(&f, 0, sizeof f);
  f.sc = &sc; f.fp = &fp;
  Material mat = { V3(m[0], m[1], m[2]), V3(m[3], m[4], m[5]), V3(m[6], m[7], m[8]) };
  Ray ray = { V3(origin[0], origin[1], origin[2]), V3(unit_dir[0], unit_dir[1], unit_dir[2]) };
  v4 rv = { random_vec[0], random_vec[1], random_vec[2], random_vec[3] };
  v3 c = reservoirSample(&f, mat, ray, rv, V3(n[0], n[1], n[2]), V3(smooth_normal[0], smooth_normal[1], smooth_normal[2]), geometry_offset, dont_filter, i);
  out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = f.renderId.w;
}
/* One whole iteration of lightTrace's bounce loop (fragment:464-599 with bounces = 1) on a scene of ONE triangle under transform
 * `rotation / shift` (24 + 8 floats: forward then inverse, as the scene arrays hold them) and `n_lights` lights, for the literal
 * known-answer table (tests/golden/shading_kat.json "bounce"): the primary ray leaves `camera` in `dir0`, hits the triangle at `suv`,
 * the bounce is shaded, the shadow ray and the next ray meet the triangle or nothing.  out[0..2] the returned colour, [3..5]
 * originalColor, [6..9] renderId, [10..13] renderOriginalId, [14] originalRMEx, [15] originalTPOx, [16] glassFilter. */
void flx_oracle_light_trace_bounce(const float geometry[12], const float attributes[28], const float *rotation, const float *shift, int transform,
                                   const float *lights, uint32_t n_lights, const float ambient[3], float random_seed, float min_importancy,
                                   const float ndc[2], const float camera[3], const float dir0[3], const float suv[3], float cos_sample_n, float out[17]) {
  static float geo[12 * 256], att[28 * 256];
  memset(geo, 0, sizeof geo); memset(att, 0, sizeof att);
  memcpy(geo, geometry, 12 * sizeof(float)); memcpy(att, attributes, 28 * sizeof(float));
  flx_scene_view sc;
  memset(&sc, 0, sizeof sc);
  sc.geometry = geo; sc.attributes = att; sc.n_entries_padded = 256;
  sc.rotation = rotation; sc.shift = shift; sc.n_transforms = (uint32_t)transform + 1u;
  sc.lights = lights; sc.n_lights = n_lights;
  flx_frame_params fp;
  memset(&fp, 0, sizeof fp);
  fp.random_seed = random_seed; fp.min_importancy = min_importancy; fp.texture_width = 1;
  fp.ambient[0] = ambient[0]; fp.ambient[1] = ambient[1]; fp.ambient[2] = ambient[2];
  Frag f;
  memset(&f, 0, sizeof f);
  f.sc = &sc; f.fp = &fp;
  f.firstRayLength = 1.0f;
  f.ndc = V3(ndc[0], ndc[1], 1.0f);
  Hit hit = { V3(suv[0], suv[1], suv[2]), transform << 1, 0 };
  v3 c = lightTrace(&f, hit, V3(dir0[0], dir0[1], dir0[2]), V3(camera[0], camera[1], camera[2]), cos_sample_n, 1);
  out[0] = c.x; out[1] = c.y; out[2] = c.z;
  out[3] = f.originalColor.x; out[4] = f.originalColor.y; out[5] = f.originalColor.z;
  out[6] = f.renderId.x; out[7] = f.renderId.y; out[8] = f.renderId.z; out[9] = f.renderId.w;
  out[10] = f.renderOriginalId.x; out[11] = f.renderOriginalId.y; out[12] = f.renderOriginalId.z; out[13] = f.renderOriginalId.w;
  out[14] = f.originalRMEx; out[15] = f.originalTPOx; out[16] = f.glassFilter;
}
void flx_oracle_ray_tracer(const flx_scene_view *scene, const float origin[3], const float dir[3], float hit_suv[3],
                           int *transform_id, int *triangle_id, uint64_t *visits) {
  uint64_t v = 0;
  Hit h = rayTracerImpl(scene, mkray(origin, dir), 0, 0.0f, &v);
  hit_suv[0] = h.suv.x; hit_suv[1] = h.suv.y; hit_suv[2] = h.suv.z;
  *transform_id = h.transformId; *triangle_id = h.triangleId;
  if (visits) *visits = v;
}
int flx_oracle_shadow_test(const flx_scene_view *scene, const float origin[3], const float dir[3], float l, uint64_t *visits) {
  uint64_t v = 0;
  int r = shadowTestImpl(scene, mkray(origin, dir), l, &v);
  if (visits) *visits = v;
  return r;
}
void flx_oracle_primary(const flx_scene_view *scene, const flx_frame_params *params, uint32_t px, uint32_t py_gl,
                        float hit_suv[3], int *transform_id, int *triangle_id, float dir[3]) {
  Frag f;
  memset(&f, 0, sizeof f);
  f.sc = scene; f.fp = params;
  PrimarySetup ps;
  primary_setup(params, &ps);
  v3 d;
  Hit h = primary_hit(&f, &ps, px, py_gl, &d);
  hit_suv[0] = h.suv.x; hit_suv[1] = h.suv.y; hit_suv[2] = h.suv.z;
  *transform_id = h.transformId; *triangle_id = h.triangleId;
  dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}
void flx_oracle_set_visit_histogram(uint64_t *hist) { g_visit_hist = hist; }
void flx_oracle_set_walk_histogram(uint64_t *hist32) { g_walk_hist = hist32; }
void flx_oracle_set_trace(uint8_t *buf, size_t cap) { g_trace = buf; g_trace_cap = cap; g_trace_len = 0; }
size_t flx_oracle_trace_length(void) { return g_trace_len; }
void flx_oracle_set_index_trace(uint32_t *buf, size_t cap) { g_itrace = buf; g_itrace_cap = cap; g_itrace_len = 0; }
size_t flx_oracle_index_trace_length(void) { return g_itrace_len; }

void flx_oracle_math(int fn, const float *a, const float *b, float *out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    float x = a[i], y = b ? b[i] : 0.0f, r;
    switch (fn) {
      case 0: r = flx_sin(x); break;
      case 1: r = flx_cos(x); break;
      case 2: r = flx_tan(x); break;
      case 3: r = flx_acos(x); break;
      case 4: r = flx_atan2(x, y); break;
      case 5: r = flx_exp(x); break;
      case 6: r = flx_pow(x, y); break;
      case 7: r = flx_tanh(x); break;
      case 8: r = flx_floor(x); break;
      case 9: r = flx_sqrt(x); break;
      case 10: r = x / y; break;
      default: r = flx_nanf(); break;
    }
    out[i] = r;
  }
}
