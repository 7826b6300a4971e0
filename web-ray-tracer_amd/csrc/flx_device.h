/*
 * flx_device.h — device-side building blocks of the HIP path tracer (gfx950).
 *
 * The arithmetic of every routine is the reference shader's, operation for operation
 * (shaders/pathtracer_fragment.glsl; cited per function), with GLSL built-ins expanded as
 * include/flx_math.h pins them, so that results equal the CPU oracle's bit for bit when compiled
 * with -ffp-contract=off.  What is NOT the shader's: the memory layout (flat float4 arrays instead
 * of textures), the primary-ray kernel (the reference rasterises) and the way work
 * is mapped to wave64 lanes (flx_kernels.hip).
 */
#ifndef FLX_DEVICE_H
#define FLX_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flx_math.h"

#define FLX_DEV __device__ __forceinline__

namespace flx {

constexpr float PHI = 1.61803398874989484820459f;     /* fragment:5 */
constexpr float SQRT3 = 1.7320508075688772f;          /* fragment:6 */
constexpr float INV_PI = 0.3183098861837907f;         /* fragment:10 */
constexpr float INV_256 = 0.00390625f;                /* fragment:11 */
constexpr float INV_255 = 0.00392156862745098f;       /* fragment:12 */
constexpr float PI_F = 3.141592653589793f;            /* fragment:4 */
constexpr float BIAS = FLX_BIAS;
constexpr float POW32 = FLX_POW32;
constexpr float NEAR_VIEW_DEPTH = 0.5f;               /* SURVEY §8a P0 */

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct M3 { f3 c0, c1, c2; };                          /* columns, like GLSL mat3 */

FLX_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
FLX_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
FLX_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
FLX_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
FLX_DEV f3 operator/(f3 a, f3 b) { return F3(a.x / b.x, a.y / b.y, a.z / b.z); }
FLX_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
FLX_DEV f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
FLX_DEV f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
FLX_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
FLX_DEV f3 cross(f3 a, f3 b) { return F3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
FLX_DEV float length(f3 a) { return flx_sqrt(dot(a, a)); }
FLX_DEV f3 normalize(f3 a) { return a / length(a); }
FLX_DEV float distance(f3 a, f3 b) { return length(a - b); }
FLX_DEV f3 mix(f3 a, f3 b, float t) { return F3(flx_mix(a.x, b.x, t), flx_mix(a.y, b.y, t), flx_mix(a.z, b.z, t)); }
FLX_DEV f3 mul(const M3 &m, f3 v) {
  return F3((m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z,
            (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z,
            (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z);
}
FLX_DEV f3 reflect(f3 I, f3 N) { return I - N * (2.0f * dot(N, I)); }
FLX_DEV f3 refract(f3 I, f3 N, float eta) {
  float d = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if (k < 0.0f) return F3(0.0f, 0.0f, 0.0f);
  return I * eta - N * (eta * d + flx_sqrt(k));
}

struct Ray { f3 origin, dir; };                        /* fragment:19-22 */
struct Material { f3 albedo, rme, tpo; };              /* fragment:24-28 */
struct Hit { f3 suv; int transformId; int triangleId; };   /* fragment:30-34 */

/* Device-resident scene: the flat arrays of SURVEY §8a D1–D6 (no textures, no 256-entry rows). */
struct DeviceScene {
  const float4 *geometry;       /* 3 x float4 per entry */
  const float4 *attributes;     /* 7 x float4 per entry */
  const float4 *rotation;       /* 3 x float4 (std140 columns) per matrix, 2 matrices per transform */
  const float4 *shift;          /* 1 x float4 per vector, 2 per transform */
  const float *lights;          /* 6 floats per light */
  const uchar4 *atlas[3];
  uint32_t atlas_w[3], atlas_h[3];
  uint32_t n_entries;           /* padded entry count = loop bound (fragment:181-184) */
  uint32_t n_lights;
  uint32_t n_transforms;
  /* Threaded copy of the skip list for the walk kernels (built at upload, flx_api.hip: build_threaded):
   * same entries, same logical visit order, but every entry names its successors explicitly, so the
   * array can be stored hot-first (shallow tree levels in front) and its prefix staged in LDS. */
  const float4 *walk;           /* 3 x float4 per entry, layout below */
  uint32_t walk_entries;        /* entries in `walk` */
  uint32_t walk_hot;            /* the first walk_hot entries are the shallowest ones (LDS candidates) */
  uint32_t walk_root;           /* threaded index of original entry 0 */
  uint32_t walk_fast_boxes;     /* 1: every AABB coordinate is finite with |x| <= 2^59 (precondition of rayCuboidR's fast path) */
  /* Small scenes in one object space: the same entries once more in the reference's own order (every successor lies further
   * on), for the wave-wide lockstep walk (walkLockPass).  lock_entries = 0: there is none. */
  const float4 *lock;           /* 3 x float4 per entry, threaded layout; links are plain indices into `lock` (or WALK_END) */
  uint32_t lock_entries;        /* entries in `lock`, the shared terminator (the last one) included */
  uint32_t lock_root;           /* index of original entry 0 */
  /* Every scene: the same forward-ordered copy of ALL live entries, for the primary rays' walk (primaryWalkF) — the rays of a
   * screen tile visit nearly the same entries, so the wave steps through them together.  (lock == fwd for the small scenes.) */
  const float4 *fwd;
  uint32_t fwd_entries, fwd_root;
  /* Per triangle: clamp(tan(acos(|geometryNormal . n_i|)), 0, 1) for its three vertex normals (fragment:500-512: what the geometry offset of a hit is weighted
   * with).  It depends on the triangle and on its transform only — not on the ray — so it is computed once per scene / transform upload by the same device
   * function (triangleAngleTan, k_angle_tan) instead of by every shade: three acos and three tan in double, a cross product and a normalize per shade.
   * float4 per entry (xyz; boxes: unused); nullptr: shadeSurface computes it itself. */
  const float4 *angle_tan;
};
#ifndef FLX_LOCKSTEP
#define FLX_LOCKSTEP 1
#endif
#define FLX_LOCK_MAX 128u       /* most entries a scene may have to be walked in lockstep */
/* threaded entry:  AABB      e0 = min.xyz max.x | e1 = max.y max.z - - | e2 = bits(nextHit) bits(nextMiss) bits(meta) bits(origIndex)
 *                  triangle  e0 = a.xyz b.x     | e1 = b.yz c.xy      | e2 = c.z        bits(next)     bits(meta) bits(origIndex)
 *                  terminator                                           e2 = -          -              bits(meta = 0)
 * meta = type | transform << 2; successor WALK_END = the loop bound of fragment:184 was reached (no fetch). */
constexpr uint32_t WALK_END = 0x30000000u;      /* kind 3, index 0: as an index it names the threaded copy's shared terminator (entry 0), so a kernel that does not count visits needs
                                                 * no test for it — it fetches the terminator and ends there (walkFetchG); larger than every real index, for the copies whose links are plain indices */
/* A successor link = threaded index | kind of the entry it names << 28 | (that entry's transform differs from this
 * entry's) << 30.  Kinds: 0 terminator, 1 box, 2 triangle, 3 = WALK_END.  The queue scheduler (flx_walkq.hip) routes a
 * walk to its next test from the link alone, without fetching the entry. */
constexpr uint32_t LINK_INDEX = 0x0fffffffu, LINK_XFORM = 0x40000000u;
constexpr int LINK_KIND_SHIFT = 28;
FLX_DEV uint32_t linkIndex(uint32_t link) { return link & LINK_INDEX; }
FLX_DEV uint32_t linkKind(uint32_t link) { return (link >> LINK_KIND_SHIFT) & 3u; }

/* Per-frame constants (flx_frame_params + what the host derives from it). */
/* What may differ between the frames of one batch (flx_render_batch_device): the camera and the per-frame uniforms. */
struct FrameView {
  float camera[3];
  float inv_view[9];            /* row-major inverse of viewMatrix */
  float view_row2[3];
  float ambient[3];
  float random_seed;
};
#define FLX_MAX_BATCH 32
struct DeviceFrame {
  uint32_t width, height;       /* full canvas of ONE frame */
  uint32_t rows;                /* packed rows this context renders: frames x frame_rows */
  uint32_t frame_rows;          /* packed rows of one frame (this context's strips of it) */
  uint32_t frames;              /* frames of the batch, stacked in the packed-row dimension; 1 for a single frame */
  uint32_t tile_rows, tile_index, tile_count;
  int samples, max_reflections;
  int samples_shift;            /* log2(samples) when samples is a power of two, else -1 (item -> tile without a division) */
  float min_importancy;
  int use_filter, is_temporal;
  float texture_width;
  FrameView view[FLX_MAX_BATCH];
};

struct WorkCounters {           /* per-thread tallies, mirrors flx_counters */
  uint32_t primary_visits, closest_visits, shadow_visits, closest_walks, shadow_walks, shades, primary_hits, atlas_texels;
};

FLX_DEV M3 rotation_at(const DeviceScene &sc, int i) {
  float4 a = sc.rotation[3 * i], b = sc.rotation[3 * i + 1], c = sc.rotation[3 * i + 2];
  M3 m; m.c0 = F3(a.x, a.y, a.z); m.c1 = F3(b.x, b.y, b.z); m.c2 = F3(c.x, c.y, c.z);
  return m;
}
FLX_DEV f3 shift_at(const DeviceScene &sc, int i) { float4 s = sc.shift[i]; return F3(s.x, s.y, s.z); }

/* frame of the batch packed row k belongs to */
FLX_DEV uint32_t frame_index(const DeviceFrame &fr, uint32_t k) { return fr.frames > 1u ? k / fr.frame_rows : 0u; }
FLX_DEV f3 view_camera(const FrameView &v) { return F3(v.camera[0], v.camera[1], v.camera[2]); }
FLX_DEV f3 view_ambient(const FrameView &v) { return F3(v.ambient[0], v.ambient[1], v.ambient[2]); }
FLX_DEV f3 frame_camera(const DeviceFrame &fr, uint32_t f) { return view_camera(fr.view[f]); }
FLX_DEV f3 frame_ambient(const DeviceFrame &fr, uint32_t f) { return view_ambient(fr.view[f]); }
/* The views of a frame's batch: the kernel's arguments (LV = false) or a copy the kernel keeps in LDS (LV = true: the chained frame kernel, whose second
 * frame's view arrives while it runs). */
template <bool LV>
FLX_DEV const FrameView &view_at(const DeviceFrame &fr, const FrameView *lv, uint32_t f) { if (LV) return lv[f]; else return fr.view[f]; }
/* image row (0 = top, within its frame) of packed row k under the tile policy */
FLX_DEV uint32_t image_row(const DeviceFrame &fr, uint32_t k) {
  if (fr.frames > 1u) k %= fr.frame_rows;
  if (fr.tile_count <= 1u) return k;
  uint32_t strip = k / fr.tile_rows;
  return (strip * fr.tile_count + fr.tile_index) * fr.tile_rows + (k - strip * fr.tile_rows);
}

/* fragment:91-105 */
FLX_DEV float to4BitRepresentation(float a, float b) {
  uint32_t aui = flx_f2uint(a * 255.0f) & 240u;
  uint32_t bui = (flx_f2uint(b * 255.0f) & 240u) >> 4;
  return (float)(aui | bui) * INV_255;
}
FLX_DEV float normalToSphearical4BitRepresentation(f3 n) {
  float phi = (flx_atan2(n.z, n.x) * INV_PI) * 0.5f + 0.5f;
  float theta = (flx_atan2(n.x, n.y) * INV_PI) * 0.5f + 0.5f;
  return to4BitRepresentation(phi, theta);
}
FLX_DEV f3 combineNormalRME(f3 n, f3 rme) {
  return F3(normalToSphearical4BitRepresentation(n), rme.x, to4BitRepresentation(rme.y, rme.z));
}

/* fragment:108-117; NEAREST + REPEAT lookup in an RGBA8 atlas */
template <bool COUNT>
FLX_DEV f3 fetchTexVal(const DeviceScene &sc, const DeviceFrame &fr, int which, float u, float v, float texNum, f3 defaultVal,
                       WorkCounters &cnt) {
  if (texNum == -1.0f) return defaultVal;
  const uchar4 *atlas = sc.atlas[which];
  uint32_t W = atlas ? sc.atlas_w[which] : 1u, H = atlas ? sc.atlas_h[which] : 1u;
  float tw = fr.texture_width;
  float atlasHeightFactor = (float)W / (float)H;
  float cx = (u + flx_mod(texNum, tw)) / tw;
  float cy = ((v + flx_floor(texNum / tw)) * atlasHeightFactor) / tw;
  if (COUNT) cnt.atlas_texels++;
  if (!atlas) return F3(0.0f, 0.0f, 0.0f);
  float fx = flx_fract(cx) * (float)W, fy = flx_fract(cy) * (float)H;
  uint32_t ix = flx_f2uint(fx), iy = flx_f2uint(fy);
  if (ix >= W) ix = W - 1u;
  if (iy >= H) iy = H - 1u;
  uchar4 t = atlas[(size_t)iy * W + ix];
  return F3((float)t.x / 255.0f, (float)t.y / 255.0f, (float)t.z / 255.0f);
}

/* fragment:119-121 */
FLX_DEV f4 noise(float random_seed, float nx, float ny, float seed) {
  float d = nx * 12.9898f + ny * 78.233f;
  float k = seed + random_seed * PHI;
  f4 r;
  r.x = flx_fract(flx_sin(d + 53.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  r.y = flx_fract(flx_sin(d + 59.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  r.z = flx_fract(flx_sin(d + 61.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  r.w = flx_fract(flx_sin(d + 67.0f * k) * 43758.5453f) * 2.0f - 1.0f;
  return r;
}

/* vote of the wave: the builtin takes the condition as it is (HIP's flx_ballot(int) first materialises it as 0 / 1 and compares
 * again: two VALU instructions per vote, and the walk kernel votes several times per entry) */
FLX_DEV unsigned long long flx_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
/* value of `v` in lane `p` (p uniform) */
FLX_DEV float laneF(float v, uint32_t p) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)p)); }
FLX_DEV int laneI(int v, uint32_t p) { return __builtin_amdgcn_readlane(v, (int)p); }

/* RN(1 / d) without the division: v_rcp_f32 and ONE FMA correction give exactly the bits of 1.0f / d for every float with
 * 2^-60 <= |d| <= 2^60 (tools/micro/rcp_exact.hip walks all 2^32 bit patterns on the MI355X: 0 differences in that range; the
 * division is ~12 dependent instructions).  recipOf() takes that path when every lane of the wave that needs the value is
 * inside the range and divides otherwise; lanes that do not need it may get anything. */
FLX_DEV float recipFast(float d) {
  const float y = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);
}
#ifndef FLX_BRANCH_HINTS
#define FLX_BRANCH_HINTS 1      /* the rare sides of three branches of the stepping loop (IEEE-division fallbacks of 1/det and of the box test, a
                                 * change of object space) laid out of line: dragon 8.07 -> 7.95 ms; hints on the walk's end, on triangle hits
                                 * and on the box test's slow path measured slower again (8.02) — profiles/r02_ab_walk_kernel.txt */
#endif
#if FLX_BRANCH_HINTS
#define FLX_LIKELY(x) __builtin_expect(!!(x), 1)
#define FLX_UNLIKELY(x) __builtin_expect(!!(x), 0)
#else
#define FLX_LIKELY(x) (x)
#define FLX_UNLIKELY(x) (x)
#endif
/* 1 / det of the triangle tests: a lane whose det is `bad` (below BIAS = 2^-16 in magnitude, or in value for the culling test) never reads the result, and one that does has
 * |det| >= 2^-16 — inside recipFast's range from below: only the upper end (and NaN, which is not `bad`) is left to test */
FLX_DEV float recipOfDet(float d, bool bad) {
  const bool ok = bad || flx_abs(d) <= 1.152921504606847e18f;      /* 2^60; NaN is not ok */
  if (FLX_LIKELY(flx_ballot(!ok) == 0ull)) return recipFast(d);
  return 1.0f / d;
}
FLX_DEV float recipOf(float d, bool needed) {
  const float a = flx_abs(d);
  const bool ok = !needed || (a >= 8.673617379884035e-19f && a <= 1.152921504606847e18f);      /* 2^-60, 2^60; NaN is not ok */
  if (FLX_LIKELY(flx_ballot(!ok) == 0ull)) return recipFast(d);
  return 1.0f / d;
}

/* fragment:123-140; returns false on miss, else suv */
FLX_DEV bool moellerTrumbore(f3 a, f3 b, f3 c, const Ray &ray, float l, f3 &suv) {
  f3 edge1 = b - a;
  f3 edge2 = c - a;
  f3 pvec = cross(ray.dir, edge2);
  float det = dot(edge1, pvec);
  if (flx_abs(det) < BIAS) return false;
  float inv_det = recipOf(det, true);
  f3 tvec = ray.origin - a;
  float u = dot(tvec, pvec) * inv_det;
  if (u < BIAS || u > 1.0f) return false;
  f3 qvec = cross(tvec, edge1);
  float v = dot(ray.dir, qvec) * inv_det;
  float uvSum = u + v;
  if (v < BIAS || uvSum > 1.0f) return false;
  float s = dot(edge2, qvec) * inv_det;
  if (s > l || s <= BIAS) return false;
  suv = F3(s, u, v);
  return s != 0.0f;             /* fragment:217 tests intersection.x != 0.0 */
}

/* fragment:143-158 */
FLX_DEV bool moellerTrumboreCull(f3 a, f3 b, f3 c, const Ray &ray, float l) {
  f3 edge1 = b - a;
  f3 edge2 = c - a;
  f3 pvec = cross(ray.dir, edge2);
  float det = dot(edge1, pvec);
  float invDet = recipOf(det, !(det < BIAS));        /* (the shader divides before the test; a rejected lane never uses the value) */
  if (det < BIAS) return false;
  f3 tvec = ray.origin - a;
  float u = dot(tvec, pvec) * invDet;
  if (u < BIAS || u > 1.0f) return false;
  f3 qvec = cross(tvec, edge1);
  float v = dot(ray.dir, qvec) * invDet;
  if (v < BIAS || u + v > 1.0f) return false;
  float s = dot(edge2, qvec) * invDet;
  return (s <= l && s > BIAS);
}

/* Primary-visibility triangle rule: front faces only, inclusive edges, near plane. */
/* (the edges b - a, c - a are given: the threaded copy stores them) */
FLX_DEV bool moellerTrumborePrimaryE(f3 a, f3 edge1, f3 edge2, const Ray &ray, float l, float viewDepthPerS, f3 &suv) {
  f3 pvec = cross(ray.dir, edge2);
  float det = dot(edge1, pvec);
  if (!(det < 0.0f)) return false;
  float inv_det = recipOf(det, true);
  f3 tvec = ray.origin - a;
  float u = dot(tvec, pvec) * inv_det;
  if (!(u >= 0.0f && u <= 1.0f)) return false;
  f3 qvec = cross(tvec, edge1);
  float v = dot(ray.dir, qvec) * inv_det;
  if (!(v >= 0.0f && u + v <= 1.0f)) return false;
  float s = dot(edge2, qvec) * inv_det;
  if (!(s < l) || !(s * viewDepthPerS >= NEAR_VIEW_DEPTH)) return false;
  suv = F3(s, u, v);
  return s != 0.0f;
}

/* fragment:161-167 */
FLX_DEV bool rayCuboid(float l, const Ray &ray, f3 minCorner, f3 maxCorner) {
  f3 v0 = (minCorner - ray.origin) / ray.dir;
  f3 v1 = (maxCorner - ray.origin) / ray.dir;
  float tmin = flx_max(flx_max(flx_min(v0.x, v1.x), flx_min(v0.y, v1.y)), flx_min(v0.z, v1.z));
  float tmax = flx_min(flx_min(flx_max(v0.x, v1.x), flx_max(v0.y, v1.y)), flx_max(v0.z, v1.z));
  return tmax >= flx_max(tmin, BIAS) && tmin < l;
}

/* fragment:282-302 */
FLX_DEV float trowbridgeReitz(float alpha, float NdotH) {
  float numerator = alpha * alpha;
  float denom = NdotH * NdotH * (numerator - 1.0f) + 1.0f;
  return numerator / flx_max(PI_F * denom * denom, BIAS);
}
FLX_DEV float schlickBeckmann(float alpha, float NdotX) {
  float k = alpha * 0.5f;
  float denominator = NdotX * (1.0f - k) + k;
  denominator = flx_max(denominator, BIAS);
  return NdotX / denominator;
}
FLX_DEV float smith(float alpha, float NdotV, float NdotL) { return schlickBeckmann(alpha, NdotV) * schlickBeckmann(alpha, NdotL); }
FLX_DEV f3 fresnel(f3 F0, float theta) {
  float p = flx_pow5(1.0f - theta);
  return F3(F0.x + (1.0f - F0.x) * p, F0.y + (1.0f - F0.y) * p, F0.z + (1.0f - F0.z) * p);
}

/* fragment:304-334 */
FLX_DEV f3 forwardTrace(const Material &material, f3 lightDir, float strength, f3 N, f3 V) {
  float lenP1 = 1.0f + length(lightDir);
  float brightness = strength / (lenP1 * lenP1);
  f3 L = normalize(lightDir);
  f3 H = normalize(V + L);
  float VdotH = flx_max(dot(V, H), 0.0f);
  float NdotL = flx_max(dot(N, L), 0.0f);
  float NdotH = flx_max(dot(N, H), 0.0f);
  float NdotV = flx_max(dot(N, V), 0.0f);
  float alpha = material.rme.x * material.rme.x;
  float BRDF = flx_mix(1.0f, NdotV, material.rme.y);
  f3 F0 = material.albedo * BRDF;
  f3 Ks = fresnel(F0, VdotH);
  float oneMinusMetal = 1.0f - material.rme.y;
  f3 Kd = F3((1.0f - Ks.x) * oneMinusMetal, (1.0f - Ks.y) * oneMinusMetal, (1.0f - Ks.z) * oneMinusMetal);
  f3 lambert = material.albedo * INV_PI;
  float tr = trowbridgeReitz(alpha, NdotH);
  float sm = smith(alpha, NdotV, NdotL);
  f3 cookTorranceNumerator = (Ks * tr) * sm;
  float cookTorranceDenominator = 4.0f * NdotV * NdotL;
  cookTorranceDenominator = flx_max(cookTorranceDenominator, BIAS);
  f3 cookTorrance = cookTorranceNumerator / cookTorranceDenominator;
  f3 radiance = Kd * lambert + cookTorrance;
  return (radiance * NdotL) * brightness;
}

/* The shader's global variables + MRT outputs that survive across samples (fragment:83-89, 74-79). */
struct PixelState {
  float firstRayLength, glassFilter, originalRMEx, originalTPOx;
  f3 originalColor;
  f4 renderId, renderOriginalId;
  float ndc_x, ndc_y;
  float seed;                   /* randomSeed of the pixel's frame */
  uint32_t lightBase = 0;       /* floats in front of this frame's lights in DeviceScene::lights (the frame server with a scene that moves keeps a version per frame slot: flx_server.hip; else 0) */
};

/* State of one path between bounces (what lightTrace keeps in locals, fragment:464-474). */
struct PathState {
  Ray ray;
  f3 lastHitPoint;
  f3 finalColor, importancyFactor;
  Hit hit;
  bool dontFilter;
};

/* What one bounce's shading hands to the traversal stage: the reservoir's light sample
 * (fragment:400-461 up to the shadowTest call) and the two rays to walk. */
struct ShadeOut {
  f3 litColor;          /* localColor + baseLuminance: returned when the picked light is visible */
  f3 baseLuminance;     /* returned when it is shadowed */
  Ray shadowRay;        /* fragment:452-453 */
  float shadowLen;      /* length(reservoirLightDir), fragment:455 */
  bool needShadow;      /* false: showColor / showShadow decided it without a walk (fragment:438-450) */
  bool shadowedNoWalk;  /* the showShadow early-out */
  bool markId;          /* dontFilter || i == 0: renderId.w is written by this bounce */
};

/* fragment:400-453: everything of reservoirSample before the shadow ray is walked. */
FLX_DEV void reservoirPick(const DeviceScene &sc, const DeviceFrame &fr, PixelState &ps, const Material &material, const Ray &ray,
                           f4 randomVec, f3 N, f3 smoothNormal, float geometryOffset, bool dontFilter, int i, ShadeOut &so) {
  f3 localColor = F3(0.0f, 0.0f, 0.0f);
  float reservoirLength = 0.0f;
  float totalWeight = 0.0f;
  int reservoirNum = 0;
  float reservoirWeight = 0.0f;
  f3 reservoirLightDir = F3(0.0f, 0.0f, 0.0f);
  f4 n0 = noise(ps.seed, randomVec.z, randomVec.w, BIAS);
  float lastRandomX = n0.x, lastRandomY = n0.y;
  const int size = (int)sc.n_lights;
  for (int j = 0; j < size; j++) {
    const float *lt = sc.lights + 6 * j + ps.lightBase;
    float strength = lt[3], variation = lt[4];
    if (strength <= 0.0f) continue;
    reservoirLength += 1.0f;
    f3 light = F3(lt[0], lt[1], lt[2]) + F3(randomVec.x, randomVec.y, randomVec.z) * variation;
    f3 dir = light - ray.origin;
    f3 colorForLight = forwardTrace(material, dir, strength, N, -ray.dir);
    localColor = localColor + colorForLight;
    float weight = length(colorForLight);
    totalWeight += weight;
    if (flx_abs(lastRandomY) * totalWeight <= weight) {
      reservoirNum = j;
      reservoirWeight = weight;
      reservoirLightDir = dir;
    }
    if (j + 1 < size) {                /* the draw feeds the next light's test (fragment:433): after the last light nobody reads it */
      f4 n1 = noise(ps.seed, lastRandomX, lastRandomY, BIAS);
      lastRandomX = n1.z; lastRandomY = n1.w;
    }
  }
  f3 unitLightDir = normalize(reservoirLightDir);
  bool showColor = reservoirLength == 0.0f || reservoirWeight == 0.0f;
  bool showShadow = dot(smoothNormal, unitLightDir) <= BIAS;
  so.baseLuminance = F3(material.rme.z, material.rme.z, material.rme.z);
  so.litColor = localColor + so.baseLuminance;
  so.markId = dontFilter || i == 0;
  if (so.markId) ps.renderId.w = (float)((reservoirNum % 128) << 1) * INV_255;
  so.needShadow = !showColor && !showShadow;
  so.shadowedNoWalk = !showColor && showShadow;
  so.shadowRay.origin = ray.origin + smoothNormal * geometryOffset;
  so.shadowRay.dir = unitLightDir;
  so.shadowLen = length(reservoirLightDir);
}

/* One iteration of lightTrace's bounce loop up to its two traversals (fragment:476-589): surface
 * fetch, material, RNG, Fresnel choice, filter bookkeeping, light pick, next direction.  On return
 * p.ray is the next ray to walk with rayTracer (fragment:591) and `so` describes the shadow ray. */
/* What a bounce's shading knows about the surface it landed on before any random number is drawn (fragment:476-533): the
 * same for every sample of a pixel at bounce 0, where all samples share the primary hit — k_wf_shade0 computes it once per
 * pixel (it holds the three acos / tan of the normal deviation, the attribute fetch and the material). */
struct SurfaceCtx {
  f3 origin;                 /* the hit point: the new ray origin */
  f3 dir;                    /* normalize(hit point - last hit point) */
  f3 smoothNormal;           /* facing the ray */
  float signDir, geometryOffset, BRDF, roughnessBRDF;
  f3 F0;
  Material material;
};

/* Who reads the per-triangle table (DeviceScene::angle_tan) is decided per source file: the per-pixel kernel gains 4.5 % from it; the persistent kernels — whose paths' chains
 * walk -> shade -> walk set the pace of a thin share — lose to the extra dependent load (a rank's eighth through the frame server with two frames in flight 1.235 -> 1.351 ms,
 * k_paths + 1 %: profiles/r04_angle_table.txt) and keep computing the terms.  (As FLX_SINCOS_TABLE: one definition per translation unit.) */
#ifndef FLX_ANGLE_TABLE
#define FLX_ANGLE_TABLE 0
#endif
/* fragment:500-512 from the triangle's transformed vertices and vertex normals: independent of the ray (DeviceScene::angle_tan) */
FLX_DEV f3 triangleAngleTan(f3 t0v, f3 t1v, f3 t2v, f3 n0, f3 n1, f3 n2) {
  f3 geometryNormal = normalize(cross(t0v - t1v, t0v - t2v));
  f3 angles = F3(flx_acos(flx_abs(dot(geometryNormal, n0))), flx_acos(flx_abs(dot(geometryNormal, n1))),
                 flx_acos(flx_abs(dot(geometryNormal, n2))));
  return F3(flx_clamp(flx_tan(angles.x), 0.0f, 1.0f), flx_clamp(flx_tan(angles.y), 0.0f, 1.0f),
            flx_clamp(flx_tan(angles.z), 0.0f, 1.0f));
}
/* ... for entry `tri` of the scene, as shadeSurface forms its operands (the entry's own transform: fragment:478-499) */
FLX_DEV f3 triangleAngleTanOf(const DeviceScene &sc, int tri) {
  const float4 g0 = sc.geometry[3 * tri], g1 = sc.geometry[3 * tri + 1], g2 = sc.geometry[3 * tri + 2];
  const M3 rTI = rotation_at(sc, (int)g2.y << 1);
  const float4 *at = sc.attributes + 7 * (size_t)tri;
  const float4 a0 = at[0], a1 = at[1], a2 = at[2];
  return triangleAngleTan(mul(rTI, F3(g0.x, g0.y, g0.z)), mul(rTI, F3(g0.w, g1.x, g1.y)), mul(rTI, F3(g1.z, g1.w, g2.x)),
                          mul(rTI, F3(a0.x, a0.y, a0.z)), mul(rTI, F3(a0.w, a1.x, a1.y)), mul(rTI, F3(a1.z, a1.w, a2.x)));
}

template <bool COUNT>
FLX_DEV void shadeSurface(const DeviceScene &sc, const DeviceFrame &fr, const Hit &hit, const Ray &ray, f3 lastHitPoint, SurfaceCtx &sf,
                          WorkCounters &cnt) {
  if (COUNT) cnt.shades++;
  M3 rTI = rotation_at(sc, hit.transformId);
  f3 sTI = shift_at(sc, hit.transformId);
  sf.origin = ray.dir * hit.suv.x + ray.origin;
  f3 uvw = F3(1.0f - hit.suv.y - hit.suv.z, hit.suv.y, hit.suv.z);
  float4 g0 = sc.geometry[3 * hit.triangleId], g1 = sc.geometry[3 * hit.triangleId + 1], g2 = sc.geometry[3 * hit.triangleId + 2];
  f3 t0v = mul(rTI, F3(g0.x, g0.y, g0.z));
  f3 t1v = mul(rTI, F3(g0.w, g1.x, g1.y));
  f3 t2v = mul(rTI, F3(g1.z, g1.w, g2.x));
  f3 offsetRayTarget = sf.origin - sTI;
  f3 diffs = F3(distance(offsetRayTarget, t0v), distance(offsetRayTarget, t1v), distance(offsetRayTarget, t2v));
  const float4 *at = sc.attributes + 7 * (size_t)hit.triangleId;
  float4 a0 = at[0], a1 = at[1], a2 = at[2], a3 = at[3], a4 = at[4], a5 = at[5], a6 = at[6];
  f3 n0 = mul(rTI, F3(a0.x, a0.y, a0.z));
  f3 n1 = mul(rTI, F3(a0.w, a1.x, a1.y));
  f3 n2 = mul(rTI, F3(a1.z, a1.w, a2.x));
  f3 smoothNormal = normalize(F3((n0.x * uvw.x + n1.x * uvw.y) + n2.x * uvw.z,
                                 (n0.y * uvw.x + n1.y * uvw.y) + n2.y * uvw.z,
                                 (n0.z * uvw.x + n1.z * uvw.y) + n2.z * uvw.z));
  f3 angleTan;
  if (FLX_ANGLE_TABLE && sc.angle_tan) { const float4 t = sc.angle_tan[hit.triangleId]; angleTan = F3(t.x, t.y, t.z); }      /* the same floats, computed at the upload (k_angle_tan) */
  else angleTan = triangleAngleTan(t0v, t1v, t2v, n0, n1, n2);
  sf.geometryOffset = dot(diffs * angleTan, uvw);
  /* uv0 = a2.yz, uv1 = (a2.w, a3.x), uv2 = a3.yz */
  float bu = (a2.y * uvw.x + a2.w * uvw.y) + a3.y * uvw.z;
  float bv = (a2.z * uvw.x + a3.x * uvw.y) + a3.z * uvw.z;
  sf.material.albedo = fetchTexVal<COUNT>(sc, fr, 0, bu, bv, a3.w, F3(a4.z, a4.w, a5.x), cnt);
  sf.material.rme = fetchTexVal<COUNT>(sc, fr, 1, bu, bv, a4.x, F3(a5.y, a5.z, a5.w), cnt);
  sf.material.tpo = fetchTexVal<COUNT>(sc, fr, 2, bu, bv, a4.y, F3(a6.x, a6.y, a6.z), cnt);

  sf.dir = normalize(sf.origin - lastHitPoint);
  sf.signDir = flx_sign(dot(sf.dir, smoothNormal));
  sf.smoothNormal = smoothNormal * (-sf.signDir);
  sf.BRDF = flx_mix(1.0f, flx_abs(dot(sf.smoothNormal, sf.dir)), sf.material.rme.y);
  sf.roughnessBRDF = sf.material.rme.x * sf.BRDF;
  sf.F0 = sf.material.albedo * sf.BRDF;
}

/* The rest of the bounce (fragment:535-589): the random vector of this (pixel, sample, bounce), Fresnel choice, filter
 * bookkeeping, light pick, next direction. */
FLX_DEV void shadeSample(const DeviceScene &sc, const DeviceFrame &fr, const SurfaceCtx &sf, PixelState &ps, PathState &p, f3 camera, float cosSampleN,
                         int i, ShadeOut &so) {
  const float fi = (float)i;
  const Material &material = sf.material;
  const f3 smoothNormal = sf.smoothNormal;
  p.ray.origin = sf.origin;
  p.ray.dir = sf.dir;
  f4 randomVec = noise(ps.seed, ps.ndc_x, ps.ndc_y, fi + cosSampleN);
  f3 randomSpheareVec = normalize(smoothNormal + normalize(F3(randomVec.x, randomVec.y, randomVec.z)));
  f3 roughNormal = normalize(mix(smoothNormal, randomSpheareVec, sf.roughnessBRDF));
  f3 H = normalize(roughNormal - p.ray.dir);
  float VdotH = flx_max(dot(-p.ray.dir, H), 0.0f);
  f3 f = fresnel(sf.F0, VdotH);
  float fresnelReflect = flx_max(f.x, flx_max(f.y, f.z));
  bool isSolid = material.tpo.x * fresnelReflect <= flx_abs(randomVec.w);

  if (p.dontFilter) {
    ps.originalTPOx = material.tpo.x;
    ps.originalColor = ps.originalColor * material.albedo;
    ps.originalRMEx += material.rme.x;
    if (fr.use_filter) {               /* renderId feeds only the filter's G-buffer */
      float scale = flx_exp2_neg_int(i);
      f3 cn = combineNormalRME(smoothNormal, material.rme);
      f4 upd; upd.x = scale * cn.x; upd.y = scale * cn.y; upd.z = scale * cn.z; upd.w = scale * 0.0f;
      ps.renderId.x += upd.x; ps.renderId.y += upd.y; ps.renderId.z += upd.z; ps.renderId.w += upd.w;
      if (i == 0) {
        ps.renderOriginalId.x += upd.x; ps.renderOriginalId.y += upd.y; ps.renderOriginalId.z += upd.z; ps.renderOriginalId.w += upd.w;
      }
    }
    p.dontFilter = (material.rme.x < 0.01f && isSolid) || !isSolid;
    if (isSolid && material.tpo.x > 0.01f) {
      ps.glassFilter += 1.0f;
      p.dontFilter = false;
    }
  } else {
    p.importancyFactor = p.importancyFactor * material.albedo;
  }

  if (i == 1) ps.firstRayLength = flx_min(length(p.ray.origin - p.lastHitPoint) / length(p.lastHitPoint - camera), ps.firstRayLength);
  reservoirPick(sc, fr, ps, material, p.ray, randomVec, roughNormal * (-sf.signDir), smoothNormal * (-sf.signDir), sf.geometryOffset, p.dontFilter, i, so);
  if (isSolid) {
    p.ray.dir = normalize(mix(reflect(p.ray.dir, smoothNormal), randomSpheareVec, sf.roughnessBRDF));
  } else {
    float eta = flx_mix(1.0f / material.tpo.z, material.tpo.z, flx_max(sf.signDir, 0.0f));
    p.ray.dir = normalize(mix(refract(p.ray.dir, smoothNormal, eta), randomSpheareVec, sf.roughnessBRDF));
  }
}

template <bool COUNT>
FLX_DEV void bounceShade(const DeviceScene &sc, const DeviceFrame &fr, PixelState &ps, PathState &p, f3 camera, float cosSampleN, int i,
                         ShadeOut &so, WorkCounters &cnt) {
  SurfaceCtx sf;
  shadeSurface<COUNT>(sc, fr, p.hit, p.ray, p.lastHitPoint, sf, cnt);
  shadeSample(sc, fr, sf, ps, p, camera, cosSampleN, i, so);
}

/* State of one skip-list walk, advanced one entry per walkStep(): the loop bodies of rayTracer
 * (fragment:184-224) and shadowTest (fragment:240-277) with their loop variables made explicit so
 * that a lane can suspend / resume a walk and the scheduler above can refill idle lanes. */
struct WalkState {
  Ray src;            /* the ray in world space */
  Ray tR;             /* the ray in the object space of transform cachedTI (fragment:174,233) */
  float minLen;       /* fragment:179 / :236 */
  int i;              /* next entry */
  int cachedTI;
  int mode;           /* 0 shadowTest, 1 rayTracer, 2 finished */
  int shadowed;       /* result of the shadow walk (0 / 1; an int so that a select, not a mask round trip, updates it) */
  f3 suv;             /* closest hit so far */
  int tri, hitTI;     /* entry index (-1: none) and 2 * transform number of the closest hit */
  f3 inv;             /* RN(1 / tR.dir), for the threaded walk's box test */
  bool fastDiv;       /* tR is in the range where divByRecip() is proven exact */
};

FLX_DEV void walkStart(WalkState &w, int mode, const Ray &ray, float len) {
  w.mode = mode; w.src = ray; w.tR = ray; w.cachedTI = 0; w.minLen = len; w.i = 0;
}
FLX_DEV void walkClearResults(WalkState &w) {
  w.inv = F3(0.0f, 0.0f, 0.0f); w.fastDiv = false;
  w.shadowed = false; w.suv = F3(0.0f, 0.0f, 0.0f); w.tri = -1; w.hitTI = 0;
}

/* Visit entry w.i.  Returns true when the current walk (shadow or closest) has ended. */
template <bool COUNT>
FLX_DEV bool walkStep(const DeviceScene &sc, WalkState &w, WorkCounters &cnt) {
  bool endWalk = false;
  const int i = w.i;
  float4 e0 = sc.geometry[3 * i], e1 = sc.geometry[3 * i + 1], e2 = sc.geometry[3 * i + 2];
  if (COUNT) { if (w.mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; }
  int tI = (int)e2.y << 1;
  if (tI != w.cachedTI) {
    int iI = tI + 1;
    M3 rotationII = rotation_at(sc, iI);
    w.cachedTI = tI;
    w.tR.origin = mul(rotationII, w.src.origin + shift_at(sc, iI));
    f3 d = mul(rotationII, w.src.dir);
    w.tR.dir = (w.mode == 0) ? normalize(d) : d;      /* fragment:261 normalises, fragment:201 does not */
  }
  int next = i + 1;
  if (e2.z == 0.0f) {
    endWalk = true;
  } else if (e2.z == 1.0f) {
    if (!rayCuboid(w.minLen, w.tR, F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y))) next += (int)e1.z;
  } else {
    f3 a = F3(e0.x, e0.y, e0.z), b = F3(e0.w, e1.x, e1.y), c = F3(e1.z, e1.w, e2.x);
    if (w.mode == 0) {
      if (moellerTrumboreCull(a, b, c, w.tR, w.minLen)) { w.shadowed = true; endWalk = true; }
    } else {
      f3 suv;
      if (moellerTrumbore(a, b, c, w.tR, w.minLen, suv)) {
        w.suv = suv; w.hitTI = tI; w.tri = i;
        w.minLen = suv.x;
      }
    }
  }
  w.i = next;
  return endWalk || next >= (int)sc.n_entries;
}

/* The current entry of a walk, held in registers. */
struct WalkEntry { float4 e0, e1, e2; };

/* ---- the same walk over the threaded copy (DeviceScene::walk) ---------------------------------------
 * w.i is a threaded index; `lds` holds the first ldsCount entries (3 float4 each) or is null. */

/* a / d from y = RN(1/d) with two FMA corrections (Markstein): q0 = RN(a*y) is within ~2 ulp of a/d; the
 * first correction q1 = RN(q0 + (a - q0*d)*y) is the rounding of a value within 2^-47 relative of a/d,
 * hence faithful; with y = RN(1/d) and q1 faithful, q2 = RN(q1 + (a - q1*d)*y) IS RN(a/d) (Markstein's
 * theorem; the residuals are exact in an FMA).  Preconditions, which keep every intermediate normal:
 * 2^-60 <= |d| <= 2^60, |a| <= 2^60, and a == 0 or |a| >= 2^-40.  Outside them callers divide.
 * tools/divtest.c compares it with `/` on 5e8 random and near-midpoint operands: 0 mismatches. */
FLX_DEV float divByRecip(float a, float d, float y) {
  float q = a * y;
  float r = __builtin_fmaf(-q, d, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-q, d, a);
  return __builtin_fmaf(r, y, q);
}
/* tR changed: refresh the reciprocal and decide whether the fast box test may be used for this ray. */
/* RN(1/d) per component and whether (d, o) are in the range where divByRecip() is proven exact. */
FLX_DEV void reciprocalOfDir(const DeviceScene &sc, f3 d, f3 o, f3 &inv, bool &fast) {
  inv = F3(recipFast(d.x), recipFast(d.y), recipFast(d.z));      /* read only when `fast` holds, i.e. inside recipFast's proven range */
  const float LO = 8.673617379884035e-19f, HI = 1.152921504606847e18f, OHI = 5.764607523034235e17f;   /* 2^-60, 2^60, 2^59 */
  const float ax = flx_abs(d.x), ay = flx_abs(d.y), az = flx_abs(d.z);
  fast = sc.walk_fast_boxes != 0u && ax >= LO && ax <= HI && ay >= LO && ay <= HI && az >= LO && az <= HI &&
         flx_abs(o.x) <= OHI && flx_abs(o.y) <= OHI && flx_abs(o.z) <= OHI;
}
#ifndef FLX_WF_RECIP_DIV
#define FLX_WF_RECIP_DIV 1     /* k_wf_walk (rays transformed on the fly) takes the reciprocal box test too: with recipFast() the three
                                * reciprocals per transform change cost 9 instructions (9-transform scene: 19.6 -> 18.9 ms) */
#endif
FLX_DEV void walkPrepareRay(const DeviceScene &sc, WalkState &w) {
#if !FLX_WF_RECIP_DIV
  (void)sc; (void)w;
  return;
#endif
  reciprocalOfDir(sc, w.tR.dir, w.tR.origin, w.inv, w.fastDiv);
}
/* rayCuboid (fragment:161-167) with the six quotients taken through divByRecip when its preconditions
 * hold (same bits as the division, so the same boolean), through `/` otherwise (zero, denormal, huge,
 * infinite or NaN components: the reference's behaviour there is whatever IEEE division gives). */
FLX_DEV bool rayCuboidRecip(float l, const WalkState &w, f3 minCorner, f3 maxCorner);
FLX_DEV bool rayCuboidR(float l, const WalkState &w, f3 minCorner, f3 maxCorner) {
#if !FLX_WF_RECIP_DIV
  return rayCuboid(l, w.tR, minCorner, maxCorner);
#else
  return rayCuboidRecip(l, w, minCorner, maxCorner);
#endif
}
FLX_DEV bool rayCuboidRecip(float l, const WalkState &w, f3 minCorner, f3 maxCorner) {
  const f3 o = w.tR.origin, d = w.tR.dir, y = w.inv;
  const f3 a0 = minCorner - o, a1 = maxCorner - o;
  /* every |a| is 0 or >= 2^-40.  Exact zeros are common (a ray leaving a flat floor has its origin ON the plane of the
   * floor's degenerate box) and must stay on the fast path: a float min over |a| that sends zeros to the division path
   * measured 1.5 % slower on the dragon frame. */
  uint32_t m = (flx_f2u(a0.x) << 1) - 2u;                          /* (bits << 1) - 2 as unsigned drops the sign, is huge for +-0 and small */
  uint32_t t;                                                      /* for tiny values: one v_lshl_add_u32 per value */
  t = (flx_f2u(a0.y) << 1) - 2u; m = t < m ? t : m;
  t = (flx_f2u(a0.z) << 1) - 2u; m = t < m ? t : m;
  t = (flx_f2u(a1.x) << 1) - 2u; m = t < m ? t : m;
  t = (flx_f2u(a1.y) << 1) - 2u; m = t < m ? t : m;
  t = (flx_f2u(a1.z) << 1) - 2u; m = t < m ? t : m;
  const bool aOk = m >= (0x2b800000u << 1) - 2u;
  f3 v0, v1;
  float tmin, tmax;
  if ((w.fastDiv && aOk)) {
    v0 = F3(divByRecip(a0.x, d.x, y.x), divByRecip(a0.y, d.y, y.y), divByRecip(a0.z, d.z, y.z));
    v1 = F3(divByRecip(a1.x, d.x, y.x), divByRecip(a1.y, d.y, y.y), divByRecip(a1.z, d.z, y.z));
    /* all six quotients are finite here, so GLSL min/max ((y < x) ? y : x) and the hardware's v_min/v_max agree except
     * for the sign of a zero result, which the comparisons below cannot see: one instruction each instead of
     * compare + select */
    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(v0.x, v1.x), __builtin_fminf(v0.y, v1.y)), __builtin_fminf(v0.z, v1.z));
    tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(v0.x, v1.x), __builtin_fmaxf(v0.y, v1.y)), __builtin_fmaxf(v0.z, v1.z));
    return tmax >= __builtin_fmaxf(tmin, BIAS) && tmin < l;
  }
  v0 = a0 / d;
  v1 = a1 / d;
  tmin = flx_max(flx_max(flx_min(v0.x, v1.x), flx_min(v0.y, v1.y)), flx_min(v0.z, v1.z));
  tmax = flx_min(flx_min(flx_max(v0.x, v1.x), flx_max(v0.y, v1.y)), flx_max(v0.z, v1.z));
  return tmax >= flx_max(tmin, BIAS) && tmin < l;
}

/* Primary visibility (the walk of fragment:172-227 with the primary triangle rule) over the threaded copy: explicit successors, stored edges, the box test through the
 * exact reciprocal division.  The entries a ray visits, their order, the arithmetic of every test and the visit count are
 * those of the shader's loop; a pixel's ray changes object space a few times per walk, so that is done in place. */
FLX_DEV bool rayCuboidFast(float l, const WalkState &w, f3 lo, f3 hi);      /* below: the interval test with the exact quotients as its fallback */
FLX_DEV Hit primaryWalkT(const DeviceScene &sc, const Ray &ray, float viewDepthPerS, uint32_t &visits) {
  Hit hit; hit.suv = F3(0.0f, 0.0f, 0.0f); hit.transformId = 0; hit.triangleId = -1;
  WalkState w;
  w.tR = ray; w.minLen = POW32;
  reciprocalOfDir(sc, ray.dir, ray.origin, w.inv, w.fastDiv);
  int cachedTI = 0;
  uint32_t link = sc.walk_root;
  while (link != WALK_END) {
    const size_t i = (size_t)linkIndex(link) * 3u;
    const float4 e0 = sc.walk[i], e1 = sc.walk[i + 1], e2 = sc.walk[i + 2];
    visits++;
    const int meta = __float_as_int(e2.z);
    if ((meta & 3) == 0) break;                           /* terminator (its fetch counts, fragment:208) */
    const int tI = (meta >> 2) << 1;
    if (tI != cachedTI) {
      const int iI = tI + 1;
      const M3 rotationII = rotation_at(sc, iI);
      cachedTI = tI;
      w.tR.origin = mul(rotationII, ray.origin + shift_at(sc, iI));
      w.tR.dir = mul(rotationII, ray.dir);
      reciprocalOfDir(sc, w.tR.dir, w.tR.origin, w.inv, w.fastDiv);
    }
    if ((meta & 3) == 1) {
      link = (uint32_t)__float_as_int(rayCuboidFast(w.minLen, w, F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y)) ? e2.x : e2.y);
    } else {
      f3 suv;
      if (moellerTrumborePrimaryE(F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y), F3(e1.z, e1.w, e2.x), w.tR, w.minLen, viewDepthPerS, suv)) {
        hit.suv = suv; hit.transformId = tI; hit.triangleId = __float_as_int(e2.w);
        w.minLen = suv.x;
      }
      link = (uint32_t)__float_as_int(e2.y);
    }
  }
  return hit;
}

typedef float flx_v4f_ __attribute__((ext_vector_type(4)));
/* Primary visibility by the wave.  The 64 rays of an 8 x 8 screen tile leave one point in nearly one direction: they visit nearly
 * the same entries (dragon 1080p: 18 per ray, 24 per tile, tests/analysis/primary_union.py).  Over the forward-ordered copy
 * (sc.fwd: every successor lies further on) the wave visits the lowest entry any of its rays stands at — a scalar load, one kind of
 * test per trip, no per-lane fetch — and the rays that stand there test it.  Where the rays have spread (deep inside the
 * dragon's tree a trip serves a handful of them) the lanes finish on their own over the same array.  Per ray: the entries,
 * their order, the arithmetic and the visit count of primaryWalkT. */
extern "C" __device__ unsigned int __ockl_wfred_min_u32(unsigned int);
#ifndef FLX_PRIMARY_PREFETCH
#define FLX_PRIMARY_PREFETCH 0       /* fetch both successors of an entry while it is tested: measured slower (k_primary 0.455 against 0.391 ms) */
#endif
#ifndef FLX_PRIMARY_LOCK_MIN
#define FLX_PRIMARY_LOCK_MIN 12      /* two trips in a row that serve fewer rays than this: the lanes go on alone */
#endif
typedef __attribute__((address_space(4))) const flx_v4f_ fwd_cf4;
FLX_DEV void primaryVisit(const DeviceScene &sc, const Ray &ray, float viewDepthPerS, WalkState &w, int &cachedTI, Hit &hit, uint32_t &nxt,
                          float4 e0, float4 e1, float4 e2, int xfBase = 0 /* matrices in front of this frame's version of the transforms (PixelState::lightBase) */) {
  const int meta = __float_as_int(e2.z);
  if ((meta & 3) == 0) { nxt = WALK_END; return; }          /* terminator (its fetch counts, fragment:208) */
  const int tI = (meta >> 2) << 1;
  if (tI != cachedTI) {
    const int iI = tI + 1 + xfBase;
    const M3 rotationII = rotation_at(sc, iI);
    cachedTI = tI;
    w.tR.origin = mul(rotationII, ray.origin + shift_at(sc, iI));
    w.tR.dir = mul(rotationII, ray.dir);
    reciprocalOfDir(sc, w.tR.dir, w.tR.origin, w.inv, w.fastDiv);
  }
  if ((meta & 3) == 1) {
    nxt = (uint32_t)__float_as_int(rayCuboidFast(w.minLen, w, F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y)) ? e2.x : e2.y);
  } else {
    f3 suv;
    if (moellerTrumborePrimaryE(F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y), F3(e1.z, e1.w, e2.x), w.tR, w.minLen, viewDepthPerS, suv)) {
      hit.suv = suv; hit.transformId = tI; hit.triangleId = __float_as_int(e2.w);
      w.minLen = suv.x;
    }
    nxt = (uint32_t)__float_as_int(e2.y);
  }
}
/* every lane of the wave calls this (active = the lane has a pixel) */
FLX_DEV Hit primaryWalkF(const DeviceScene &sc, bool active, const Ray &ray, float viewDepthPerS, uint32_t &visits, int xfBase = 0) {
  Hit hit; hit.suv = F3(0.0f, 0.0f, 0.0f); hit.transformId = 0; hit.triangleId = -1;
  WalkState w;
  w.tR = ray; w.minLen = POW32;
  reciprocalOfDir(sc, ray.dir, ray.origin, w.inv, w.fastDiv);
  int cachedTI = 0;
  uint32_t nxt = active ? sc.fwd_root : WALK_END;
  const fwd_cf4 *L = (const fwd_cf4 *)sc.fwd;
  uint32_t thin = 0;
  for (;;) {
    const uint32_t i = __builtin_amdgcn_readfirstlane(__ockl_wfred_min_u32(nxt));
    if (i == WALK_END) return hit;
    uint32_t iu = i;
    asm volatile("" : "+s"(iu));                             /* (a scalar the compiler cannot trade for the lane's own `nxt`: scalar loads) */
    const fwd_cf4 *E = L + (size_t)iu * 3u;
    const flx_v4f_ a = E[0], b = E[1], c = E[2];
    const bool mine = nxt == i;
    if (mine) {
      visits++;
      primaryVisit(sc, ray, viewDepthPerS, w, cachedTI, hit, nxt, make_float4(a.x, a.y, a.z, a.w), make_float4(b.x, b.y, b.z, b.w), make_float4(c.x, c.y, c.z, c.w), xfBase);
    }
    thin = (uint32_t)__popcll(flx_ballot(mine)) < (uint32_t)FLX_PRIMARY_LOCK_MIN ? thin + 1u : 0u;
    if (thin >= 2u) break;
  }
  /* The lanes on their own, over the same array.  What is left are the long walks: a ray that grazes the dragon visits several
   * hundred entries (tests/analysis/primary_union.py: the longest ray of a tile visits 7 entries at the median, 257 at the 99th
   * percentile, 477 at most).  Fetching both possible successors of an entry while it is tested (FLX_PRIMARY_PREFETCH) does not
   * shorten them — measured, like the same idea in the bounce walks (profiles/r01_ab_tail_prefetch.txt) — and neither does
   * finishing the last walks of a wave with the whole wave, 64 consecutive entries tested per step (profiles/r02_ab_lockstep.txt):
   * the long rays of a tile are most of its rays, not a few stragglers. */
#if FLX_PRIMARY_PREFETCH
  if (nxt != WALK_END) {
    size_t i = (size_t)nxt * 3u;
    float4 e0 = sc.fwd[i], e1 = sc.fwd[i + 1], e2 = sc.fwd[i + 2];
    while (nxt != WALK_END) {
      const int meta = __float_as_int(e2.z);
      const bool box = (meta & 3) == 1;
      const uint32_t linkA = (uint32_t)__float_as_int(box ? e2.x : e2.y), linkB = (uint32_t)__float_as_int(e2.y);
      /* (a terminator's links are zeros and WALK_END has no entry: entry 0 stands in, its data is not used) */
      const size_t ia = (size_t)((meta & 3) == 0 || linkA == WALK_END ? 0u : linkA) * 3u, ib = (size_t)((meta & 3) == 0 || linkB == WALK_END ? 0u : linkB) * 3u;
      const float4 a0 = sc.fwd[ia], a1 = sc.fwd[ia + 1], a2 = sc.fwd[ia + 2];
      float4 b0 = a0, b1 = a1, b2 = a2;
      if (box) { b0 = sc.fwd[ib]; b1 = sc.fwd[ib + 1]; b2 = sc.fwd[ib + 2]; }
      visits++;
      primaryVisit(sc, ray, viewDepthPerS, w, cachedTI, hit, nxt, e0, e1, e2);
      const bool tookA = nxt == linkA;
      e0 = tookA ? a0 : b0; e1 = tookA ? a1 : b1; e2 = tookA ? a2 : b2;
    }
  }
#else
  while (nxt != WALK_END) {
    const size_t i = (size_t)nxt * 3u;
    const float4 e0 = sc.fwd[i], e1 = sc.fwd[i + 1], e2 = sc.fwd[i + 2];
    visits++;
    primaryVisit(sc, ray, viewDepthPerS, w, cachedTI, hit, nxt, e0, e1, e2, xfBase);
  }
#endif
  return hit;
}

template <bool COUNT>
FLX_DEV bool walkFetchT(const DeviceScene &sc, const float4 *lds, uint32_t ldsCount, WalkState &w, WalkEntry &cur, WorkCounters &cnt) {
  if ((uint32_t)w.i == WALK_END) return true;
  const uint32_t i = linkIndex((uint32_t)w.i);
  if (i < ldsCount) { cur.e0 = lds[3 * i]; cur.e1 = lds[3 * i + 1]; cur.e2 = lds[3 * i + 2]; }
  else { cur.e0 = sc.walk[3 * (size_t)i]; cur.e1 = sc.walk[3 * (size_t)i + 1]; cur.e2 = sc.walk[3 * (size_t)i + 2]; }
  if (COUNT) { if (w.mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; }
  const int meta = __float_as_int(cur.e2.z);
  int tI = (meta >> 2) << 1;
  if (tI != w.cachedTI) {
    int iI = tI + 1;
    M3 rotationII = rotation_at(sc, iI);
    w.cachedTI = tI;
    w.tR.origin = mul(rotationII, w.src.origin + shift_at(sc, iI));
    f3 d = mul(rotationII, w.src.dir);
    w.tR.dir = (w.mode == 0) ? normalize(d) : d;
    walkPrepareRay(sc, w);
  }
  return (meta & 3) == 0;
}
FLX_DEV bool walkIsBoxT(const WalkEntry &cur) { return (__float_as_int(cur.e2.z) & 3) == 1; }
/* ... the same from the link the walk followed to the entry (w.i names the entry `cur` holds until the test has chosen the next link) */
FLX_DEV bool walkIsBoxL(const WalkState &w) { return linkKind((uint32_t)w.i) == 1u; }
FLX_DEV void walkBoxT(WalkState &w, const WalkEntry &cur) {
  const bool hit = rayCuboidR(w.minLen, w, F3(cur.e0.x, cur.e0.y, cur.e0.z), F3(cur.e0.w, cur.e1.x, cur.e1.y));
  w.i = hit ? __float_as_int(cur.e2.x) : __float_as_int(cur.e2.y);
}
/* moellerTrumbore (fragment:123-140) and moellerTrumboreCull (fragment:143-158) as ONE straight-line
 * instruction stream: a wave holds shadow walks and closest-hit walks side by side, and the two
 * functions share every arithmetic operation (edges, pvec, det, 1/det, u, qvec, v, s) — they differ
 * only in the predicates.  edge1 = b - a and edge2 = c - a (fragment:124-125) come precomputed from the threaded
 * copy (build_threaded).  The shader's early returns have no side effects, so evaluating all of it
 * and combining the predicates at the end gives the same accept/reject and the same (s,u,v), NaNs
 * included: two-sided accepts unless (s > l || s <= BIAS), cull accepts only if (s <= l && s > BIAS). */
#ifndef FLX_MT_BRANCHFREE
#define FLX_MT_BRANCHFREE 1      /* dragon frame kernel 5.90 -> 5.86 ms, the other workloads unchanged (profiles/r03_ab_fused_trip.txt) */
#endif
FLX_DEV bool moellerTrumboreAny(f3 a, f3 edge1, f3 edge2, const Ray &ray, float l, bool cull, f3 &suv) {
  f3 pvec = cross(ray.dir, edge2);
  float det = dot(edge1, pvec);
  bool detBad = cull ? (det < BIAS) : (flx_abs(det) < BIAS);
  float inv_det = recipOfDet(det, detBad);               /* a rejected lane's u, v, s are never read */
  f3 tvec = ray.origin - a;
  float u = dot(tvec, pvec) * inv_det;
  f3 qvec = cross(tvec, edge1);
  float v = dot(ray.dir, qvec) * inv_det;
  float uvSum = u + v;
  float s = dot(edge2, qvec) * inv_det;
  bool uBad = (u < BIAS) || (u > 1.0f);
  bool vBad = (v < BIAS) || (uvSum > 1.0f);
#if FLX_MT_BRANCHFREE
  /* the cull rule is (s <= l) && (s > BIAS), the two-sided one !(s > l) && !(s <= BIAS) = ((s <= l) || unordered(s, l)) && ((s > BIAS) || s is NaN):
   * one pair of comparisons for both kinds of walk, the unordered cases added for the two-sided ones — no execution-mask branch on `cull` */
  const bool sLe = (s <= l) | (!cull & __builtin_isunordered(s, l));
  const bool sGt = (s > BIAS) | (!cull & (s != s));
  const bool sOk = sLe & sGt;
  suv = F3(s, u, v);
  return !detBad & !uBad & !vBad & sOk;
#else
  bool sOk = cull ? ((s <= l) && (s > BIAS)) : (!(s > l) && !(s <= BIAS));
  suv = F3(s, u, v);
  return !detBad && !uBad && !vBad && sOk;
#endif
}
#ifndef FLX_WF_LINK_XFORM
#define FLX_WF_LINK_XFORM 1
#endif
#ifndef FLX_WF_LINK_KIND
#define FLX_WF_LINK_KIND 1
#endif
#ifndef FLX_WF_LINK_ISBOX
#define FLX_WF_LINK_ISBOX 1
#endif
#ifndef FLX_WF_FLAT_FETCH
#define FLX_WF_FLAT_FETCH 1      /* entry fetch through one generic pointer (flat_load) instead of an LDS branch and a global branch: nine
                                  * instructions fewer per trip, +0.3 % frame after frame, +0.6 % in batches (profiles/r02_ab_walk_kernel.txt) */
#endif
FLX_DEV bool walkTriT(WalkState &w, const WalkEntry &cur) {
  /* the threaded copy stores a triangle as (a, b - a, c - a) */
  f3 a = F3(cur.e0.x, cur.e0.y, cur.e0.z), edge1 = F3(cur.e0.w, cur.e1.x, cur.e1.y), edge2 = F3(cur.e1.z, cur.e1.w, cur.e2.x);
  f3 suv;
  const bool cull = w.mode == 0;
  const bool hit = moellerTrumboreAny(a, edge1, edge2, w.tR, w.minLen, cull, suv);
  bool ended = false;
  if ((hit)) {
    if (cull) { w.shadowed = true; ended = true; }
    else if (suv.x != 0.0f) {                        /* fragment:217 */
      w.suv = suv; w.hitTI = (__float_as_int(cur.e2.z) >> 2) << 1; w.tri = __float_as_int(cur.e2.w);
      w.minLen = suv.x;
    }
  }
  w.i = __float_as_int(cur.e2.y);
  return ended;
}
/* walkTriT's second half on its own: what a triangle test's answer does to the walk (fragment:217-222 / :270-273) and the link it leaves by — for callers that run the
 * test on operands they picked themselves (the two-job walk waves, flx_wavefront.hip: k_wf_frame2) */
FLX_DEV bool walkTriApply(WalkState &w, const WalkEntry &cur, bool hit, f3 suv) {
  bool ended = false;
  if (hit) {
    if (w.mode == 0) { w.shadowed = true; ended = true; }
    else if (suv.x != 0.0f) {                        /* fragment:217 */
      w.suv = suv; w.hitTI = (__float_as_int(cur.e2.z) >> 2) << 1; w.tri = __float_as_int(cur.e2.w);
      w.minLen = suv.x;
    }
  }
  w.i = __float_as_int(cur.e2.y);
  return ended;
}
FLX_DEV f3 sel3(bool c, f3 a, f3 b) { return F3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
FLX_DEV void walkStartT(const DeviceScene &sc, WalkState &w, int mode, const Ray &ray, float len) {
  w.mode = mode; w.src = ray; w.tR = ray; w.cachedTI = 0; w.minLen = len; w.i = (int)sc.walk_root;
  walkPrepareRay(sc, w);
}

/* ---- threaded walk with the ray pre-transformed into every object space --------------------------
 * A ray meets a new transform ~3 times per walk (dragon scene); done per lane inside the stepping loop
 * that is ~100 VALU instructions (two mat3 products, a normalize) executed for ONE lane while 63 wait,
 * in almost every wave-iteration.  The walk kernel therefore transforms a ray into ALL object spaces
 * when the walk is set up (a batched point where many lanes set up together) and parks the results in LDS,
 * 40 bytes per transform; the stepping loop only reloads those.
 * The arithmetic per (ray, transform) is exactly fragment:197-202 / :257-262. */
FLX_DEV void walkSetupRays(const DeviceScene &sc, uint32_t nTransforms, const float4 *xf, float2 *rays, const Ray &src, bool shadowMode) {
  for (uint32_t t = 0; t < nTransforms; t++) {
    /* xf: the inverse rotation (3 columns) and inverse shift of every transform, staged in LDS by the kernel: the same
     * address for all lanes (a broadcast read) instead of four global loads per transform in the middle of the set-up */
    const float4 c0 = xf[4 * t], c1 = xf[4 * t + 1], c2 = xf[4 * t + 2], sh = xf[4 * t + 3];
    M3 rotationII; rotationII.c0 = F3(c0.x, c0.y, c0.z); rotationII.c1 = F3(c1.x, c1.y, c1.z); rotationII.c2 = F3(c2.x, c2.y, c2.z);
    f3 o = mul(rotationII, src.origin + F3(sh.x, sh.y, sh.z));
    f3 d = mul(rotationII, src.dir);
    if (flx_ballot(shadowMode) != 0ull) {                   /* skip the normalize when no lane of the wave sets up a shadow walk */
      f3 dn = normalize(d);
      if (shadowMode) d = dn;                             /* fragment:261 normalises, fragment:201 does not */
    }
    f3 inv; bool fast;
    reciprocalOfDir(sc, d, o, inv, fast);
    /* 40 bytes per transform: five float2 (origin, dir, 1/dir, fast flag) */
    rays[5 * t] = make_float2(o.x, o.y);
    rays[5 * t + 1] = make_float2(o.z, d.x);
    rays[5 * t + 2] = make_float2(d.y, d.z);
    rays[5 * t + 3] = make_float2(inv.x, inv.y);
    rays[5 * t + 4] = make_float2(inv.z, fast ? 1.0f : 0.0f);
  }
}
/* The tree top and the pre-transformed rays live in LDS; the walk kernel hands them around as plain pointers, and indexing
 * those costs 64-bit address arithmetic per fetch (v_mad_u64_u32: a quarter-rate instruction).  Saying that they are LDS
 * pointers makes it one 32-bit multiply-add. */
typedef float flx_v4f __attribute__((ext_vector_type(4)));
typedef float flx_v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const flx_v4f lds_cf4;
typedef __attribute__((address_space(3))) const flx_v2f lds_cf2;
FLX_DEV float4 ldsLoad4(const lds_cf4 *p) { const flx_v4f v = *p; return make_float4(v.x, v.y, v.z, v.w); }
/* the walk's ray in the object space of transform t, from the lane's pre-transformed set */
FLX_DEV void walkLoadRay(const float2 *raysGeneric, int t, WalkState &w) {
  const lds_cf2 *rays = (const lds_cf2 *)raysGeneric + __umul24((unsigned)t, 5u);      /* (a transform number: 24 bits are plenty) */
  const flx_v2f a = rays[0], b = rays[1], c = rays[2], d = rays[3], e = rays[4];
  w.tR.origin = F3(a.x, a.y, b.x);
  w.tR.dir = F3(b.y, c.x, c.y);
  w.inv = F3(d.x, d.y, e.x);
  w.fastDiv = e.y != 0.0f;
}
/* (walkG = sc.walk: the kernels whose arguments stay in the kernarg segment hand it over in registers — pinnedWalkCopy() — so that the stepping loop does
 * not read it again, with a wait that also covers the LDS reads in flight, every time a lane's entry lies outside the tree top) */
template <bool COUNT>
FLX_DEV bool walkFetchG(const float4 *walkG, const float4 *lds, uint32_t ldsCount, const float2 *rays, WalkState &w, WalkEntry &cur,
                        WorkCounters &cnt) {
#if FLX_WF_LINK_KIND
  /* What the entry a link names IS, the link says itself (build_threaded: kind 0 terminator, 1 box, 2 triangle, 3 = WALK_END, the loop bound of fragment:184): a walk ends
   * on kinds 0 and 3 without waiting for — or, the terminator's fetch being a counted visit and nothing else, issuing — a load; kinds 1 and 2 are fetched, and nothing the
   * trip decides next (another object space? box or triangle? still walking?) reads the loaded words, so the wait for them moves down to the test that uses them. */
  /* (No branch around the loads: a link of kind 0 or 3 has index 0 — the shared terminator, the first entry of the LDS top — and its lanes load that, a broadcast, once per
   * walk; a branch would be one more mask region in every trip.) */
  const uint32_t link0 = (uint32_t)w.i;
  const bool stop = ((link0 + (1u << LINK_KIND_SHIFT)) & (2u << LINK_KIND_SHIFT)) == 0u;      /* kind + 1 has bit 1 set for the kinds 1 and 2 only */
#else
  /* the loop bound of fragment:184: no fetch, so no visit is counted.  Uncounted kernels skip the test (four instructions a trip): WALK_END's index bits name the shared
   * terminator, whose fetch ends the walk all the same */
  if (COUNT) { if (((uint32_t)w.i == WALK_END)) return true; }
#endif
  const uint32_t i = linkIndex((uint32_t)w.i);
#if FLX_WF_FLAT_FETCH
  {   /* one instruction stream for both homes of an entry: a generic pointer into LDS or into the global copy (flat_load) */
    const float4 *src = (i < ldsCount) ? lds + 3u * i : walkG + 3 * (size_t)i;
    cur.e0 = src[0]; cur.e1 = src[1]; cur.e2 = src[2];
  }
#else
  if (i < ldsCount) { const lds_cf4 *L = (const lds_cf4 *)lds + __umul24(i, 3u); cur.e0 = ldsLoad4(L); cur.e1 = ldsLoad4(L + 1); cur.e2 = ldsLoad4(L + 2); }      /* (i < ldsCount <= 3 328) */
  else { cur.e0 = walkG[3 * (size_t)i]; cur.e1 = walkG[3 * (size_t)i + 1]; cur.e2 = walkG[3 * (size_t)i + 2]; }
#endif
#if FLX_WF_LINK_KIND
  if (COUNT) { if (linkKind(link0) != 3u) { if (w.mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; } }      /* (the loop bound is no visit; the terminator's fetch is one, fragment:208) */
#else
  if (COUNT) { if (w.mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; }
#endif
#if FLX_WF_LINK_XFORM
  /* the link says whether the entry it names stands in another object space than the entry it came from (LINK_XFORM, build_threaded) — which is the space the walk's ray
   * is in: no comparison with the space cached, and the verdict does not wait for the entry */
  if (FLX_UNLIKELY(((uint32_t)w.i & LINK_XFORM) != 0u)) {
    const int tI = (__float_as_int(cur.e2.z) >> 2) << 1;
    w.cachedTI = tI;
    walkLoadRay(rays, tI >> 1, w);
  }
#else
  const int tI = (__float_as_int(cur.e2.z) >> 2) << 1;
  if (FLX_UNLIKELY(tI != w.cachedTI)) {
    w.cachedTI = tI;
    walkLoadRay(rays, tI >> 1, w);
  }
#endif
#if FLX_WF_LINK_KIND
  return stop;
#else
  return (__float_as_int(cur.e2.z) & 3) == 0;
#endif
}
template <bool COUNT>
FLX_DEV bool walkFetchP(const DeviceScene &sc, const float4 *lds, uint32_t ldsCount, const float2 *rays, WalkState &w, WalkEntry &cur,
                        WorkCounters &cnt) {
  return walkFetchG<COUNT>(sc.walk, lds, ldsCount, rays, w, cur, cnt);
}
/* sc.walk held in scalar registers from here on: the compiler may move the registers about, not read the argument again */
FLX_DEV const float4 *pinnedWalkCopy(const DeviceScene &sc) {
  const float4 *p = sc.walk;
  asm volatile("" : "+s"(p));
  return p;
}
/* Entry fetch split from its use, so the loads of BOTH possible successors can be issued before the
 * current entry is tested and complete while the test's ~100 VALU instructions run. */
FLX_DEV void walkLoadEntry(const DeviceScene &sc, const float4 *lds, uint32_t ldsCount, uint32_t i, WalkEntry &e) {
  if (i == WALK_END) { e.e0 = e.e1 = e.e2 = make_float4(0.f, 0.f, 0.f, 0.f); return; }
  i = linkIndex(i);
  if (i < ldsCount) { const lds_cf4 *L = (const lds_cf4 *)lds + __umul24(i, 3u); e.e0 = ldsLoad4(L); e.e1 = ldsLoad4(L + 1); e.e2 = ldsLoad4(L + 2); }
  else { e.e0 = sc.walk[3 * (size_t)i]; e.e1 = sc.walk[3 * (size_t)i + 1]; e.e2 = sc.walk[3 * (size_t)i + 2]; }
}
/* The rest of walkFetchP once `cur` holds entry w.i: visit count, transform change, terminator test. */
template <bool COUNT>
FLX_DEV bool walkArriveP(const float2 *rays, WalkState &w, const WalkEntry &cur, WorkCounters &cnt) {
  if ((uint32_t)w.i == WALK_END) return true;
  if (COUNT) { if (w.mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; }
  const int meta = __float_as_int(cur.e2.z);
  const int tI = (meta >> 2) << 1;
  if (tI != w.cachedTI) {
    w.cachedTI = tI;
    walkLoadRay(rays, tI >> 1, w);
  }
  return (meta & 3) == 0;
}
/* rayCuboid (fragment:161-167) decided from ONE multiplication per quotient where that is provably enough.
 *
 * What the shader computes: the six quotients R = RN(a / d) (a = corner - origin), per axis near_k = min, far_k = max of its
 * two, tmin = max_k near_k, tmax = min_k far_k, and the boolean  tmax >= max(tmin, BIAS) && tmin < l.  Only the boolean
 * leaves the function.  With y = RN(1 / d) (w.inv; exact bits of 1.0f / d inside w.fastDiv's range, tools/micro/rcp_exact.hip)
 * the product q = RN(a * y) is within 3 ulp-halves of the quotient: q = (a/d)(1 + e1)(1 + e2), R = (a/d)(1 + e3), |e| <= 2^-24,
 * hence |q - R| < 2^-22 |q| while both are normal (|a| <= 2^60 and 2^-60 <= |d| <= 2^60 under w.fastDiv: no overflow; results
 * below 2^-126 are off by at most 2^-149 in absolute terms and far below BIAS = 2^-16 and below every l this function accepts).
 * lo(x) = x - 2^-21 |x| and hi(x) = x + 2^-21 |x| (one FMA each; its own rounding, 2^-24 |x|, fits in the slack between 2^-22 and
 * 2^-21) therefore bracket R, and being monotone they commute with min / max: lo(far_k') <= far_k, near_k <= hi(near_k'),
 * lo(min_k far') <= tmax <= hi(min_k far'), and the same for tmin.
 *   sure TRUE   every cross pair has room, lo(far_i') >= hi(max(near_j', near_k')) for the three i (the pairs of ONE axis hold by
 *               monotonicity of x -> RN(x / d): near_i <= far_i always — this is what keeps a flat box, whose two planes on an
 *               axis coincide, decidable), and lo(tmax') >= BIAS, and hi(tmin') < l;
 *   sure FALSE  hi(tmax') < max(lo(tmin'), BIAS), or lo(tmin') >= l.
 * Anything else — a ray grazing an edge within 2^-21, a direction outside the range, l below 2^-60 — is `unsure` and the caller
 * takes the exact quotients (rayCuboidRecip).  Same boolean as the shader in every case; ~40 instead of ~61 VALU instructions
 * per box test and a dependent chain of 7 instead of 10 (profiles/r02_ab_box_interval.txt). */
#ifndef FLX_WF_BOX_INTERVAL
#define FLX_WF_BOX_INTERVAL 1
#endif
FLX_DEV bool rayCuboidInterval(float l, const WalkState &w, f3 minCorner, f3 maxCorner, bool &sure) {
  const f3 o = w.tR.origin, y = w.inv;
  const float qx0 = (minCorner.x - o.x) * y.x, qx1 = (maxCorner.x - o.x) * y.x;
  const float qy0 = (minCorner.y - o.y) * y.y, qy1 = (maxCorner.y - o.y) * y.y;
  const float qz0 = (minCorner.z - o.z) * y.z, qz1 = (maxCorner.z - o.z) * y.z;
  const float nx = __builtin_fminf(qx0, qx1), fx = __builtin_fmaxf(qx0, qx1);
  const float ny = __builtin_fminf(qy0, qy1), fy = __builtin_fmaxf(qy0, qy1);
  const float nz = __builtin_fminf(qz0, qz1), fz = __builtin_fmaxf(qz0, qz1);
  constexpr float D = 4.76837158203125e-07f;           /* 2^-21 */
  const float nyz = __builtin_fmaxf(ny, nz), nxz = __builtin_fmaxf(nx, nz), nxy = __builtin_fmaxf(nx, ny);
  const float tmin = __builtin_fmaxf(nxy, nz), tmax = __builtin_fminf(__builtin_fminf(fx, fy), fz);
  /* (bitwise & and |, not && and ||: straight-line code, the wave would execute both sides of a branch anyway) */
  const bool cx = __builtin_fmaf(-flx_abs(fx), D, fx) >= __builtin_fmaf(flx_abs(nyz), D, nyz);
  const bool cy = __builtin_fmaf(-flx_abs(fy), D, fy) >= __builtin_fmaf(flx_abs(nxz), D, nxz);
  const bool cz = __builtin_fmaf(-flx_abs(fz), D, fz) >= __builtin_fmaf(flx_abs(nxy), D, nxy);
  const float tminHi = __builtin_fmaf(flx_abs(tmin), D, tmin), tminLo = __builtin_fmaf(-flx_abs(tmin), D, tmin);
  const float tmaxHi = __builtin_fmaf(flx_abs(tmax), D, tmax), tmaxLo = __builtin_fmaf(-flx_abs(tmax), D, tmax);
  const bool sureTrue = cx & cy & cz & (tmaxLo >= BIAS) & (tminHi < l);
  const bool sureFalse = (tmaxHi < __builtin_fmaxf(tminLo, BIAS)) | (tminLo >= l);
  sure = w.fastDiv & (l >= 8.673617379884035e-19f) & (sureTrue | sureFalse);      /* (NaN anywhere: every comparison false, unsure) */
  return sureTrue;
}
/* rayCuboid's boolean: from the interval test where it is sure, from the exact quotients where some lane of the wave is not */
FLX_DEV bool rayCuboidFast(float l, const WalkState &w, f3 lo, f3 hi) {
#if FLX_WF_BOX_INTERVAL
  bool sure;
  bool hit = rayCuboidInterval(l, w, lo, hi, sure);
#ifndef FLX_BOX_UNSURE_BALLOT
#define FLX_BOX_UNSURE_BALLOT 0
#endif
#if FLX_BOX_UNSURE_BALLOT
  if (FLX_UNLIKELY(flx_ballot(!sure) != 0ull)) {                         /* rare */
    if (!sure) hit = rayCuboidRecip(l, w, lo, hi);
  }
#else
  /* (a plain divergent branch: the ballot around it cost two vector instructions per box test — the compiler brings the mask into a VGPR to compare it with zero) */
  if (FLX_UNLIKELY(!sure)) hit = rayCuboidRecip(l, w, lo, hi);           /* rare */
#endif
  return hit;
#else
  return rayCuboidRecip(l, w, lo, hi);
#endif
}
FLX_DEV void walkBoxP(WalkState &w, const WalkEntry &cur) {
  const bool hit = rayCuboidFast(w.minLen, w, F3(cur.e0.x, cur.e0.y, cur.e0.z), F3(cur.e0.w, cur.e1.x, cur.e1.y));
  w.i = hit ? __float_as_int(cur.e2.x) : __float_as_int(cur.e2.y);
}

/* The two traversals of one bounce in ONE loop: shadowTest (fragment:231-280) on so.shadowRay, then
 * rayTracer (fragment:172-227) on the next ray.  The reference runs them back to back; they are
 * independent (the next direction does not depend on the shadow result), so each lane simply
 * starts its closest-hit walk the moment its own shadow walk ends instead of waiting for the
 * slowest shadow ray of the wave.  Per ray the entries visited, their order and every arithmetic
 * operation are unchanged. */
/* fragment:591 traces the next ray even when the loop guard (:475) is about to end the loop — in the last iteration, or once the
 * path's importancy has dropped below the threshold: a hit nobody shades ("incl. the useless last one").  Its
 * result reaches no output, so such a walk is not made (needClosest = false; the bounce then ends like a miss).  The guard: */
FLX_DEV bool nextBounceRuns(const DeviceFrame &fr, int i, f3 importancyFactor, f3 originalColor) {
  return (i + 1) < fr.max_reflections && length(importancyFactor * originalColor) >= fr.min_importancy * SQRT3;
}
/* ---- lockstep walk: small scenes in one object space (flx_api.hip: build_lockstep) --------------------------------
 * A scene of a few dozen entries (cornell, cornell.obj, the theater) is walked by the WAVE, not by the lane: the entries
 * stand in the reference's own order, where every successor lies further on, the wave steps through them once, and at
 * entry i the lanes whose walk stands at i test it.  The entry is the same for all of them: it comes through scalar loads
 * (no per-lane fetch, no address arithmetic, nothing for the next test to wait on — the loads of entry i + 1 do not depend
 * on the test of entry i), its numbers are scalar operands of the tests, and a trip runs EITHER the box test OR the
 * triangle test, never both under two masks.  Per ray the entries visited, their order, every arithmetic operation and the
 * visit counts are those of walkBounce()'s lane walk (and of rayTracer / shadowTest, fragment:172-280). */
typedef __attribute__((address_space(4))) const flx_v4f const_cf4;
template <bool COUNT, bool CULL>
FLX_DEV void walkLockPass(const DeviceScene &sc, bool active, const Ray &ray, float len, WalkState &w, uint32_t &visits) {
  w.tR = ray; w.minLen = len;
  reciprocalOfDir(sc, ray.dir, ray.origin, w.inv, w.fastDiv);
  uint32_t nxt = active ? sc.lock_root : WALK_END;
  const const_cf4 *L = (const const_cf4 *)sc.lock;
  const uint32_t n = sc.lock_entries;
  for (uint32_t i = 0; i < n; i++) {
    const bool mine = nxt == i;
    if (flx_ballot(mine) == 0ull) {
      if (flx_ballot(nxt != WALK_END) == 0ull) break;        /* every walk of the wave has ended */
      continue;
    }
    const const_cf4 *E = L + (size_t)__builtin_amdgcn_readfirstlane(i) * 3u;      /* (one address, three offsets) */
    const flx_v4f e0 = E[0], e1 = E[1], e2 = E[2];
    const int meta = __float_as_int(e2.z);
    if (mine) {
      if (COUNT) visits++;
      if ((meta & 3) == 1) {
        const bool hit = rayCuboidFast(w.minLen, w, F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y));
        nxt = (uint32_t)__float_as_int(hit ? e2.x : e2.y);
      } else if ((meta & 3) == 2) {
        f3 suv;
        const bool hit = moellerTrumboreAny(F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y), F3(e1.z, e1.w, e2.x), w.tR, w.minLen, CULL, suv);
        nxt = (uint32_t)__float_as_int(e2.y);
        if (hit) {
          if (CULL) { w.shadowed = true; nxt = WALK_END; }
          else if (suv.x != 0.0f) {                        /* fragment:217 */
            w.suv = suv; w.hitTI = (meta >> 2) << 1; w.tri = __float_as_int(e2.w);
            w.minLen = suv.x;
          }
        }
      } else {
        nxt = WALK_END;                                    /* terminator: its fetch counts (fragment:208), the walk ends */
      }
    }
  }
}

/* LOCK: the kernel variant for scenes that have the lockstep copy (sc.lock_entries != 0); the other variant does not carry its code */
template <bool COUNT, bool LOCK>
FLX_DEV void walkBounce(const DeviceScene &sc, bool needShadow, bool needClosest, const Ray &shadowRay, float shadowLen, const Ray &nextRay,
                        bool &shadowed, Hit &hit, WorkCounters &cnt) {
  WalkState w;
  walkClearResults(w);
  if (LOCK) {
    /* (no early return for a lane without walks: the passes are wave-wide, every lane that came here goes through them) */
    w.cachedTI = 0;
    if (COUNT) { if (needShadow) cnt.shadow_walks++; if (needClosest) cnt.closest_walks++; }
    if (flx_ballot(needShadow) != 0ull) walkLockPass<COUNT, true>(sc, needShadow, shadowRay, shadowLen, w, cnt.shadow_visits);
    if (flx_ballot(needClosest) != 0ull) walkLockPass<COUNT, false>(sc, needClosest, nextRay, POW32, w, cnt.closest_visits);
    shadowed = w.shadowed;
    hit.suv = w.suv; hit.transformId = w.hitTI; hit.triangleId = w.tri;
    return;
  }
  if (!needShadow && !needClosest) {
    shadowed = false;
    hit.suv = w.suv; hit.transformId = w.hitTI; hit.triangleId = w.tri;
    return;
  }
  if (sc.n_transforms > 1u) {
    /* scenes with several object spaces: a ray changes space a few times per walk and the reciprocal would have to be taken
     * again each time (three divisions) — the plain walk over the reference's array measures faster here (dragon: 26.2 vs 27.1 ms) */
    if (needShadow) walkStart(w, 0, shadowRay, shadowLen); else walkStart(w, 1, nextRay, POW32);
    if (COUNT) { if (needShadow) cnt.shadow_walks++; if (needClosest) cnt.closest_walks++; }
    while (w.mode != 2) {
      if (walkStep<COUNT>(sc, w, cnt)) {
        if (w.mode == 0 && needClosest) walkStart(w, 1, nextRay, POW32); else w.mode = 2;
      }
    }
    shadowed = w.shadowed;
    hit.suv = w.suv; hit.transformId = w.hitTI; hit.triangleId = w.tri;
    return;
  }
  /* one object space: over the threaded copy — explicit successors, stored edges, exact reciprocal box test, one test stream for
   * both kinds of walk (moellerTrumboreAny).  Entries, order, arithmetic and visit counts are those of walkStep(). */
  w.mode = needShadow ? 0 : 1;
  w.src = needShadow ? shadowRay : nextRay;
  w.minLen = needShadow ? shadowLen : POW32;
  if (COUNT) { if (needShadow) cnt.shadow_walks++; if (needClosest) cnt.closest_walks++; }
  for (;;) {
    w.tR = w.src; w.cachedTI = 0;
    reciprocalOfDir(sc, w.tR.dir, w.tR.origin, w.inv, w.fastDiv);
    uint32_t link = sc.walk_root;
    /* what the entry a link names is — box, triangle, the terminator, the loop bound — and whether it stands in another object space the link says itself (build_threaded):
     * the loop's control does not wait for the entry's loads (as in walkFetchG, FLX_WF_LINK_KIND) */
    for (;;) {
      if (((link + (1u << LINK_KIND_SHIFT)) & (2u << LINK_KIND_SHIFT)) == 0u) {      /* kind 0 or 3: the walk ends; the terminator's fetch is a counted visit and nothing else */
        if (COUNT) { if (linkKind(link) == 0u) { if (w.mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; } }
        break;
      }
      const size_t i = (size_t)linkIndex(link) * 3u;
      WalkEntry cur;
      cur.e0 = sc.walk[i]; cur.e1 = sc.walk[i + 1]; cur.e2 = sc.walk[i + 2];
      if (COUNT) { if (w.mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; }
      const bool isBox = linkKind(link) == 1u;
      if (FLX_UNLIKELY((link & LINK_XFORM) != 0u)) {
        const int tI = (__float_as_int(cur.e2.z) >> 2) << 1;
        const int iI = tI + 1;
        const M3 rotationII = rotation_at(sc, iI);
        w.cachedTI = tI;
        w.tR.origin = mul(rotationII, w.src.origin + shift_at(sc, iI));
        const f3 d = mul(rotationII, w.src.dir);
        w.tR.dir = (w.mode == 0) ? normalize(d) : d;        /* fragment:261 normalises, fragment:201 does not */
        reciprocalOfDir(sc, w.tR.dir, w.tR.origin, w.inv, w.fastDiv);
      }
      if (isBox) {
        walkBoxP(w, cur);
        link = (uint32_t)w.i;
      } else {
        const bool ended = walkTriT(w, cur);
        link = ended ? WALK_END : (uint32_t)w.i;
      }
    }
    if (w.mode == 0 && needClosest) { w.mode = 1; w.src = nextRay; w.minLen = POW32; continue; }
    break;
  }
  shadowed = w.shadowed;
  hit.suv = w.suv; hit.transformId = w.hitTI; hit.triangleId = w.tri;
}

/* fragment:445-460 + 580: fold the shadow result into the path. */
FLX_DEV void bounceFinish(PixelState &ps, PathState &p, const ShadeOut &so, bool shadowedByWalk) {
  bool shadowed = so.shadowedNoWalk || (so.needShadow && shadowedByWalk);
  if (shadowed && so.markId) ps.renderId.w += INV_255;
  f3 localColor = shadowed ? so.baseLuminance : so.litColor;
  p.finalColor = p.finalColor + localColor * p.importancyFactor;
}

/* One full bounce iteration (fragment:476-595); false when the path ends on a miss (fragment:593). */
/* bounce() with the surface part given: bounce 0 of a pixel's samples, which share the primary hit */
template <bool COUNT, bool LOCK>
FLX_DEV bool bounceOn(const DeviceScene &sc, const DeviceFrame &fr, const SurfaceCtx &sf, PixelState &ps, PathState &p, f3 camera, float cosSampleN,
                      int i, WorkCounters &cnt) {
  ShadeOut so;
  shadeSample(sc, fr, sf, ps, p, camera, cosSampleN, i, so);
  bool shadowed;
  walkBounce<COUNT, LOCK>(sc, so.needShadow, nextBounceRuns(fr, i, p.importancyFactor, ps.originalColor), so.shadowRay, so.shadowLen, p.ray, shadowed, p.hit, cnt);
  bounceFinish(ps, p, so, shadowed);
  if (p.hit.triangleId == -1) return false;
  p.lastHitPoint = p.ray.origin;
  return true;
}

template <bool COUNT, bool LOCK>
FLX_DEV bool bounce(const DeviceScene &sc, const DeviceFrame &fr, PixelState &ps, PathState &p, f3 camera, float cosSampleN, int i,
                    WorkCounters &cnt) {
  ShadeOut so;
  bounceShade<COUNT>(sc, fr, ps, p, camera, cosSampleN, i, so, cnt);
  bool shadowed;
  walkBounce<COUNT, LOCK>(sc, so.needShadow, nextBounceRuns(fr, i, p.importancyFactor, ps.originalColor), so.shadowRay, so.shadowLen, p.ray, shadowed, p.hit, cnt);
  bounceFinish(ps, p, so, shadowed);
  if (p.hit.triangleId == -1) return false;
  p.lastHitPoint = p.ray.origin;
  return true;
}

/* Primary ray of pixel (px, py_gl) of frame f of the batch: unit direction, NDC, view depth per unit s. */
FLX_DEV f3 primary_dir_v(const DeviceFrame &fr, const FrameView &v, uint32_t px, uint32_t py_gl, float &nx, float &ny, float &viewDepthPerS) {
  nx = ((float)px + 0.5f) / (float)fr.width * 2.0f - 1.0f;
  ny = ((float)py_gl + 0.5f) / (float)fr.height * 2.0f - 1.0f;
  const float *iv = v.inv_view;
  f3 d = F3((iv[0] * nx + iv[1] * ny) + iv[2], (iv[3] * nx + iv[4] * ny) + iv[5], (iv[6] * nx + iv[7] * ny) + iv[8]);
  d = normalize(d);
  viewDepthPerS = dot(F3(v.view_row2[0], v.view_row2[1], v.view_row2[2]), d);
  return d;
}
FLX_DEV f3 primary_dir(const DeviceFrame &fr, uint32_t f, uint32_t px, uint32_t py_gl, float &nx, float &ny, float &viewDepthPerS) {
  return primary_dir_v(fr, fr.view[f], px, py_gl, nx, ny, viewDepthPerS);
}

}  // namespace flx
#endif
