/*
 * flx_filter.hip — the denoise chain on gfx950: first / second / final filter kernels.
 *
 * What they compute: shaders/pathtracer_first_filter.glsl:18-123, pathtracer_second_filter.glsl:17-79,
 * pathtracer_final_filter.glsl:13-71, one thread per texel, over RGBA8 planes exactly like the
 * reference's render targets (modules/pathtracerWGL2.js:790-799): a store keeps
 * floor(clamp(x,0,1)*255 + 0.5), a fetch returns k/255, a fetch outside the image returns 0.
 * The pass schedule (which slot feeds which, including the two dropped outputs and the read of a
 * never-written plane at pass 4) is replayed by flx_api.hip from pathtracerWGL2.js:462-550.
 *
 * Memory: the planes are plain uint32 arrays in HBM, 8.3 MB each at 1080p, so the five inputs of a
 * pass sit in the XCD L2s / Infinity Cache; taps are 4-byte gathers, neighbouring threads hit the
 * same cache lines.  Threads are mapped 16x16 to keep a workgroup's taps together.
 */
#include <type_traits>
#include "flx_kernels.h"
#include "flx_kernel_util.h"

namespace flx {

struct Tex { const uint32_t *p; };       /* RGBA8 plane, rows top-down; null = never written (reads 0) */

__device__ __forceinline__ f4 F4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
__device__ __forceinline__ f4 add4(f4 a, f4 b) { return F4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ f4 scale4(f4 a, float s) { return F4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ bool eq3(f4 a, f4 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
__device__ __forceinline__ bool eq4(f4 a, f4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

/* The texel as stored.  Two texels are equal as vec4s exactly when their bytes are (k / 255 is injective), a channel is zero
 * exactly when its byte is, k / 255 > 0.1 (or >= 0.1) exactly when k >= 26, and int(k / 255 * 255.0) == k for every byte
 * (tools/unorm_check.c walks all 256): the filters decide on the bytes and turn into floats only what they accumulate. */
__device__ __forceinline__ uint32_t fetchRaw(Tex t, int W, int H, int x, int y_gl) {
  if (!t.p || x < 0 || y_gl < 0 || x >= W || y_gl >= H) return 0u;
  return t.p[(size_t)(H - 1 - y_gl) * W + x];
}
/* k / 255 correctly rounded without the division: RN(k * RN(1/255)) and one residual correction (exact for all 256 bytes) */
__device__ __forceinline__ float unorm8(uint32_t k) {
  const float r = 1.0f / 255.0f, kf = (float)k;
  const float q = kf * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, kf), r, q);
}
__device__ __forceinline__ f4 unpack(uint32_t q) { return F4(unorm8(q & 255u), unorm8((q >> 8) & 255u), unorm8((q >> 16) & 255u), unorm8(q >> 24)); }
__device__ __forceinline__ f4 fetch(Tex t, int W, int H, int x, int y_gl) { return unpack(fetchRaw(t, W, H, x, y_gl)); }      /* texelFetch, zero outside */
__device__ __forceinline__ bool rawEq3(uint32_t a, uint32_t b) { return ((a ^ b) & 0x00ffffffu) == 0u; }
__device__ __forceinline__ uint32_t rawW(uint32_t q) { return q >> 24; }

/* The second and the final filter reach at most 9 texels from the centre (stencil radius 3 x a scale of at most 3, SURVEY §8a
 * F2 / F3): a workgroup's 16 x 16 texels and their halo of the five planes are staged in LDS once (34 x 34 x 5 words = 23 KB)
 * and the ~37 x 5 data-dependent taps of every texel read from there.  Outside the image, and for a plane that is not bound,
 * the tile holds the 0 texelFetch yields there (fetchRaw). */
constexpr int FILTER_HALO = 9;
constexpr int FILTER_TW = 16 + 2 * FILTER_HALO;
struct TileOrigin { int x0, row0; };      /* image coordinates (x, row from the top) of the tile's first texel */
__device__ __forceinline__ TileOrigin tile_origin(int W) {
  const int tiles_x = (W + 15) >> 4;
  TileOrigin o; o.x0 = (int)(blockIdx.x % tiles_x) << 4; o.row0 = (int)(blockIdx.x / tiles_x) << 4;
  return o;
}
__device__ __forceinline__ void stage_tile(uint32_t *lds, Tex t, int W, int H, TileOrigin o) {
  for (int idx = threadIdx.x; idx < FILTER_TW * FILTER_TW; idx += 256) {
    const int lr = idx / FILTER_TW, lc = idx - lr * FILTER_TW;
    const int gx = o.x0 - FILTER_HALO + lc, grow = o.row0 - FILTER_HALO + lr;
    lds[idx] = (t.p && gx >= 0 && grow >= 0 && gx < W && grow < H) ? t.p[(size_t)grow * W + gx] : 0u;
  }
}
/* fetchRaw(t, W, H, x, y_gl) from the staged tile; (x, y_gl) within FILTER_HALO of the workgroup's texels */
__device__ __forceinline__ uint32_t fetchTile(const uint32_t *lds, int H, TileOrigin o, int x, int y_gl) {
  return lds[((H - 1 - y_gl) - o.row0 + FILTER_HALO) * FILTER_TW + (x - o.x0 + FILTER_HALO)];
}

__device__ __forceinline__ uint32_t quant(float x) { return quant_unorm8(x); }                  /* flx_kernel_util.h */
__device__ __forceinline__ uint32_t pack(f4 v) { return pack_rgba8(v.x, v.y, v.z, v.w); }

__device__ const float STENCIL3_37[37][2] = {
                              {-3, -1}, {-3, 0}, {-3, 1},
                    {-2, -2}, {-2, -1}, {-2, 0}, {-2, 1}, {-2, 2},
  {-1, -3}, {-1, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {-1, 2}, {-1, 3},
  { 0, -3}, { 0, -2}, { 0, -1}, { 0, 0}, { 0, 1}, { 0, 2}, { 0, 3},
  { 1, -3}, { 1, -2}, { 1, -1}, { 1, 0}, { 1, 1}, { 1, 2}, { 1, 3},
                    { 2, -2}, { 2, -1}, { 2, 0}, { 2, 1}, { 2, 2},
                              { 3, -1}, { 3, 0}, { 3, 1}
};
__device__ const float STENCIL3_36[36][2] = {
                              {-3, -1}, {-3, 0}, {-3, 1},
                    {-2, -2}, {-2, -1}, {-2, 0}, {-2, 1}, {-2, 2},
  {-1, -3}, {-1, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {-1, 2}, {-1, 3},
  { 0, -3}, { 0, -2}, { 0, -1},          { 0, 1}, { 0, 2}, { 0, 3},
  { 1, -3}, { 1, -2}, { 1, -1}, { 1, 0}, { 1, 1}, { 1, 2}, { 1, 3},
                    { 2, -2}, { 2, -1}, { 2, 0}, { 2, 1}, { 2, 2},
                              { 3, -1}, { 3, 0}, { 3, 1}
};
__device__ const int STENCIL1[4][2] = { {-1, 0}, {0, -1}, {0, 1}, {1, 0} };

__device__ __forceinline__ bool texel_of_thread(int W, int H, int &x, int &y_gl) {
  const int tiles_x = (W + 15) >> 4;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  x = (tx << 4) + (threadIdx.x & 15);
  const int row = (ty << 4) + (threadIdx.x >> 4);
  y_gl = H - 1 - row;
  return x < W && row < H;
}

/* The first filter's workgroup: its 37 taps are gathers of one texel per lane, up to 42 texels away — what a gather costs is the number of 128-byte lines the wave's 64 texels lie
 * in, so its waves are 32 x 2 texels (two lines per gather) instead of the 16 x 4 of the other kernels (four half lines).  FLX_FILTER_FIRST_SHAPE: 0 = 16 x 16 workgroup of
 * 16 x 4 waves, 1 = 64 x 4 workgroup of 64 x 1 waves, 2 = 32 x 8 workgroup of 32 x 2 waves: the cornell.obj filter frame 0.930 (0) 0.915 (1) 0.910 ms (2), 5 us per pass
 * (profiles/r04_paths_occupancy.txt; taps in flight: 8 stays — 4: 0.940, 13: 0.951). */
#ifndef FLX_FILTER_FIRST_SHAPE
#define FLX_FILTER_FIRST_SHAPE 2
#endif
constexpr int FF_W = FLX_FILTER_FIRST_SHAPE == 1 ? 64 : FLX_FILTER_FIRST_SHAPE == 2 ? 32 : 16, FF_H = 256 / FF_W;
/* The first filter's taps lie (int)(stencil x k^2 x 3.5) texels from the centre, stencil in -3 .. 3, k = 1 + originalColor.w in 1 .. 2: up to 42 texels — but on a frame like
 * BASELINE configs[1] the median k of the pixels that have taps is 1.10 (12 texels) and a quarter of them sit at k = 2.  FLX_FILTER_FIRST_LDS: a workgroup that has pixels whose
 * taps stay within FF_HALO texels stages its four planes with that halo in LDS once (coalesced rows) and those pixels take their 37 x 4 gathers from there; the others gather
 * from memory as before.  Same values, same order of the sums. */
/* MEASURED AND OFF (profiles/r04_paths_occupancy.txt): the cornell.obj filter frame 0.914 -> 0.99 - 1.00 ms whatever the share of workgroups that stage (FLX_FILTER_FIRST_LDS_MIN 32 .. 224) —
 * the 31.5 KB of LDS per workgroup cost the waves that hide the far gathers' latency; the near gathers were mostly cache hits already. */
#ifndef FLX_FILTER_FIRST_LDS
#define FLX_FILTER_FIRST_LDS 0
#endif
#ifndef FLX_FILTER_FIRST_LDS_MIN
#define FLX_FILTER_FIRST_LDS_MIN 96
#endif
constexpr int FF_HALO = 13, FF_TW = FF_W + 2 * FF_HALO, FF_TH = FF_H + 2 * FF_HALO;
__device__ __forceinline__ bool texel_of_thread_first(int W, int H, int &x, int &y_gl) {
  const int tiles_x = (W + FF_W - 1) / FF_W;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  x = tx * FF_W + (int)(threadIdx.x % FF_W);
  const int row = ty * FF_H + (int)(threadIdx.x / FF_W);
  y_gl = H - 1 - row;
  return x < W && row < H;
}

/* a float4 plane of the path-trace pass -> the RGBA8 target it renders into */
__global__ __launch_bounds__(256) void k_quantize(const float4 *__restrict__ src, uint32_t *__restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 v = src[i];
  dst[i] = pack(F4(v.x, v.y, v.z, v.w));
}

/* Temporal accumulation, the shader modules/pathtracerWGL2.js:571-662 generates: average the history slots whose
 * location id equals the newest frame's; slots are visited in groups of four with vec4(0) standing in for slots >= n
 * (an all-zero id — an uncovered pixel — "matches" those stand-ins too, as in the shader). */
__global__ __launch_bounds__(256) void k_temporal(TemporalRings r, int W, int H, int hdr, int use_filter, uint32_t *dColor, uint32_t *dIp, float4 *out) {
  int x, y;
  if (!texel_of_thread(W, H, x, y)) return;
  Tex c0 = { r.c[0] }, ip0 = { r.ip[0] }, id0 = { r.id[0] }, oid0 = { r.oid[0] };
  const f4 id = fetch(id0, W, H, x, y), originalId = fetch(oid0, W, H, x, y);
  float counter = 1.0f, glassCounter = 1.0f;
  const f4 cc = fetch(c0, W, H, x, y), ci = fetch(ip0, W, H, x, y);
  const float centerW = cc.w;
  float color[3] = { cc.x + ci.x * 256.0f, cc.y + ci.y * 256.0f, cc.z + ci.z * 256.0f };
  float glassFilter = ci.w;
  for (int i = 1; i < r.n; i += 4) {
    f4 cs[4], ips[4], ids[4], oids[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = i + j;
      if (k < r.n) {
        Tex tc = { r.c[k] }, tip = { r.ip[k] }, tid = { r.id[k] }, toid = { r.oid[k] };
        cs[j] = fetch(tc, W, H, x, y); ips[j] = fetch(tip, W, H, x, y); ids[j] = fetch(tid, W, H, x, y); oids[j] = fetch(toid, W, H, x, y);
      } else {
        cs[j] = ips[j] = ids[j] = oids[j] = F4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) if (eq4(ids[j], id)) {
      color[0] += cs[j].x + ips[j].x * 256.0f; color[1] += cs[j].y + ips[j].y * 256.0f; color[2] += cs[j].z + ips[j].z * 256.0f;
      counter += 1.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) if (eq4(oids[j], originalId)) {
      glassFilter += ips[j].w;
      glassCounter += 1.0f;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) color[k] /= counter;
  glassFilter /= glassCounter;
  const size_t o = (size_t)(H - 1 - y) * W + x;
  if (use_filter) {
    dColor[o] = pack(F4(flx_mod(color[0], 1.0f), flx_mod(color[1], 1.0f), flx_mod(color[2], 1.0f), centerW));
    dIp[o] = pack(F4(flx_floor(color[0]) / 256.0f, flx_floor(color[1]) / 256.0f, flx_floor(color[2]) / 256.0f, glassFilter));
  } else {
    if (hdr == 1) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        color[k] = color[k] / (color[k] + 1.0f);
        const float gamma = 0.8f;
        color[k] = flx_pow(4.0f * color[k], 1.0f / gamma) / 4.0f * 1.3f;
      }
    }
    out[o] = make_float4(color[0], color[1], color[2], centerW);
  }
}

#ifndef FLX_FILTER_FIRST_CH
#define FLX_FILTER_FIRST_CH 8
#endif
/* the 37 taps of one texel (k_filter_first), TILE: from the workgroup's staged planes */
struct FirstTapsIn {
  const uint32_t *pId, *pOId, *pIp, *pColor;
  uint32_t mId, mOId, mIp;
  size_t centre;
  int x, y, W, H, x0, r0;
  float k;
  uint32_t rCenterId, rCenterOId;
  int centerLightNum, centerShadow;
};
template <bool TILE, typename TILES>
__device__ __forceinline__ void first_filter_taps(const FirstTapsIn &t, const TILES &ftile, f4 &color, float &count) {
  const int x = t.x, y = t.y, W = t.W, H = t.H, x0 = t.x0, r0 = t.r0;
  const float k = t.k;
  const size_t centre = t.centre;
  const uint32_t *pId = t.pId, *pOId = t.pOId, *pIp = t.pIp;
  const uint32_t mId = t.mId, mOId = t.mOId, mIp = t.mIp, rCenterId = t.rCenterId, rCenterOId = t.rCenterOId;
  const int centerLightNum = t.centerLightNum, centerShadow = t.centerShadow;
  (void)x0; (void)r0; (void)ftile;
  constexpr int CH = FLX_FILTER_FIRST_CH;
  for (int base = 0; base < 37; base += CH) {
    size_t at[CH]; bool in[CH], pass[CH];
    uint32_t id[CH], oid[CH], rc[CH], rip[CH];
#pragma unroll
    for (int j = 0; j < CH; j++) {
      const int i = base + j < 37 ? base + j : 36;
      const int cx = x + (int)(STENCIL3_37[i][0] * k * k * 3.5f);
      const int cy = y + (int)(STENCIL3_37[i][1] * k * k * 3.5f);
      in[j] = base + j < 37 && cx >= 0 && cy >= 0 && cx < W && cy < H;
#if FLX_FILTER_FIRST_LDS
      if constexpr (TILE) {                               /* (a tap outside the image: the centre's place in the tile, its value is not used) */
        at[j] = in[j] ? (size_t)((H - 1 - cy) - r0) * FF_TW + (size_t)(cx - x0) : (size_t)((H - 1 - y) - r0) * FF_TW + (size_t)(x - x0);
        id[j] = ftile[0][at[j]]; oid[j] = ftile[1][at[j]];
      } else
#endif
      {
        at[j] = in[j] ? (size_t)(H - 1 - cy) * W + cx : centre;
        id[j] = pId[at[j]]; oid[j] = pOId[at[j]];
      }
    }
#pragma unroll
    for (int j = 0; j < CH; j++) {
      const uint32_t idj = in[j] ? id[j] & mId : 0u, oidj = in[j] ? oid[j] & mOId : 0u;
      const int idW = (int)rawW(idj);
      pass[j] = base + j < 37 && rawEq3(rCenterId, idj) && rCenterOId == oidj && (centerLightNum != idW / 2 || centerShadow == idW % 2);
#if FLX_FILTER_FIRST_LDS
      if constexpr (TILE) { rc[j] = ftile[2][at[j]]; rip[j] = ftile[3][at[j]]; } else
#endif
      {
        const size_t a = pass[j] ? at[j] : centre;
        rc[j] = t.pColor[a]; rip[j] = pIp[a];
      }
    }
#pragma unroll
    for (int j = 0; j < CH; j++) if (pass[j]) {
      const f4 nextColor = unpack(in[j] ? rc[j] : 0u);
      const f4 nextColorIp = unpack(in[j] ? rip[j] & mIp : 0u);
      color = add4(color, add4(nextColor, scale4(nextColorIp, 256.0f)));
      count += 1.0f;
    }
  }
}

/* pathtracer_first_filter.glsl:18-123 */
__global__ __launch_bounds__(256) void k_filter_first(Tex tColor, Tex tIp, Tex tOColor, Tex tId, Tex tOId, uint32_t *dColor, uint32_t *dIp,
                                                      uint32_t *dId, int W, int H) {
  int x, y;
  const bool inside = texel_of_thread_first(W, H, x, y);
  if (!FLX_FILTER_FIRST_LDS && !inside) return;              /* (with the tile every thread of the workgroup reaches the barriers; fetchRaw() gives 0 outside the image) */
#if FLX_FILTER_FIRST_LDS
  __shared__ uint32_t ftile[4][FF_TW * FF_TH];               /* id, original id, colour, colour's integer part: the workgroup's texels and FF_HALO around them */
#endif
  const f4 centerColor = unpack(fetchRaw(tColor, W, H, x, y));
  const uint32_t rCenterIp = fetchRaw(tIp, W, H, x, y), rCenterOColor = fetchRaw(tOColor, W, H, x, y);
  const uint32_t rCenterId = fetchRaw(tId, W, H, x, y), rCenterOId = fetchRaw(tOId, W, H, x, y);
  const int centerIdw = (int)rawW(rCenterId);               /* int(centerId.w * 255.0) */
  const int centerLightNum = centerIdw / 2;
  const int centerShadow = centerIdw % 2;
  uint32_t rRenderId = rCenterId;
  f4 renderColorIp = F4(0.0f, 0.0f, 0.0f, 0.0f);
  f4 color = F4(0.0f, 0.0f, 0.0f, 0.0f);
  float count = 0.0f;
  if (rawW(rCenterOId) != 0u && rawW(rCenterIp) != 0u) {
    uint32_t ids[4], oIds[4], ipws[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      ids[i] = fetchRaw(tId, W, H, x + STENCIL1[i][0], y + STENCIL1[i][1]);
      oIds[i] = fetchRaw(tOId, W, H, x + STENCIL1[i][0], y + STENCIL1[i][1]);
      ipws[i] = rawW(fetchRaw(tIp, W, H, x + STENCIL1[i][0], y + STENCIL1[i][1]));
    }
    int vote[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (ipws[i] == 0u) {
        vote[i] = 1;
        if (rawEq3(ids[i], rCenterId) && oIds[i] == rCenterOId) vote[i]++;
#pragma unroll
        for (int j = i + 1; j < 4; j++) if (rawEq3(ids[i], ids[j]) && oIds[i] == oIds[j]) vote[i]++;
      }
    }
    int maxVote = vote[0];
    int idNumber = 0;
#pragma unroll
    for (int i = 1; i < 4; i++) if (vote[i] >= maxVote) { maxVote = vote[i]; idNumber = i; }
    rRenderId = idNumber == 0 ? ids[0] : idNumber == 1 ? ids[1] : idNumber == 2 ? ids[2] : ids[3];
    renderColorIp.w = flx_max(1.0f - flx_sign((float)maxVote), 0.0f);
  }
  const bool wantTaps = inside && rawW(rCenterOColor) != 0u;
  const float k = 1.0f + unorm8(rawW(rCenterOColor));
#if FLX_FILTER_FIRST_LDS
  /* (the farthest tap: |stencil| = 3; the same float expression as the taps', monotone in |stencil|) */
  bool small = wantTaps && (int)(3.0f * k * k * 3.5f) <= FF_HALO;
  const int tilesX = (W + FF_W - 1) / FF_W;
  const int x0 = (int)(blockIdx.x % tilesX) * FF_W - FF_HALO, r0 = (int)(blockIdx.x / tilesX) * FF_H - FF_HALO;      /* the tile's first column and first memory row */
  /* (staging costs ~31 loads per thread: it pays from FLX_FILTER_FIRST_LDS_MIN texels with near taps on) */
  const bool staged = __syncthreads_count(small) >= FLX_FILTER_FIRST_LDS_MIN;
  small = small && staged;
  if (staged) {
    for (int idx = threadIdx.x; idx < FF_TW * FF_TH; idx += 256) {
      const int lr = idx / FF_TW, lc = idx - lr * FF_TW;
      const int gx = x0 + lc, gr = r0 + lr;
      const bool in = gx >= 0 && gr >= 0 && gx < W && gr < H;
      const size_t off = in ? (size_t)gr * W + gx : 0;
      ftile[0][idx] = in && tId.p ? tId.p[off] : 0u;
      ftile[1][idx] = in && tOId.p ? tOId.p[off] : 0u;
      ftile[2][idx] = in ? tColor.p[off] : 0u;
      ftile[3][idx] = in && tIp.p ? tIp.p[off] : 0u;
    }
    __syncthreads();
  }
#else
  const bool small = false;
#endif
  if (!inside) return;
  if (!wantTaps) {
    color = centerColor;
    count = 1.0f;
  } else {
    /* The 37 taps reach up to 42 texels from the centre and each is two gathers, a decision, two more
     * gathers.  Taken one tap after the other that is 37 round trips per texel; the taps do not depend on each other, so they
     * go eight at a time: the sixteen id gathers in flight together, then the colours of the taps that passed (a tap that did
     * not re-reads the centre texel: its value is not used), then the sums in tap order as the shader adds them.
     * fetchRaw()'s "0 outside the image or from an unbound plane" as arithmetic, so that every load is unconditional: */
    const size_t centre = (size_t)(H - 1 - y) * W + x;
    const uint32_t *pId = tId.p ? tId.p : tColor.p, *pOId = tOId.p ? tOId.p : tColor.p, *pIp = tIp.p ? tIp.p : tColor.p;      /* (tColor is always bound) */
    const uint32_t mId = tId.p ? ~0u : 0u, mOId = tOId.p ? ~0u : 0u, mIp = tIp.p ? ~0u : 0u;
    FirstTapsIn t = { pId, pOId, pIp, tColor.p, mId, mOId, mIp, centre, x, y, W, H, 0, 0, k, rCenterId, rCenterOId, centerLightNum, centerShadow };
#if FLX_FILTER_FIRST_LDS
    t.x0 = x0; t.r0 = r0;
    if (small) first_filter_taps<true>(t, ftile, color, count); else first_filter_taps<false>(t, ftile, color, count);
#else
    first_filter_taps<false>(t, 0, color, count);
#endif
  }
  const float invCount = 1.0f / count;
  const float sg = flx_sign(centerColor.w);
  const float cx_ = color.x * invCount, cy_ = color.y * invCount, cz_ = color.z * invCount;
  const size_t o = (size_t)(H - 1 - y) * W + x;
  dColor[o] = pack(F4(sg * flx_mod(cx_, 1.0f), sg * flx_mod(cy_, 1.0f), sg * flx_mod(cz_, 1.0f), sg * centerColor.w));
  dIp[o] = pack(F4(sg * (flx_floor(cx_) * INV_256), sg * (flx_floor(cy_) * INV_256), sg * (flx_floor(cz_) * INV_256), sg * renderColorIp.w));
  if (dId) dId[o] = rRenderId;                               /* a texel copied as it is stored */
}

__global__ __launch_bounds__(256) void k_filter_second(Tex tColor, Tex tIp, Tex tOColor, Tex tId, Tex tOId, uint32_t *dColor, uint32_t *dIp,
                                                       uint32_t *dOrig, int W, int H) {
  __shared__ uint32_t tile[5][FILTER_TW * FILTER_TW];
  const TileOrigin org = tile_origin(W);
  int x, y;
  const bool inImage = texel_of_thread(W, H, x, y);
  /* Every output of this pass is centerColor.w times something, and quant() stores 0 for +-0 and for NaN alike: a texel whose
   * colour has w = 0 — one the path-trace pass did not cover — writes zeros whatever its 36 taps hold.  A workgroup of such
   * texels (most of a frame around a small model) skips the staging and the taps, a lone one the taps. */
  const size_t o = inImage ? (size_t)(H - 1 - y) * W + x : 0;
  const bool covered = inImage && rawW(fetchRaw(tColor, W, H, x, y)) != 0u;
  if (!__syncthreads_or(covered)) {
    if (inImage) { dColor[o] = 0u; dIp[o] = 0u; if (dOrig) dOrig[o] = 0u; }
    return;
  }
  stage_tile(tile[0], tColor, W, H, org); stage_tile(tile[1], tIp, W, H, org); stage_tile(tile[2], tOColor, W, H, org);
  stage_tile(tile[3], tId, W, H, org); stage_tile(tile[4], tOId, W, H, org);
  __syncthreads();
  if (!inImage) return;
  if (!covered) { dColor[o] = 0u; dIp[o] = 0u; if (dOrig) dOrig[o] = 0u; return; }
  const f4 centerColor = unpack(fetchTile(tile[0], H, org, x, y));
  const f4 centerColorIp = unpack(fetchTile(tile[1], H, org, x, y));
  const f4 centerOColor = unpack(fetchTile(tile[2], H, org, x, y));
  const uint32_t rCenterId = fetchTile(tile[3], H, org, x, y), rCenterOId = fetchTile(tile[4], H, org, x, y);
  const uint32_t centerIpW = rawW(fetchTile(tile[1], H, org, x, y));
  f4 color = add4(centerColor, scale4(F4(centerColorIp.x, centerColorIp.y, centerColorIp.z, 0.0f), 256.0f));
  f4 oColor = centerOColor;
  float ipw = centerColorIp.w;
  float count = 1.0f, oCount = 1.0f;
  const float scale = 1.0f + 2.0f * flx_tanh(centerOColor.w + unorm8(rawW(rCenterOId)) * 4.0f);
  for (int i = 0; i < 36; i++) {
    const int cx = x + (int)(STENCIL3_36[i][0] * scale);
    const int cy = y + (int)(STENCIL3_36[i][1] * scale);
    const uint32_t nextOId = fetchTile(tile[4], H, org, cx, cy);
    if (!rawEq3(rCenterOId, nextOId)) continue;
    const uint32_t id = fetchTile(tile[3], H, org, cx, cy);
    const uint32_t rNextIp = fetchTile(tile[1], H, org, cx, cy);
    const uint32_t minOIdW = rawW(rCenterOId) < rawW(nextOId) ? rawW(rCenterOId) : rawW(nextOId);
    const uint32_t maxIpW = rawW(rNextIp) > centerIpW ? rawW(rNextIp) : centerIpW;
    if (minOIdW >= 26u && (id == rCenterId || maxIpW >= 26u)) {                 /* min(...) > 0.1 && (ids equal || max(...) >= 0.1) */
      const f4 nextColor = unpack(fetchTile(tile[0], H, org, cx, cy)), nextColorIp = unpack(rNextIp), nextOColor = unpack(fetchTile(tile[2], H, org, cx, cy));
      color = add4(color, add4(nextColor, scale4(F4(nextColorIp.x, nextColorIp.y, nextColorIp.z, 0.0f), 256.0f)));
      count += 1.0f;
      ipw += nextColorIp.w;
      oColor = add4(oColor, nextOColor);
      oCount += 1.0f;
    } else if (rawEq3(id, rCenterId)) {
      const f4 nextColor = unpack(fetchTile(tile[0], H, org, cx, cy)), nextColorIp = unpack(rNextIp);
      color = add4(color, add4(nextColor, scale4(F4(nextColorIp.x, nextColorIp.y, nextColorIp.z, 0.0f), 256.0f)));
      count += 1.0f;
    }
  }
  const float invCount = 1.0f / count;
  const float w = centerColor.w;
  const float cx_ = color.x * invCount, cy_ = color.y * invCount, cz_ = color.z * invCount;
  dColor[o] = pack(F4(w * flx_mod(cx_, 1.0f), w * flx_mod(cy_, 1.0f), w * flx_mod(cz_, 1.0f), w * (color.w * invCount)));
  dIp[o] = pack(F4(w * (flx_floor(cx_) * INV_256), w * (flx_floor(cy_) * INV_256), w * (flx_floor(cz_) * INV_256), w * ipw));
  if (dOrig) dOrig[o] = pack(F4((w * oColor.x) / oCount, (w * oColor.y) / oCount, (w * oColor.z) / oCount, (w * oColor.w) / oCount));
}

__global__ __launch_bounds__(256) void k_filter_final(Tex tColor, Tex tIp, Tex tOColor, Tex tId, Tex tOId, float4 *out, int W, int H, int hdr) {
  __shared__ uint32_t tile[5][FILTER_TW * FILTER_TW];
  const TileOrigin org = tile_origin(W);
  int x, y;
  const bool inImage = texel_of_thread(W, H, x, y);
  /* a texel whose colour has w = 0 yields vec4(0) whatever its taps hold (the last branch of the shader): workgroups of such
   * texels skip the staging and the taps, lone ones the taps */
  const bool covered = inImage && rawW(fetchRaw(tColor, W, H, x, y)) != 0u;
  if (!__syncthreads_or(covered)) {
    if (inImage) out[(size_t)(H - 1 - y) * W + x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return;
  }
  stage_tile(tile[0], tColor, W, H, org); stage_tile(tile[1], tIp, W, H, org); stage_tile(tile[2], tOColor, W, H, org);
  stage_tile(tile[3], tId, W, H, org); stage_tile(tile[4], tOId, W, H, org);
  __syncthreads();
  if (!inImage) return;
  if (!covered) { out[(size_t)(H - 1 - y) * W + x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); return; }
  const uint32_t rCenterColor = fetchTile(tile[0], H, org, x, y);
  const uint32_t centerIpW = rawW(fetchTile(tile[1], H, org, x, y));
  const f4 centerOColor = unpack(fetchTile(tile[2], H, org, x, y));
  const uint32_t rCenterId = fetchTile(tile[3], H, org, x, y), rCenterOId = fetchTile(tile[4], H, org, x, y);
  f4 color = F4(0.0f, 0.0f, 0.0f, 0.0f), oColor = color;
  float count = 0.0f, oCount = 0.0f;
  const float scale = 0.7f + 2.0f * flx_tanh(centerOColor.w + unorm8(rawW(rCenterOId)) * 4.0f);
  for (int i = 0; i < 37; i++) {
    const int cx = x + (int)(STENCIL3_37[i][0] * scale);
    const int cy = y + (int)(STENCIL3_37[i][1] * scale);
    const uint32_t nextOId = fetchTile(tile[4], H, org, cx, cy);
    if (!rawEq3(rCenterOId, nextOId)) continue;               /* both accumulations need the original ids to agree */
    const uint32_t rNextIp = fetchTile(tile[1], H, org, cx, cy);
    const uint32_t maxIpW = rawW(rNextIp) > centerIpW ? rawW(rNextIp) : centerIpW;
    const uint32_t minOIdW = rawW(rCenterOId) < rawW(nextOId) ? rawW(rCenterOId) : rawW(nextOId);
    const bool blurTranslucent = maxIpW != 0u && minOIdW > 0u;
    if (blurTranslucent) {
      oColor = add4(oColor, unpack(fetchTile(tile[2], H, org, cx, cy)));
      oCount += 1.0f;
    }
    if (blurTranslucent || rawEq3(rCenterId, fetchTile(tile[3], H, org, cx, cy))) {
      color = add4(color, add4(unpack(fetchTile(tile[0], H, org, cx, cy)), scale4(unpack(rNextIp), 255.0f)));
      count += 1.0f;
    }
  }
  float4 res = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (rawW(rCenterColor) > 0u) {
    float f[3] = { color.x / count, color.y / count, color.z / count };
    float m[3];
    if (oCount == 0.0f) { m[0] = centerOColor.x; m[1] = centerOColor.y; m[2] = centerOColor.z; }
    else { m[0] = oColor.x / oCount; m[1] = oColor.y / oCount; m[2] = oColor.z / oCount; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      f[c] = f[c] * m[c];
      if (hdr == 1) {
        f[c] = f[c] / (f[c] + 1.0f);
        const float gamma = 0.8f;
        f[c] = flx_pow(4.0f * f[c], 1.0f / gamma) / 4.0f * 1.3f;
      }
    }
    res = make_float4(f[0], f[1], f[2], 1.0f);
  }
  out[(size_t)(H - 1 - y) * W + x] = res;
}

/* ---- anti-aliasing post passes: modules/fxaa.js:7-137, modules/taa.js:11-59 — one thread per texel over the RGBA8
 * texture the renderer drew into; the float the shader outputs is kept ---- */
__device__ __forceinline__ float fxaa_luma(f4 c) { return (c.y * (0.587f / 0.299f) + c.x) * c.w; }
__device__ __forceinline__ f4 mix4(f4 a, f4 b, float t) { return F4(flx_mix(a.x, b.x, t), flx_mix(a.y, b.y, t), flx_mix(a.z, b.z, t), flx_mix(a.w, b.w, t)); }

__global__ __launch_bounds__(256) void k_fxaa(Tex t, int W, int H, float4 *__restrict__ out) {
  int px, py;
  if (!texel_of_thread(W, H, px, py)) return;
#define FETCH(dx, dy) fetch(t, W, H, px + (dx), py + (dy))
#define LUMA(dx, dy) fxaa_luma(FETCH(dx, dy))
  float4 *dst = out + (size_t)(H - 1 - py) * W + px;
  const f4 original = FETCH(0, 0);
  float luma[3][3];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int i = 0; i < 3; i++) luma[j][i] = LUMA(i - 1, j - 1);
  const float edge_vert = flx_abs((0.25f * luma[0][0]) + (-0.5f * luma[0][1]) + (0.25f * luma[0][2])) +
                          flx_abs((0.50f * luma[1][0]) + (-1.0f * luma[1][1]) + (0.50f * luma[1][2])) +
                          flx_abs((0.25f * luma[2][0]) + (-0.5f * luma[2][1]) + (0.25f * luma[2][2]));
  const float edge_horz = flx_abs((0.25f * luma[0][0]) + (-0.5f * luma[1][0]) + (0.25f * luma[2][0])) +
                          flx_abs((0.50f * luma[0][1]) + (-1.0f * luma[1][1]) + (0.50f * luma[2][1])) +
                          flx_abs((0.25f * luma[0][2]) + (-0.5f * luma[1][2]) + (0.25f * luma[2][2]));
  const bool horz_span = edge_horz >= edge_vert;
  const int sx = horz_span ? 1 : 0, sy = horz_span ? 0 : 1;
  {
    const float c = luma[1][1], n = luma[0][1], w = luma[1][0], s = luma[2][1], e = luma[1][2];
    const float lo = flx_min(c, flx_min(flx_min(n, w), flx_min(s, e))), hi = flx_max(c, flx_max(flx_max(n, w), flx_max(s, e)));
    const float range = hi - lo;
    if (range < flx_max(1.0f / 32.0f, hi * 1.0f / 2.0f)) { *dst = make_float4(original.x, original.y, original.z, original.w); return; }
  }
  int nx = -sx, ny = -sy, qx = sx, qy = sy;
  f4 color = original;
  float pixel_count = 1.0f;
  bool done_n = false, done_p = false;
  const float luma_mcn = flx_max(flx_max(flx_abs(luma[0][1] - luma[1][1]), flx_abs(luma[1][2] - luma[1][1])),
                                 flx_max(flx_abs(luma[2][1] - luma[1][1]), flx_abs(luma[1][0] - luma[1][1])));
  const float gradient = flx_abs(luma_mcn - luma[1][1]);
  for (int i = 0; i < 6; i++) {
    int x, y;
    if (!done_n) { x = nx; y = ny; }
    else if (!done_p) { x = qx; y = qy; }
    else break;
    const f4 f00 = FETCH(x - 1, y - 1), f10 = FETCH(x, y - 1), f20 = FETCH(x + 1, y - 1);
    const f4 f01 = FETCH(x - 1, y), f11 = FETCH(x, y), f21 = FETCH(x + 1, y);
    const f4 f02 = FETCH(x - 1, y + 1), f12 = FETCH(x, y + 1), f22 = FETCH(x + 1, y + 1);
    f4 sum = add4(add4(add4(add4(add4(add4(add4(add4(f00, f10), f20), f01), f11), f21), f02), f12), f22);
    const f4 blur = scale4(sum, 1.0f / 9.0f);
    const bool done = flx_abs(fxaa_luma(blur) - luma_mcn) >= gradient;
    const float c = fxaa_luma(f11), n = fxaa_luma(f10), w = fxaa_luma(f01), s = fxaa_luma(f12), e = fxaa_luma(f21);
    const float luma_l = 0.25f * (((n + w) + e) + s);
    const float range_l = flx_abs(luma_l - c);
    const float lo = flx_min(c, flx_min(flx_min(n, w), flx_min(s, e))), hi = flx_max(c, flx_max(flx_max(n, w), flx_max(s, e)));
    const float range = hi - lo;
    float blend = flx_max(0.0f, (range_l / range) - 0.0f) * 1.0f;
    blend = flx_min(7.0f / 8.0f, blend);
    color = add4(color, mix4(f11, blur, blend));
    pixel_count += 1.0f;
    if (!done_n) { done_n = done; nx -= sx; ny -= sy; }
    else { done_p = done; qx += sx; qy += sy; }
  }
  *dst = make_float4(color.x / pixel_count, color.y / pixel_count, color.z / pixel_count, color.w / pixel_count);
#undef FETCH
#undef LUMA
}

struct TaaRing { const uint32_t *p[9]; };          /* newest first; null = not rendered yet (zero texture) */
__global__ __launch_bounds__(256) void k_taa(TaaRing r, int W, int H, float4 *__restrict__ out) {
  int x, y;
  if (!texel_of_thread(W, H, x, y)) return;
  const Tex t0 = { r.p[0] };
  f4 lo = F4(1.0f, 1.0f, 1.0f, 1.0f), hi = F4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const f4 p = fetch(t0, W, H, x + (i - 1), y + (j - 1));
      lo = F4(flx_min(lo.x, p.x), flx_min(lo.y, p.y), flx_min(lo.z, p.z), flx_min(lo.w, p.w));
      hi = F4(flx_max(hi.x, p.x), flx_max(hi.y, p.y), flx_max(hi.z, p.z), flx_max(hi.w, p.w));
    }
  f4 o = fetch(t0, W, H, x, y);
#pragma unroll
  for (int k = 1; k < 9; k++) {
    const Tex tk = { r.p[k] };
    const f4 c = fetch(tk, W, H, x, y);
    o = add4(o, F4(flx_min(flx_max(c.x, lo.x), hi.x), flx_min(flx_max(c.y, lo.y), hi.y), flx_min(flx_max(c.z, lo.z), hi.z), flx_min(flx_max(c.w, lo.w), hi.w)));
  }
  out[(size_t)(H - 1 - y) * W + x] = make_float4(o.x / 9.0f, o.y / 9.0f, o.z / 9.0f, o.w / 9.0f);
}

void launch_fxaa(const uint32_t *plane, float4 *out, int W, int H, hipStream_t stream) {
  const dim3 grid(((W + 15) >> 4) * ((H + 15) >> 4)), block(256);
  Tex t = { plane };
  hipLaunchKernelGGL(k_fxaa, grid, block, 0, stream, t, W, H, out);
}

void launch_taa(const uint32_t *const planes[9], float4 *out, int W, int H, hipStream_t stream) {
  const dim3 grid(((W + 15) >> 4) * ((H + 15) >> 4)), block(256);
  TaaRing r;
  for (int i = 0; i < 9; i++) r.p[i] = planes[i];
  hipLaunchKernelGGL(k_taa, grid, block, 0, stream, r, W, H, out);
}

void launch_quantize(const float4 *src, uint32_t *dst, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(k_quantize, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
}

void launch_temporal(const TemporalRings &rings, int W, int H, int hdr, int use_filter, uint32_t *dColor, uint32_t *dIp, float4 *out,
                     hipStream_t stream) {
  const dim3 grid(((W + 15) >> 4) * ((H + 15) >> 4)), block(256);
  hipLaunchKernelGGL(k_temporal, grid, block, 0, stream, rings, W, H, hdr, use_filter, dColor, dIp, out);
}

void launch_filter_chain(const FilterPlanes &pl, float4 *out, int W, int H, int hdr, hipStream_t stream) {
  const size_t n = (size_t)W * H;
  const dim3 grid(((W + 15) >> 4) * ((H + 15) >> 4)), block(256);
  (void)hipMemsetAsync(pl.O[1], 0, n * 4, stream);          /* read at pass 4 before anything wrote it this frame */
  int cur = 0, nId = 0, nOriginal = 0;
  for (int i = 0; i < 6; i++) {
    int np = (i % 2) ^ 1;
    const int npOriginal = ((i - 3) % 2) ^ 1;
    if (3 <= i) np += 2;
    uint32_t *third = nullptr;
    if (3 <= i - 2) third = pl.O[npOriginal];
    else if (np < 2) third = pl.Id[np];                       /* IdRenderTexture[2], [3] do not exist: output dropped */
    Tex tColor = { pl.R[cur] }, tIp = { pl.Ip[cur] }, tOColor = { pl.O[nOriginal] }, tId = { pl.Id[nId] }, tOId = { pl.OId };
    if (cur < 2) hipLaunchKernelGGL(k_filter_first, dim3(((W + FF_W - 1) / FF_W) * ((H + FF_H - 1) / FF_H)), block, 0, stream, tColor, tIp, tOColor, tId, tOId, pl.R[np], pl.Ip[np], third, W, H);
    else hipLaunchKernelGGL(k_filter_second, grid, block, 0, stream, tColor, tIp, tOColor, tId, tOId, pl.R[np], pl.Ip[np], third, W, H);
    cur = np;
    if (3 <= i) nOriginal = npOriginal; else nId = np;
  }
  Tex tColor = { pl.R[2] }, tIp = { pl.Ip[2] }, tOColor = { pl.O[1] }, tId = { pl.Id[1] }, tOId = { pl.OId };
  hipLaunchKernelGGL(k_filter_final, grid, block, 0, stream, tColor, tIp, tOColor, tId, tOId, out, W, H, hdr);
}

}  // namespace flx
