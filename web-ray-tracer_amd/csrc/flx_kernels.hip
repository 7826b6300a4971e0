/*
 * flx_kernels.hip — HIP kernels of the path tracer for gfx950 (wave64, no MFMA: divergent traversal).
 *
 * Launch geometry: the frame a context renders is W x rows pixels (rows = the packed rows its tile
 * policy owns).  A workgroup is 256 threads = 4 waves; each wave covers an 8x8 pixel tile (lanes of a
 * wave are neighbours on screen, so their rays enter the same part of the skip list and their 48-byte
 * entry loads hit the same cache lines); a workgroup covers 16x16.
 */
/* the kernels of this file shade inline between walks: there the table form of the sin / cos polynomial measures faster (theater
 * 10.22 -> 10.10 ms, cornell.obj filter frame 1.055 -> 1.021 ms); the wavefront pipeline's dense shade kernels keep the selects
 * (dragon 8.07 against 8.18 ms with the table) */
#define FLX_SINCOS_TABLE 1
#define FLX_ANGLE_TABLE 1                  /* k_trace_pixels and k_paths read the per-triangle table (flx_device.h; k_paths: FLX_PATHS_ANGLE_TABLE) */
#include "flx_kernels.h"
#include "flx_kernel_util.h"

#ifndef FLX_PRIMARY_FWD
#define FLX_PRIMARY_FWD 1                  /* primary rays: the wave steps through the forward-ordered copy together (primaryWalkF; k_primary 0.424 -> 0.391 ms) */
#endif

namespace flx {

#ifndef FLX_TRACE_BLOCK
#define FLX_TRACE_BLOCK 256                /* threads of a k_trace_pixels workgroup: 256 = a 16 x 16 pixel tile, 64 = one of its four 8 x 8 quarters per workgroup (A/B) */
#endif
__device__ __forceinline__ void tile_pixel(const DeviceFrame &fr, uint32_t &px, uint32_t &k) {
  const uint32_t tiles_x = (fr.width + 15u) >> 4;
  const uint32_t tile = FLX_TRACE_BLOCK == 256 ? blockIdx.x : blockIdx.x >> 2;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const uint32_t wave = FLX_TRACE_BLOCK == 256 ? threadIdx.x >> 6 : blockIdx.x & 3u, lane = threadIdx.x & 63u;
  px = (tx << 4) + ((wave & 1u) << 3) + (lane & 7u);
  k = (ty << 4) + ((wave >> 1) << 3) + (lane >> 3);
}

/* ---- v1: one thread per pixel, the whole fragment program (fragment:601-646) --------------------- */
#ifndef FLX_TRACE_WAVES
#define FLX_TRACE_WAVES 4                  /* waves per SIMD the register allocation of k_trace_pixels must allow: 231 VGPRs / 2 waves left to
                                            * itself; 128 VGPRs + 376 B of scratch per lane at 4 waves is 14 % faster on the cornell.obj filter frame
                                            * (0.969 -> 0.832 ms; 5: 0.911, 6: 0.949, 8: 1.059; profiles/r02_ab_occupancy.txt) */
#endif
#ifndef FLX_TRACE_WAVES_BIG
#define FLX_TRACE_WAVES_BIG 6              /* the variant for scenes without the lockstep copy (the lane walk over the threaded copy): the dragon's
                                            * 1080p 8 spp 4 bounces filter frame 15.26 (4) 14.24 (5) 13.68 ms (6) */
#endif
template <bool COUNT, bool LOCK>
__global__ __launch_bounds__(FLX_TRACE_BLOCK, LOCK ? FLX_TRACE_WAVES : FLX_TRACE_WAVES_BIG) void k_trace_pixels(DeviceScene sc, DeviceFrame fr, float4 *__restrict__ out, GBufferPtrs gb,
                                                      unsigned long long *__restrict__ counters) {
  uint32_t px, k;
  tile_pixel(fr, px, k);
  WorkCounters cnt = {};
  const bool inImage = px < fr.width && k < fr.rows;
  PixelState ps;
  ps.ndc_x = ps.ndc_y = 0.0f;
  float viewDepthPerS = 0.0f;
  uint32_t frameIdx = 0;
  f3 dir0 = F3(0.0f, 0.0f, 1.0f), camera = F3(0.0f, 0.0f, 0.0f);
  if (inImage) {
    const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
    frameIdx = frame_index(fr, k);
    dir0 = primary_dir(fr, frameIdx, px, py_gl, ps.ndc_x, ps.ndc_y, viewDepthPerS);
    camera = frame_camera(fr, frameIdx);
  }
  Ray pr; pr.origin = camera; pr.dir = dir0;
#if FLX_PRIMARY_FWD
  Hit hit0 = primaryWalkF(sc, inImage, pr, viewDepthPerS, cnt.primary_visits);      /* the wave walks together: every lane goes in */
#else
  Hit hit0; hit0.suv = F3(0.0f, 0.0f, 0.0f); hit0.transformId = 0; hit0.triangleId = -1;
  if (inImage) hit0 = primaryWalkT(sc, pr, viewDepthPerS, cnt.primary_visits);
#endif
  if (inImage) {
    ps.firstRayLength = 1.0f; ps.glassFilter = 0.0f; ps.originalRMEx = 0.0f; ps.originalTPOx = 0.0f;
    ps.originalColor = F3(0.0f, 0.0f, 0.0f);
    ps.renderId.x = ps.renderId.y = ps.renderId.z = ps.renderId.w = 0.0f;
    ps.renderOriginalId = ps.renderId;
    ps.seed = fr.view[frameIdx].random_seed;
    const size_t o = (size_t)k * fr.width + px;
    float4 color = make_float4(0.f, 0.f, 0.f, 0.f), colorIp = color, origColor = color, rid = color, roid = color, loc = color;
    if (hit0.triangleId != -1) {
      if (COUNT) cnt.primary_hits++;
      f3 finalColor = F3(0.0f, 0.0f, 0.0f);
      /* every sample's first bounce lands on the primary hit: what its shading knows before it draws a random number
       * (fetches, normals, the acos / tan of the normal deviation, material) is computed once */
      SurfaceCtx sf0;
      WorkCounters sfCnt = {};
      const bool firstBounce = fr.max_reflections > 0 && length(F3(1.0f, 1.0f, 1.0f) * F3(1.0f, 1.0f, 1.0f)) >= fr.min_importancy * SQRT3;
      if (firstBounce) shadeSurface<COUNT>(sc, fr, hit0, pr, camera, sf0, sfCnt);
      for (int s = 0; s < fr.samples; s++) {
        float cosSampleN = flx_cos((float)s);
        PathState p;
        p.dontFilter = true;
        p.finalColor = F3(0.0f, 0.0f, 0.0f);
        p.importancyFactor = F3(1.0f, 1.0f, 1.0f);
        ps.originalColor = F3(1.0f, 1.0f, 1.0f);
        p.ray.origin = camera; p.ray.dir = dir0;
        p.lastHitPoint = camera;
        p.hit = hit0;
        bool alive = firstBounce;
        if (firstBounce) {
          if (COUNT) { cnt.shades += sfCnt.shades; cnt.atlas_texels += sfCnt.atlas_texels; }      /* counted per path, as when each path shades it */
          alive = bounceOn<COUNT, LOCK>(sc, fr, sf0, ps, p, camera, cosSampleN, 0, cnt);
        }
        for (int i = 1; alive && i < fr.max_reflections && length(p.importancyFactor * ps.originalColor) >= fr.min_importancy * SQRT3; i++) {
          if (!bounce<COUNT, LOCK>(sc, fr, ps, p, camera, cosSampleN, i, cnt)) break;
        }
        finalColor = finalColor + (p.finalColor + p.importancyFactor * frame_ambient(fr, frameIdx));
      }
      float invSamples = 1.0f / (float)fr.samples;
      finalColor = finalColor * invSamples;
      if (fr.use_filter == 1) {
        color = make_float4(flx_fract(finalColor.x), flx_fract(finalColor.y), flx_fract(finalColor.z), 1.0f);
        colorIp = make_float4(flx_floor(finalColor.x) * INV_256, flx_floor(finalColor.y) * INV_256, flx_floor(finalColor.z) * INV_256, ps.glassFilter);
      } else {
        finalColor = finalColor * ps.originalColor;
        if (fr.is_temporal == 1) {
          color = make_float4(flx_fract(finalColor.x), flx_fract(finalColor.y), flx_fract(finalColor.z), 1.0f);
          colorIp = make_float4(flx_floor(finalColor.x) * INV_256, flx_floor(finalColor.y) * INV_256, flx_floor(finalColor.z) * INV_256, 1.0f);
        } else {
          color = make_float4(finalColor.x, finalColor.y, finalColor.z, 1.0f);
        }
      }
      origColor = make_float4(ps.originalColor.x, ps.originalColor.y, ps.originalColor.z, flx_min(ps.originalRMEx, ps.firstRayLength) + INV_255);
      rid = make_float4(ps.renderId.x, ps.renderId.y, ps.renderId.z, ps.renderId.w + INV_255);
      roid = make_float4(0.0f, 0.0f, 0.0f, ps.originalTPOx + INV_255);
      if (gb.location_id) {                  /* fragment:640-642; relativePosition in object space, camera in world space, as in the shader */
        const float4 g0 = sc.geometry[3 * hit0.triangleId], g1 = sc.geometry[3 * hit0.triangleId + 1], g2 = sc.geometry[3 * hit0.triangleId + 2];
        const float w0 = 1.0f - hit0.suv.y - hit0.suv.z;
        const f3 rel = (F3(g0.x, g0.y, g0.z) * w0 + F3(g0.w, g1.x, g1.y) * hit0.suv.y) + F3(g1.z, g1.w, g2.x) * hit0.suv.z;
        const float div = 2.0f * distance(rel, camera);
        loc = make_float4(flx_mod(rel.x, div) / div, flx_mod(rel.y, div) / div, flx_mod(rel.z, div) / div, INV_255);
      }
    }
    if (out) out[o] = color;
    if (gb.color) gb.color[o] = color;
    if (gb.color_ip) gb.color_ip[o] = colorIp;
    if (gb.original_color) gb.original_color[o] = origColor;
    if (gb.id) gb.id[o] = rid;
    if (gb.original_id) gb.original_id[o] = roid;
    if (gb.location_id) gb.location_id[o] = loc;
    /* filter frames nobody asked the float G-buffers of: straight into the chain's RGBA8 render targets */
    if (gb.q_color) gb.q_color[o] = pack_rgba8(color.x, color.y, color.z, color.w);
    if (gb.q_color_ip) gb.q_color_ip[o] = pack_rgba8(colorIp.x, colorIp.y, colorIp.z, colorIp.w);
    if (gb.q_original_color) gb.q_original_color[o] = pack_rgba8(origColor.x, origColor.y, origColor.z, origColor.w);
    if (gb.q_id) gb.q_id[o] = pack_rgba8(rid.x, rid.y, rid.z, rid.w);
    if (gb.q_original_id) gb.q_original_id[o] = pack_rgba8(roid.x, roid.y, roid.z, roid.w);
  }
  flush_counters<COUNT>(cnt, counters);
}

/* ---- v1s: the per-pixel kernel with the SAMPLES of a pixel side by side (round 5; profiles/r05_sample_parallel.txt) --------------------------------
 * k_trace_pixels runs a pixel's samples one after the other because the shader's globals carry state from sample to sample (fragment:83-89: firstRayLength,
 * glassFilter, originalRMEx, originalTPOx, renderId, renderOriginalId; originalColor of the LAST sample scales the sum, :632).  That makes a wave 12 bounces long on
 * configs[1] (4 spp x 3 bounces) and a 1080p frame 5 073 covered waves over 4 096 wave slots: a second round that is a quarter full.
 * None of those globals is READ by a sample's own arithmetic — they are written (+=, =, min) and reach the G-buffers after the loop — so the samples of a pixel
 * are independent and what each bounce would have done to the globals can be written down and applied afterwards, in the shader's order:
 *   workgroup = one 8 x 8 screen tile x S samples, wave s = sample s of the tile's 64 pixels (neighbouring pixels per wave, as before: the lockstep walk stays coherent);
 *   wave 0 walks the tile's primary rays (the wave-wide walk of k_primary) and leaves the hits in LDS; every wave shades the surface for its sample (the per-triangle
 *   table makes that cheap: round 2's sample-parallel forms lost to three acos + three tan in double per lane) and runs its path's bounces with the globals zeroed before
 *   every bounce, so that afterwards they hold exactly what the bounce contributes: 0 + x = x for the sums, the assigned value for `=`, min(x, +inf) = x for the minimum
 *   (NaN included); the contributions go to LDS (24 B per bounce and lane);
 *   after a barrier wave 0 replays them per pixel in sample-major, bounce-minor order — the order of the sequential loop — adds the samples' colours in sample order and
 *   writes the pixel.  Same floats in the same order: bit-identical G-buffers (tests/test_parity_gpu.py's filter cases, tests/test_filter_parity_gpu.py).
 * S = 2, 4 or 8 and at most FLX_TS_MAX_BOUNCES bounces (the log must fit in LDS at four workgroups per CU); anything else runs k_trace_pixels. */
#ifndef FLX_TS_MAX_BOUNCES
#define FLX_TS_MAX_BOUNCES 4
#endif
#ifndef FLX_TS_WAVES
#define FLX_TS_WAVES 4                     /* waves per SIMD the register allocation must allow */
#endif
template <bool COUNT, bool LOCK, int S>
__global__ __launch_bounds__(64 * S, FLX_TS_WAVES) void k_trace_samples(DeviceScene sc, DeviceFrame fr, float4 *__restrict__ out, GBufferPtrs gb,
                                                                        unsigned long long *__restrict__ counters, int maxB) {
  /* LDS: [64 hits][64 x originalColor of the last sample][S x 64 per-sample results: colour.xyz, firstRayLength candidate | renderOriginalId.xyz, flag bits]
   *      [maxB x S x 64 log entries of 24 B: rme.x, renderId.xyz contribution, tpo.x, renderId.w] */
  extern __shared__ float4 ldsTS[];
  float4 *ldsHit = ldsTS;
  float4 *ldsLast = ldsTS + 64;
  float4 *ldsRes = ldsTS + 128;
  float2 *ldsLog = (float2 *)(ldsRes + 2 * S * 64);
  const uint32_t lane = threadIdx.x & 63u, smp = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x;
  uint32_t px, k;
  tile8_pixel(fr, tile, lane, px, k);
  WorkCounters cnt = {};
  const bool inImage = px < fr.width && k < fr.rows;
  PixelState ps;
  ps.ndc_x = ps.ndc_y = 0.0f;
  float viewDepthPerS = 0.0f;
  uint32_t frameIdx = 0;
  f3 dir0 = F3(0.0f, 0.0f, 1.0f), camera = F3(0.0f, 0.0f, 0.0f);
  if (inImage) {
    const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
    frameIdx = frame_index(fr, k);
    dir0 = primary_dir(fr, frameIdx, px, py_gl, ps.ndc_x, ps.ndc_y, viewDepthPerS);
    camera = frame_camera(fr, frameIdx);
  }
  Ray pr; pr.origin = camera; pr.dir = dir0;
  if (smp == 0u) {
    const Hit h = primaryWalkF(sc, inImage, pr, viewDepthPerS, cnt.primary_visits);      /* the wave walks together: every lane goes in */
    ldsHit[lane] = make_float4(h.suv.x, h.suv.y, h.suv.z, __int_as_float(inImage ? h.triangleId : -1));
  }
  __syncthreads();
  Hit hit0;
  { const float4 h = ldsHit[lane]; hit0.suv = F3(h.x, h.y, h.z); hit0.triangleId = __float_as_int(h.w); hit0.transformId = 0; }
  const bool covered = hit0.triangleId != -1;
  const bool anyCovered = flx_ballot(covered) != 0ull;        /* (every wave looks at the same 64 hits: uniform over the workgroup) */
  if (!anyCovered && smp != 0u) return;                        /* sky: nothing to trace; wave 0 writes the tile's pixels below (no barrier follows on this path) */
  if (covered) hit0.transformId = (int)sc.geometry[3 * hit0.triangleId + 2].y << 1;      /* (what primaryWalkF's Hit carries: 2 x the entry's transform number) */
  const uint32_t slot = smp * 64u + lane;
  float4 res0 = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7f800000)), res1 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t bits = 0;                                           /* per bounce i, bits 4i .. 4i + 3: present, tpo assigned, renderId.w assigned, glassFilter's increment */
  if (covered) {
    if (COUNT && smp == 0u) cnt.primary_hits++;
    ps.seed = fr.view[frameIdx].random_seed;
    ps.originalTPOx = 0.0f;
    const bool firstBounce = fr.max_reflections > 0 && length(F3(1.0f, 1.0f, 1.0f) * F3(1.0f, 1.0f, 1.0f)) >= fr.min_importancy * SQRT3;
    const float cosSampleN = flx_cos((float)smp);
    PathState p;
    p.dontFilter = true;
    p.finalColor = F3(0.0f, 0.0f, 0.0f);
    p.importancyFactor = F3(1.0f, 1.0f, 1.0f);
    ps.originalColor = F3(1.0f, 1.0f, 1.0f);
    p.ray.origin = camera; p.ray.dir = dir0;
    p.lastHitPoint = camera;
    p.hit = hit0;
    bool alive = firstBounce;
    for (int i = 0; alive && i < fr.max_reflections && (i == 0 || length(p.importancyFactor * ps.originalColor) >= fr.min_importancy * SQRT3); i++) {
      /* the globals as the bounce finds them when nothing has touched them: what it leaves is its contribution */
      const bool oldDontFilter = p.dontFilter;
      ps.originalRMEx = 0.0f; ps.glassFilter = 0.0f; ps.firstRayLength = __int_as_float(0x7f800000);
      ps.renderId.x = ps.renderId.y = ps.renderId.z = ps.renderId.w = 0.0f;
      ps.renderOriginalId = ps.renderId;
      bool goesOn;
      if (i == 0) {
        SurfaceCtx sf0;
        shadeSurface<COUNT>(sc, fr, hit0, pr, camera, sf0, cnt);
        goesOn = bounceOn<COUNT, LOCK>(sc, fr, sf0, ps, p, camera, cosSampleN, 0, cnt);
        res1 = make_float4(ps.renderOriginalId.x, ps.renderOriginalId.y, ps.renderOriginalId.z, 0.0f);
      } else {
        goesOn = bounce<COUNT, LOCK>(sc, fr, ps, p, camera, cosSampleN, i, cnt);
        if (i == 1) res0.w = ps.firstRayLength;
      }
      bits |= (1u | (oldDontFilter ? 2u : 0u) | ((p.dontFilter || i == 0) ? 4u : 0u) | (ps.glassFilter != 0.0f ? 8u : 0u)) << (4 * i);
      float2 *L = ldsLog + ((size_t)(i * S) * 64u + slot) * 3u;
      L[0] = make_float2(ps.originalRMEx, ps.renderId.x);
      L[1] = make_float2(ps.renderId.y, ps.renderId.z);
      L[2] = make_float2(ps.originalTPOx, ps.renderId.w);
      alive = goesOn;
    }
    const f3 v = p.finalColor + p.importancyFactor * frame_ambient(fr, frameIdx);      /* fragment:613 for this sample */
    res0.x = v.x; res0.y = v.y; res0.z = v.z;
    res1.w = __int_as_float((int)bits);
    if (smp == (uint32_t)(S - 1)) ldsLast[lane] = make_float4(ps.originalColor.x, ps.originalColor.y, ps.originalColor.z, 0.0f);
  }
  if (anyCovered) {
    ldsRes[slot * 2u] = res0; ldsRes[slot * 2u + 1u] = res1;
    __syncthreads();
  }
  if (smp != 0u) { flush_counters<COUNT>(cnt, counters); return; }
  /* ---- wave 0: the pixel, with the samples' contributions applied in the shader's order ---- */
  if (inImage) {
    ps.firstRayLength = 1.0f; ps.glassFilter = 0.0f; ps.originalRMEx = 0.0f; ps.originalTPOx = 0.0f;
    ps.originalColor = F3(0.0f, 0.0f, 0.0f);
    ps.renderId.x = ps.renderId.y = ps.renderId.z = ps.renderId.w = 0.0f;
    ps.renderOriginalId = ps.renderId;
    const size_t o = (size_t)k * fr.width + px;
    float4 color = make_float4(0.f, 0.f, 0.f, 0.f), colorIp = color, origColor = color, rid = color, roid = color, loc = color;
    if (covered) {
      f3 finalColor = F3(0.0f, 0.0f, 0.0f);
      for (int s = 0; s < S; s++) {
        const uint32_t sl = (uint32_t)s * 64u + lane;
        const float4 r0 = ldsRes[sl * 2u], r1 = ldsRes[sl * 2u + 1u];
        uint32_t fl = (uint32_t)__float_as_int(r1.w);
        for (int i = 0; (fl & 1u) != 0u; i++, fl >>= 4) {      /* the bounces the sample ran: a prefix */
          const float2 *L = ldsLog + ((size_t)(i * S) * 64u + sl) * 3u;
          const float2 a = L[0], b = L[1], c = L[2];
          if (fl & 2u) {                                        /* fragment:544-558, while the path's dontFilter held */
            ps.originalTPOx = c.x;
            ps.originalRMEx += a.x;
            if (fr.use_filter) {
              ps.renderId.x += a.y; ps.renderId.y += b.x; ps.renderId.z += b.y; ps.renderId.w += 0.0f;
              if (i == 0) { ps.renderOriginalId.x += r1.x; ps.renderOriginalId.y += r1.y; ps.renderOriginalId.z += r1.z; ps.renderOriginalId.w += 0.0f; }
            }
            if (fl & 8u) ps.glassFilter += 1.0f;
          }
          if (i == 1) ps.firstRayLength = flx_min(r0.w, ps.firstRayLength);      /* fragment:565 */
          if (fl & 4u) ps.renderId.w = c.y;                     /* fragment:437 (and :1275's + 1 / 255 where the light was shadowed) */
        }
        finalColor = finalColor + F3(r0.x, r0.y, r0.z);
      }
      { const float4 lc = ldsLast[lane]; ps.originalColor = F3(lc.x, lc.y, lc.z); }      /* the global holds the LAST sample's after the loop (fragment:632) */
      float invSamples = 1.0f / (float)fr.samples;
      finalColor = finalColor * invSamples;
      if (fr.use_filter == 1) {
        color = make_float4(flx_fract(finalColor.x), flx_fract(finalColor.y), flx_fract(finalColor.z), 1.0f);
        colorIp = make_float4(flx_floor(finalColor.x) * INV_256, flx_floor(finalColor.y) * INV_256, flx_floor(finalColor.z) * INV_256, ps.glassFilter);
      } else {
        finalColor = finalColor * ps.originalColor;
        if (fr.is_temporal == 1) {
          color = make_float4(flx_fract(finalColor.x), flx_fract(finalColor.y), flx_fract(finalColor.z), 1.0f);
          colorIp = make_float4(flx_floor(finalColor.x) * INV_256, flx_floor(finalColor.y) * INV_256, flx_floor(finalColor.z) * INV_256, 1.0f);
        } else {
          color = make_float4(finalColor.x, finalColor.y, finalColor.z, 1.0f);
        }
      }
      origColor = make_float4(ps.originalColor.x, ps.originalColor.y, ps.originalColor.z, flx_min(ps.originalRMEx, ps.firstRayLength) + INV_255);
      rid = make_float4(ps.renderId.x, ps.renderId.y, ps.renderId.z, ps.renderId.w + INV_255);
      roid = make_float4(0.0f, 0.0f, 0.0f, ps.originalTPOx + INV_255);
      if (gb.location_id) {
        const float4 g0 = sc.geometry[3 * hit0.triangleId], g1 = sc.geometry[3 * hit0.triangleId + 1], g2 = sc.geometry[3 * hit0.triangleId + 2];
        const float w0 = 1.0f - hit0.suv.y - hit0.suv.z;
        const f3 rel = (F3(g0.x, g0.y, g0.z) * w0 + F3(g0.w, g1.x, g1.y) * hit0.suv.y) + F3(g1.z, g1.w, g2.x) * hit0.suv.z;
        const float div = 2.0f * distance(rel, camera);
        loc = make_float4(flx_mod(rel.x, div) / div, flx_mod(rel.y, div) / div, flx_mod(rel.z, div) / div, INV_255);
      }
    }
    if (out) out[o] = color;
    if (gb.color) gb.color[o] = color;
    if (gb.color_ip) gb.color_ip[o] = colorIp;
    if (gb.original_color) gb.original_color[o] = origColor;
    if (gb.id) gb.id[o] = rid;
    if (gb.original_id) gb.original_id[o] = roid;
    if (gb.location_id) gb.location_id[o] = loc;
    if (gb.q_color) gb.q_color[o] = pack_rgba8(color.x, color.y, color.z, color.w);
    if (gb.q_color_ip) gb.q_color_ip[o] = pack_rgba8(colorIp.x, colorIp.y, colorIp.z, colorIp.w);
    if (gb.q_original_color) gb.q_original_color[o] = pack_rgba8(origColor.x, origColor.y, origColor.z, origColor.w);
    if (gb.q_id) gb.q_id[o] = pack_rgba8(rid.x, rid.y, rid.z, rid.w);
    if (gb.q_original_id) gb.q_original_id[o] = pack_rgba8(roid.x, roid.y, roid.z, roid.w);
  }
  flush_counters<COUNT>(cnt, counters);
}

template <bool LOCK, int S>
static void launch_trace_samples_s(const DeviceScene &sc, const DeviceFrame &fr, float4 *out, const GBufferPtrs &gb, unsigned long long *counters, hipStream_t stream) {
  const uint32_t tiles = ((fr.width + 7u) >> 3) * ((fr.rows + 7u) >> 3);
  const int maxB = fr.max_reflections > 0 ? fr.max_reflections : 1;
  const size_t lds = (size_t)(128 + 2 * S * 64) * sizeof(float4) + (size_t)maxB * S * 64 * 24;
  if (counters) hipLaunchKernelGGL((k_trace_samples<true, LOCK, S>), dim3(tiles), dim3(64 * S), lds, stream, sc, fr, out, gb, counters, maxB);
  else hipLaunchKernelGGL((k_trace_samples<false, LOCK, S>), dim3(tiles), dim3(64 * S), lds, stream, sc, fr, out, gb, counters, maxB);
}
/* the samples of a pixel side by side (k_trace_samples) where the frame allows it; false: not such a frame */
static bool launch_trace_samples(const DeviceScene &sc, const DeviceFrame &fr, float4 *out, const GBufferPtrs &gb, unsigned long long *counters, hipStream_t stream, bool lock) {
  if (fr.max_reflections > FLX_TS_MAX_BOUNCES || !(fr.samples == 2 || fr.samples == 4 || fr.samples == 8)) return false;
  if ((size_t)(128 + 2 * fr.samples * 64) * sizeof(float4) + (size_t)(fr.max_reflections > 0 ? fr.max_reflections : 1) * fr.samples * 64 * 24 > 64u * 1024u) return false;      /* (8 samples x 4 bounces: 67 KB) */
  if (lock) {
    if (fr.samples == 2) launch_trace_samples_s<true, 2>(sc, fr, out, gb, counters, stream);
    else if (fr.samples == 4) launch_trace_samples_s<true, 4>(sc, fr, out, gb, counters, stream);
    else launch_trace_samples_s<true, 8>(sc, fr, out, gb, counters, stream);
  } else {
    if (fr.samples == 2) launch_trace_samples_s<false, 2>(sc, fr, out, gb, counters, stream);
    else if (fr.samples == 4) launch_trace_samples_s<false, 4>(sc, fr, out, gb, counters, stream);
    else launch_trace_samples_s<false, 8>(sc, fr, out, gb, counters, stream);
  }
  return true;
}

void launch_trace_pixels(const DeviceScene &sc, const DeviceFrame &fr, float4 *out, const GBufferPtrs &gb,
                         unsigned long long *counters, hipStream_t stream, int sample_parallel) {
  const uint32_t tiles = ((fr.width + 15u) >> 4) * ((fr.rows + 15u) >> 4);
  const bool lock = FLX_LOCKSTEP && sc.lock_entries != 0u;      /* small scene in one object space: the variant with the wave-wide walk */
  if (sample_parallel && launch_trace_samples(sc, fr, out, gb, counters, stream, lock)) return;
  if (lock) {
    if (counters) hipLaunchKernelGGL((k_trace_pixels<true, true>), dim3(tiles * (256u / FLX_TRACE_BLOCK)), dim3(FLX_TRACE_BLOCK), 0, stream, sc, fr, out, gb, counters);
    else hipLaunchKernelGGL((k_trace_pixels<false, true>), dim3(tiles * (256u / FLX_TRACE_BLOCK)), dim3(FLX_TRACE_BLOCK), 0, stream, sc, fr, out, gb, counters);
  } else {
    if (counters) hipLaunchKernelGGL((k_trace_pixels<true, false>), dim3(tiles * (256u / FLX_TRACE_BLOCK)), dim3(FLX_TRACE_BLOCK), 0, stream, sc, fr, out, gb, counters);
    else hipLaunchKernelGGL((k_trace_pixels<false, false>), dim3(tiles * (256u / FLX_TRACE_BLOCK)), dim3(FLX_TRACE_BLOCK), 0, stream, sc, fr, out, gb, counters);
  }
}

/* ---- v2: primary kernel + persistent path kernel with lane refill + resolve ---------------------- */
/*
 * Work item = one (pixel, sample) path.  Items are numbered [8x8 tile][sample][lane] so that the 64
 * items a wave draws together are one sample of one screen tile.  A wave is persistent: every lane
 * runs "one bounce per loop trip" (shade -> shadow walk + closest-hit walk -> fold), and at the top of
 * each trip the lanes whose path has ended draw fresh items (wave-level compaction / restart): the
 * wave64 never idles on the tail of its longest path, and all lanes are always in the same stage,
 * so the long shading code is never executed for a handful of lanes.  Items come from a global
 * counter in chunks of PATH_CHUNK per wave (one atomic per chunk, ~65 k atomics per 1080p x 8 frame).
 * Each path writes its radiance to its own slot; k_resolve adds the samples of a pixel in sample
 * order, so the frame equals the sequential shader's bit for bit.
 */
constexpr uint32_t PATH_CHUNK = 256;

template <bool COUNT>
__global__ __launch_bounds__(256) void k_primary(DeviceScene sc, DeviceFrame fr, float4 *__restrict__ hits,
                                                 unsigned long long *__restrict__ counters) {
  const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
  uint32_t px, k;
  tile8_pixel(fr, tile, threadIdx.x & 63u, px, k);
  WorkCounters cnt = {};
  const bool inImage = px < fr.width && k < fr.rows;
#if FLX_PRIMARY_FWD
  {   /* the wave walks together: every lane goes in, the ones without a pixel with no ray */
    float nx, ny, viewDepthPerS = 0.0f;
    Ray pr; pr.origin = F3(0.0f, 0.0f, 0.0f); pr.dir = F3(0.0f, 0.0f, 1.0f);
    if (inImage) {
      const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
      const uint32_t frameIdx = frame_index(fr, k);
      pr.dir = primary_dir(fr, frameIdx, px, py_gl, nx, ny, viewDepthPerS);
      pr.origin = frame_camera(fr, frameIdx);
    }
    Hit h = primaryWalkF(sc, inImage, pr, viewDepthPerS, cnt.primary_visits);
    if (inImage) {
      if (COUNT && h.triangleId != -1) cnt.primary_hits++;
      hits[(size_t)k * fr.width + px] = make_float4(h.suv.x, h.suv.y, h.suv.z, __int_as_float(h.triangleId));
    }
  }
#else
  if (inImage) {
    const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
    float nx, ny, viewDepthPerS;
    Ray pr;
    const uint32_t frameIdx = frame_index(fr, k);
    pr.dir = primary_dir(fr, frameIdx, px, py_gl, nx, ny, viewDepthPerS);
    pr.origin = frame_camera(fr, frameIdx);
    Hit h = primaryWalkT(sc, pr, viewDepthPerS, cnt.primary_visits);
    if (COUNT && h.triangleId != -1) cnt.primary_hits++;
    hits[(size_t)k * fr.width + px] = make_float4(h.suv.x, h.suv.y, h.suv.z, __int_as_float(h.triangleId));
  }
#endif
  flush_counters<COUNT>(cnt, counters);
}

#ifndef FLX_PATHS_WAVES
#define FLX_PATHS_WAVES 4                  /* k_paths: 163 VGPRs / 3 waves per SIMD left to itself.  Theater 1080p 16 spp 6 bounces by waves per SIMD, round 2 (lane walk):
                                            * 11.74 (3) 11.37 (4) 11.14 (5) 10.50 (6) 10.25 (7) 11.04 ms (8) (profiles/r02_ab_occupancy.txt) — and swept again late in round 4, the
                                            * kernel leaner by then (profiles/r04_paths_occupancy.txt, one box): lane walk 9.54 (4) 10.31 (5) 10.03 (6) 10.17 ms (7); with the
                                            * wave's lockstep walk 10.19 (3) 9.13 (4) 9.50 (5) 9.80 (6) 9.92 ms (7): 128 registers and the lockstep walk, which needs no waves to
                                            * hide a fetch behind */
#endif
/* FLX_PATHS_KERNARG: the scene and the frame stay in the kernarg segment and are read where they are used (what flx_frame_common.h does for the frame kernels: by-value structs are
 * loaded whole at the kernel's entry and kept — here 83 spilled scalar registers, 182 lane reads / writes in the bounce loop) */
#ifndef FLX_PATHS_KERNARG
#define FLX_PATHS_KERNARG 0
#endif
struct PathsArgs { DeviceScene sc; DeviceFrame fr; };
typedef const __attribute__((address_space(4))) PathsArgs *PathsArgsP;
__device__ __forceinline__ const PathsArgs &paths_args(PathsArgsP p) { asm volatile("" : "+s"(p)); return *(const PathsArgs *)p; }
#if FLX_PATHS_KERNARG
#define PATHS_ARGS() const PathsArgs &PA_ = paths_args((PathsArgsP)__builtin_amdgcn_kernarg_segment_ptr()); const DeviceScene &sc = PA_.sc; const DeviceFrame &fr = PA_.fr; (void)sc; (void)fr
#else
#define PATHS_ARGS() const DeviceScene &sc = pa.sc; const DeviceFrame &fr = pa.fr; (void)sc; (void)fr
#endif
template <bool COUNT, bool LOCK>
__global__ __launch_bounds__(256, FLX_PATHS_WAVES) void k_paths(PathsArgs pa /* the kernel's FIRST parameter: offset 0 of the kernarg segment */, const float4 *__restrict__ hits,
                                               float4 *__restrict__ sampleRadiance, float4 *__restrict__ lastOriginal,
                                               uint32_t *__restrict__ queue, uint32_t total_items,
                                               unsigned long long *__restrict__ counters) {
  /* (round 4 at first: this kernel's shading computed the per-triangle angle terms itself — at seven waves per SIMD the table's load was one more dependent fetch in a kernel that hid
   * its latencies with waves: theater 9.96 ms without the table, 10.03 with it, profiles/r04_angle_table.txt.  At four waves per SIMD the arithmetic saved counts: flx_kernels.h) */
  if (!FLX_PATHS_ANGLE_TABLE && !FLX_PATHS_KERNARG) pa.sc.angle_tan = nullptr;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t S; size_t P;
  { PATHS_ARGS(); S = (uint32_t)fr.samples; P = (size_t)fr.rows * fr.width; }
  f3 camera = F3(0.0f, 0.0f, 0.0f);             /* of the frame the lane's path belongs to */
  uint32_t frameIdx = 0;
  WorkCounters cnt = {};
  bool alive = false;
  PathState p;
  PixelState ps;
  int bounceIdx = 0;
  float cosSampleN = 0.0f;
  size_t slot = 0;              /* where this path's radiance goes */
  uint32_t sampleIdx = 0;
  uint32_t chunkNext = 0, chunkEnd = 0;      /* wave-uniform */
  bool itemsLeft = true;                     /* wave-uniform */
  auto finishPath = [&]() {                  /* fragment:598 + what main() needs from the last sample */
    PATHS_ARGS();
    const f3 r = p.finalColor + p.importancyFactor * frame_ambient(fr, frameIdx);
    sampleRadiance[slot] = make_float4(r.x, r.y, r.z, 1.0f);
    if (sampleIdx == S - 1u)
      lastOriginal[slot - (size_t)sampleIdx * P] = make_float4(ps.originalColor.x, ps.originalColor.y, ps.originalColor.z, 1.0f);
  };

  for (;;) {
    /* -- refill: dead lanes draw new items until the wave is full or the queue is dry ------------ */
    for (;;) {
      const unsigned long long idle = __ballot(!alive);
      if (idle == 0ull) break;
      PATHS_ARGS();
      if (chunkNext == chunkEnd) {
        if (!itemsLeft) break;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(queue, PATH_CHUNK);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= total_items) { itemsLeft = false; break; }
        chunkNext = base;
        chunkEnd = (base + PATH_CHUNK < total_items) ? base + PATH_CHUNK : total_items;
      }
      const uint32_t nIdle = (uint32_t)__popcll(idle);
      const uint32_t avail = chunkEnd - chunkNext;
      const uint32_t take = nIdle < avail ? nIdle : avail;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      if (!alive && rank < take) {
        const uint32_t item = chunkNext + rank;
        const uint32_t l = item & 63u, ts = item >> 6;
        const uint32_t s = ts % S, tile = ts / S;
        uint32_t px, k;
        tile8_pixel(fr, tile, l, px, k);
        if (px < fr.width && k < fr.rows) {
          const size_t o = (size_t)k * fr.width + px;
          const float4 h = hits[o];
          const int tri = __float_as_int(h.w);
          if (tri != -1) {
            const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
            float viewDepthPerS;
            frameIdx = frame_index(fr, k);
            camera = frame_camera(fr, frameIdx);
            ps.seed = fr.view[frameIdx].random_seed;
            const f3 dir0 = primary_dir(fr, frameIdx, px, py_gl, ps.ndc_x, ps.ndc_y, viewDepthPerS);
            ps.firstRayLength = 1.0f; ps.glassFilter = 0.0f; ps.originalRMEx = 0.0f; ps.originalTPOx = 0.0f;
            ps.renderId.x = ps.renderId.y = ps.renderId.z = ps.renderId.w = 0.0f;
            ps.renderOriginalId = ps.renderId;
            ps.originalColor = F3(1.0f, 1.0f, 1.0f);
            p.dontFilter = true;
            p.finalColor = F3(0.0f, 0.0f, 0.0f);
            p.importancyFactor = F3(1.0f, 1.0f, 1.0f);
            p.ray.origin = camera; p.ray.dir = dir0;
            p.lastHitPoint = camera;
            p.hit.suv = F3(h.x, h.y, h.z);
            p.hit.triangleId = tri;
            p.hit.transformId = (int)sc.geometry[3 * tri + 2].y << 1;
            cosSampleN = flx_cos((float)s);
            bounceIdx = 0;
            sampleIdx = s;
            slot = (size_t)s * P + o;
            /* loop guard of fragment:475 before the first bounce (fails only for bounces = 0 or minImportancy > 1) */
            alive = fr.max_reflections > 0 && length(p.importancyFactor * ps.originalColor) >= fr.min_importancy * SQRT3;
            if (!alive) finishPath();
          }
        }
      }
      chunkNext += take;
    }
    if (__ballot(alive) == 0ull) break;

    /* -- one bounce for every live lane (fragment:475-596) ------------------------------------------ */
    if (alive) {
      PATHS_ARGS();
      bool cont = bounce<COUNT, LOCK>(sc, fr, ps, p, camera, cosSampleN, bounceIdx, cnt);
      bounceIdx++;
      if (cont) cont = bounceIdx < fr.max_reflections && length(p.importancyFactor * ps.originalColor) >= fr.min_importancy * SQRT3;
      if (!cont) { finishPath(); alive = false; }
    }
  }
  flush_counters<COUNT>(cnt, counters);
}

/* fragment:608-632, one thread per pixel (flx_kernel_util.h: resolve_pixel) */
/* tileTime (or nullptr): += what the paths of every 8 x 8 screen tile cost (the w of their radiance slots: time in the frame kernel's walk lanes) — the measure the
 * next frame's tile order is made from (k_tile_order) */
__global__ __launch_bounds__(256) void k_resolve(DeviceFrame fr, const float4 *__restrict__ hits, const float4 *__restrict__ sampleRadiance,
                                                 const float4 *__restrict__ lastOriginal, float4 *__restrict__ out, size_t sampleStride, float *__restrict__ tileTime) {
  const size_t P = (size_t)fr.rows * fr.width;
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float cost = 0.0f;
  if (o < P) out[o] = resolve_pixel(fr, hits, sampleRadiance, lastOriginal, o, sampleStride, &cost);
  if (tileTime == nullptr) return;
  const uint32_t k = (uint32_t)(o / fr.width), px = (uint32_t)(o - (size_t)k * fr.width);
  const uint32_t tile = (k >> 3) * ((fr.width + 7u) >> 3) + (px >> 3);
  if ((fr.width & 7u) == 0u) {                              /* eight consecutive lanes = the eight pixels of a tile's row: one atomic per group */
    cost += __shfl_xor(cost, 1); cost += __shfl_xor(cost, 2); cost += __shfl_xor(cost, 4);
    if ((threadIdx.x & 7u) == 0u && o < P && cost != 0.0f) atomicAdd(tileTime + tile, cost);
  } else if (o < P && cost != 0.0f) atomicAdd(tileTime + tile, cost);
}

/* The frame kernel's draw order over the frame's n screen tiles, from what the tiles cost in the last frame (k_resolve: tileTime; cleared here for the next one).
 * Tiles are put into classes by cost (2 048 logarithmic bins: the float's exponent and three mantissa bits) and written class by class, in screen order inside a
 * class — neighbouring tiles walk the same part of the tree, and an order that scatters them costs more than it gains (tools/tile_order_ab.py: heaviest-first exactly
 * 6.0 ms, random 6.7, against 5.77 in screen order).  mode 0: two classes, the lightest tenth of the tiles last (a launch ends in the chains of the paths drawn last:
 * short ones there; the dragon frame 5.77 -> 5.65 ms); mode 1: sixteen classes of equal size, heaviest first (thin frames whose front ran in its own kernel: the
 * items are drawn 64 paths at a time by all workgroups, the longest chains should start first; a rank's eighth 1.45 -> 1.38 ms).  One workgroup. */
#define FLX_DEV_K __device__ __forceinline__
constexpr uint32_t TO_THREADS = 512, TO_BINS = 2048, TO_CLASSES = 16, TO_BPT = TO_BINS / TO_THREADS;      /* (58 KB of LDS) */
FLX_DEV_K uint32_t to_bin(float c) { return c > 0.0f ? (__float_as_uint(c) >> 20) & (TO_BINS - 1u) : 0u; }      /* (positive floats order like their bits; bit 31 is clear) */
/* inclusive scan of one value per thread over the workgroup (wave scans by shuffles, the wave totals through LDS) */
FLX_DEV_K uint32_t to_block_scan(uint32_t v, uint32_t *waveTotals /* [waves] */, uint32_t t) {
  const uint32_t lane = t & 63u, wave = t >> 6;
  for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t u = __shfl_up(v, d); if (lane >= d) v += u; }
  if (lane == 63u) waveTotals[wave] = v;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wave; w++) before += waveTotals[w];
  __syncthreads();
  return v + before;
}
__global__ __launch_bounds__(TO_THREADS) void k_tile_order(float *__restrict__ tileTime, uint32_t *__restrict__ order, uint32_t n, int mode) {
  __shared__ uint32_t hist[TO_BINS];
  __shared__ uint8_t binClass[TO_BINS];
  __shared__ uint16_t counts[TO_CLASSES * TO_THREADS];        /* [class][thread]: tiles of the class in the thread's run (a run is < 2^16 tiles: frames of up to 2^25 tiles) ... */
  __shared__ uint32_t offs[TO_CLASSES * TO_THREADS];          /* ... and where the first of them goes */
  __shared__ uint32_t waveTotals[TO_THREADS / 64u];
  const uint32_t t = threadIdx.x;
  for (uint32_t b = t; b < TO_BINS; b += TO_THREADS) hist[b] = 0u;
  for (uint32_t j = t; j < TO_CLASSES * TO_THREADS; j += TO_THREADS) counts[j] = 0;
  __syncthreads();
  for (uint32_t i = t; i < n; i += TO_THREADS) atomicAdd(&hist[to_bin(tileTime[i])], 1u);
  __syncthreads();
  {   /* bins -> classes: tiles in lighter bins (an exclusive prefix over the bins, TO_BPT consecutive bins per thread) */
    uint32_t h[TO_BPT], sum = 0;
    for (uint32_t k = 0; k < TO_BPT; k++) { h[k] = hist[TO_BPT * t + k]; sum += h[k]; }
    uint32_t below = to_block_scan(sum, waveTotals, t) - sum;
    for (uint32_t k = 0; k < TO_BPT; k++) {
      uint32_t c;
      if (mode == 0) c = (below + h[k]) * 10u <= n ? 1u : 0u;      /* the bins that lie wholly within the lightest tenth: last */
      else { c = (uint32_t)(((uint64_t)below * TO_CLASSES) / (n ? n : 1u)); c = TO_CLASSES - 1u - (c < TO_CLASSES ? c : TO_CLASSES - 1u); }
      binClass[TO_BPT * t + k] = (uint8_t)c;
      below += h[k];
    }
  }
  __syncthreads();
  /* a stable partition by class: every thread owns a run of consecutive tiles; count per (class, thread), scan in class-major order, write */
  const uint32_t chunk = (n + TO_THREADS - 1u) / TO_THREADS, lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
  for (uint32_t i = lo; i < hi; i++) counts[(uint32_t)binClass[to_bin(tileTime[i])] * TO_THREADS + t]++;
  __syncthreads();
  {   /* thread j owns the 16 consecutive entries j * 16 .. of the flattened [class][thread] table */
    uint32_t sum = 0;
    for (uint32_t k = 0; k < TO_CLASSES; k++) sum += counts[t * TO_CLASSES + k];
    const uint32_t incl = to_block_scan(sum, waveTotals, t);
    uint32_t at = incl - sum;
    for (uint32_t k = 0; k < TO_CLASSES; k++) { offs[t * TO_CLASSES + k] = at; at += counts[t * TO_CLASSES + k]; }
  }
  __syncthreads();
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t c = binClass[to_bin(tileTime[i])];
    order[offs[c * TO_THREADS + t]++] = i;
    tileTime[i] = 0.0f;                                       /* for the next frame's sums */
  }
}
void launch_tile_order(float *tileTime, uint32_t *order, uint32_t n, int mode, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(TO_THREADS), 0, stream, tileTime, order, n, mode);
}

/* DeviceScene::angle_tan: per triangle entry what every shade of it would compute (flx_device.h: triangleAngleTanOf), once per scene / transform upload */
__global__ __launch_bounds__(256) void k_angle_tan(DeviceScene sc, float4 *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= sc.n_entries) return;
  float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const float kind = sc.geometry[3 * (size_t)i + 2].z;
  if (kind != 0.0f && kind != 1.0f) {                         /* a triangle entry (1: a box, 0: the end of the list / padding) — flx_api.hip: build_threaded reads the kind the same way */
    const f3 t = triangleAngleTanOf(sc, (int)i);
    r = make_float4(t.x, t.y, t.z, 0.0f);
  }
  out[i] = r;
}
void launch_angle_tan(const DeviceScene &sc, float4 *out, hipStream_t stream) {
  if (sc.n_entries == 0u) return;
  DeviceScene s = sc;
  s.angle_tan = nullptr;
  hipLaunchKernelGGL(k_angle_tan, dim3((sc.n_entries + 255u) / 256u), dim3(256), 0, stream, s, out);
}

uint64_t path_item_count64(const DeviceFrame &fr) {
  return (uint64_t)((fr.width + 7u) >> 3) * ((fr.rows + 7u) >> 3) * (uint64_t)fr.samples * 64u;
}
uint32_t path_item_count(const DeviceFrame &fr) { return (uint32_t)path_item_count64(fr); }      /* callers have checked it fits */

void launch_primary(const DeviceScene &sc, const DeviceFrame &fr, float4 *hits, unsigned long long *counters, hipStream_t stream) {
  const uint32_t tiles = ((fr.width + 7u) >> 3) * ((fr.rows + 7u) >> 3);
  const uint32_t blocks = (tiles + 3u) / 4u;
  if (counters) hipLaunchKernelGGL(k_primary<true>, dim3(blocks), dim3(256), 0, stream, sc, fr, hits, counters);
  else hipLaunchKernelGGL(k_primary<false>, dim3(blocks), dim3(256), 0, stream, sc, fr, hits, counters);
}

void launch_paths(const DeviceScene &sc, const DeviceFrame &fr, const float4 *hits, float4 *sampleRadiance, float4 *lastOriginal,
                  uint32_t *queue, uint32_t blocks, unsigned long long *counters, hipStream_t stream) {
  const uint32_t total = path_item_count(fr);
  const bool lock = FLX_LOCKSTEP && sc.lock_entries != 0u;
  PathsArgs pa;
  pa.sc = sc; pa.fr = fr;
  if (FLX_PATHS_KERNARG && !FLX_PATHS_ANGLE_TABLE) pa.sc.angle_tan = nullptr;
  if (lock) {
    if (counters) hipLaunchKernelGGL((k_paths<true, true>), dim3(blocks), dim3(256), 0, stream, pa, hits, sampleRadiance, lastOriginal, queue, total, counters);
    else hipLaunchKernelGGL((k_paths<false, true>), dim3(blocks), dim3(256), 0, stream, pa, hits, sampleRadiance, lastOriginal, queue, total, counters);
  } else {
    if (counters) hipLaunchKernelGGL((k_paths<true, false>), dim3(blocks), dim3(256), 0, stream, pa, hits, sampleRadiance, lastOriginal, queue, total, counters);
    else hipLaunchKernelGGL((k_paths<false, false>), dim3(blocks), dim3(256), 0, stream, pa, hits, sampleRadiance, lastOriginal, queue, total, counters);
  }
}

void launch_resolve(const DeviceFrame &fr, const float4 *hits, const float4 *sampleRadiance, const float4 *lastOriginal, float4 *out,
                    hipStream_t stream, size_t sampleStride, float *tileTime) {
  const size_t P = (size_t)fr.rows * fr.width;
  hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((P + 255) / 256)), dim3(256), 0, stream, fr, hits, sampleRadiance, lastOriginal, out, sampleStride ? sampleStride : P, tileTime);
}

/* ---- diagnostics: include/flx_math.h on the device ------------------------------------------------ */
__global__ void k_debug_math(int fn, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
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

/* ---- diagnostics: the intersection routines as the kernels call them, one row per thread (flx_debug_intersect) ---- */
__global__ void k_debug_intersect(int fn, const float *__restrict__ in, float *__restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  DeviceScene sc = {};
  sc.walk_fast_boxes = 1u;                                    /* the rows' boxes are small: the precondition of the reciprocal box test holds */
  if (fn == 2 || fn == 5) {                                   /* rayCuboid: l, origin, dir, min, max */
    const float *r = in + (size_t)i * 13u;
    WalkState w;
    w.tR.origin = F3(r[1], r[2], r[3]); w.tR.dir = F3(r[4], r[5], r[6]);
    reciprocalOfDir(sc, w.tR.dir, w.tR.origin, w.inv, w.fastDiv);
    const f3 lo = F3(r[7], r[8], r[9]), hi = F3(r[10], r[11], r[12]);
    out[i] = (fn == 2 ? rayCuboidFast(r[0], w, lo, hi) : rayCuboid(r[0], w.tR, lo, hi)) ? 1.0f : 0.0f;
    return;
  }
  const float *r = in + (size_t)i * 16u;                      /* triangles: a, b, c, origin, dir, l */
  const f3 a = F3(r[0], r[1], r[2]), b = F3(r[3], r[4], r[5]), c = F3(r[6], r[7], r[8]);
  Ray ray; ray.origin = F3(r[9], r[10], r[11]); ray.dir = F3(r[12], r[13], r[14]);
  const float l = r[15];
  f3 suv = F3(0.0f, 0.0f, 0.0f);
  bool hit;
  switch (fn) {
    case 0: hit = moellerTrumboreAny(a, b - a, c - a, ray, l, false, suv); break;      /* the walk kernels' routine over the stored edges */
    case 1: hit = moellerTrumboreAny(a, b - a, c - a, ray, l, true, suv); break;
    case 3: hit = moellerTrumbore(a, b, c, ray, l, suv); break;                         /* the per-pixel kernel's */
    default: hit = moellerTrumboreCull(a, b, c, ray, l); break;
  }
  if (fn == 0 || fn == 3) {
    float *o = out + (size_t)i * 3u;
    o[0] = hit ? suv.x : 0.0f; o[1] = hit ? suv.y : 0.0f; o[2] = hit ? suv.z : 0.0f;
  } else out[i] = hit ? 1.0f : 0.0f;
}

/* ---- diagnostics: the two walks of one ray, as the kernels run them (flx_debug_walk) ---------------------------------
 * in: 7 floats per ray (origin, direction, l); out: 8 floats per ray (s, u, v, 2 x transform, entry index of the closest hit or -1, entries the closest-hit
 * walk fetched, shadowTest's answer for length l, entries the shadow walk fetched).
 * variant 0: the wavefront pipeline's lane walk over the threaded copy (walkFetchP / walkBoxP / walkTriT, rays pre-transformed into every object space in LDS);
 * variant 1: walkBounce's lane walk (k_trace_pixels / k_paths without the lockstep copy); variant 2: the wave's lockstep walk (scenes that have the copy). */
template <int VARIANT>
__global__ __launch_bounds__(64) void k_debug_walk(DeviceScene sc, const float *__restrict__ in, float *__restrict__ out, uint32_t n) {
  extern __shared__ float4 ldsDebug[];
  const uint32_t lane = threadIdx.x, i = blockIdx.x * 64u + lane;
  const bool have = i < n;
  const float *r = in + (size_t)(have ? i : 0u) * 7u;
  Ray ray; ray.origin = F3(r[0], r[1], r[2]); ray.dir = F3(r[3], r[4], r[5]);
  const float l = r[6];
  WorkCounters cnt = {};
  Hit hit; hit.suv = F3(0.0f, 0.0f, 0.0f); hit.transformId = 0; hit.triangleId = -1;
  bool shadowed = false;
  if (VARIANT == 0) {
    const uint32_t T = sc.n_transforms;
    float4 *ldsXf = ldsDebug;
    float2 *myRays = (float2 *)(ldsDebug + (size_t)T * 4u) + (size_t)lane * T * 5u;
    for (uint32_t t = lane; t < T * 4u; t += 64u) {
      const uint32_t tr = t >> 2, k = t & 3u, iI = 2u * tr + 1u;
      ldsXf[t] = k < 3u ? sc.rotation[3u * iI + k] : sc.shift[iI];
    }
    __syncthreads();
    for (int mode = 0; mode < 2; mode++) {                   /* the shadow walk, then the closest-hit walk: the order of a path's two walks */
      const bool shadowMode = mode == 0;
      WalkState w;
      walkClearResults(w);
      w.src = ray; w.mode = mode;
      WalkEntry cur; cur.e0 = cur.e1 = cur.e2 = make_float4(0.f, 0.f, 0.f, 0.f);
      walkSetupRays(sc, T, ldsXf, myRays, ray, shadowMode);
      w.tR = ray; w.cachedTI = 0; w.minLen = shadowMode ? l : POW32; w.i = (int)sc.walk_root;
      reciprocalOfDir(sc, ray.dir, ray.origin, w.inv, w.fastDiv);
      bool ended = !have || walkFetchP<true>(sc, nullptr, 0u, myRays, w, cur, cnt);
      while (!ended) {
        if (walkIsBoxT(cur)) walkBoxP(w, cur); else ended = walkTriT(w, cur);
        if (!ended) ended = walkFetchP<true>(sc, nullptr, 0u, myRays, w, cur, cnt);
      }
      if (shadowMode) shadowed = w.shadowed != 0;
      else { hit.suv = w.suv; hit.transformId = w.hitTI; hit.triangleId = w.tri; }
    }
  } else {
    bool sh = false;
    walkBounce<true, VARIANT == 2>(sc, have, false, ray, l, ray, sh, hit, cnt);      /* (two calls: each walk's own counters) */
    shadowed = sh;
    Hit h2; h2.suv = F3(0.0f, 0.0f, 0.0f); h2.transformId = 0; h2.triangleId = -1;
    walkBounce<true, VARIANT == 2>(sc, false, have, ray, l, ray, sh, h2, cnt);
    hit = h2;
  }
  if (!have) return;
  float *o = out + (size_t)i * 8u;
  o[0] = hit.triangleId != -1 ? hit.suv.x : 0.0f; o[1] = hit.triangleId != -1 ? hit.suv.y : 0.0f; o[2] = hit.triangleId != -1 ? hit.suv.z : 0.0f;
  o[3] = (float)(hit.triangleId != -1 ? hit.transformId : 0); o[4] = (float)hit.triangleId;
  o[5] = (float)cnt.closest_visits; o[6] = shadowed ? 1.0f : 0.0f; o[7] = (float)cnt.shadow_visits;
}

bool launch_debug_walk(int variant, const DeviceScene &sc, const float *in, float *out, uint32_t n, hipStream_t stream) {
  const dim3 grid((n + 63u) / 64u), block(64);
  const size_t lds = (size_t)sc.n_transforms * 4u * sizeof(float4) + 64u * (size_t)sc.n_transforms * 40u;
  if (variant == 0) {
    if (lds > 60u * 1024u) return false;
    hipLaunchKernelGGL(k_debug_walk<0>, grid, block, lds, stream, sc, in, out, n);
  } else if (variant == 1) {
    DeviceScene s1 = sc; s1.lock_entries = 0u;
    hipLaunchKernelGGL(k_debug_walk<1>, grid, block, 0, stream, s1, in, out, n);
  } else {
    if (sc.lock_entries == 0u) return false;
    hipLaunchKernelGGL(k_debug_walk<2>, grid, block, 0, stream, sc, in, out, n);
  }
  return true;
}

void launch_debug_intersect(int fn, const float *in, float *out, uint32_t n, hipStream_t stream) {
  hipLaunchKernelGGL(k_debug_intersect, dim3((n + 63u) / 64u), dim3(64), 0, stream, fn, in, out, n);
}

void launch_debug_math(int fn, const float *a, const float *b, float *out, uint32_t n, hipStream_t stream) {
  hipLaunchKernelGGL(k_debug_math, dim3((n + 255u) / 256u), dim3(256), 0, stream, fn, a, b, out, n);
}

}  // namespace flx
