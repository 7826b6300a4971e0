/* flx_kernel_util.h — small device helpers shared by the kernel files. */
#ifndef FLX_KERNEL_UTIL_H
#define FLX_KERNEL_UTIL_H

#include "flx_device.h"

namespace flx {

/* A float channel as an RGBA8 render target stores it (modules/pathtracerWGL2.js:790-799): floor(clamp(x, 0, 1) * 255 + 0.5), NaN -> 0 */
__device__ __forceinline__ uint32_t quant_unorm8(float x) {
  if (!(x > 0.0f)) return 0u;
  if (x >= 1.0f) return 255u;
  return (uint32_t)(x * 255.0f + 0.5f);
}
__device__ __forceinline__ uint32_t pack_rgba8(float x, float y, float z, float w) {
  return quant_unorm8(x) | (quant_unorm8(y) << 8) | (quant_unorm8(z) << 16) | (quant_unorm8(w) << 24);
}

/* 8x8 pixel tile `tile` (row-major over the context's W x rows image), lane -> pixel. */
__device__ __forceinline__ void tile8_pixel(const DeviceFrame &fr, uint32_t tile, uint32_t lane, uint32_t &px, uint32_t &k) {
  const uint32_t tiles_x = (fr.width + 7u) >> 3;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  px = (tx << 3) + (lane & 7u);
  k = (ty << 3) + (lane >> 3);
}

/* Path item -> (screen tile, sample); the lane is item & 63. */
__device__ __forceinline__ void item_tile(const DeviceFrame &fr, uint32_t item, uint32_t &tile, uint32_t &s) {
  const uint32_t ts = item >> 6;
  if (fr.samples_shift >= 0) { tile = ts >> fr.samples_shift; s = ts & ((1u << fr.samples_shift) - 1u); }
  else { const uint32_t S = (uint32_t)fr.samples; tile = ts / S; s = ts - tile * S; }
}

/* Path item -> (pixel, sample).  Items are numbered [8x8 tile][sample][lane]: the 64 items a wave draws
 * together are one sample of one screen tile. Returns false for lanes outside the frame. */
__device__ __forceinline__ bool item_pixel(const DeviceFrame &fr, uint32_t item, uint32_t &px, uint32_t &k, uint32_t &s) {
  const uint32_t l = item & 63u;
  uint32_t tile;
  item_tile(fr, item, tile, s);
  tile8_pixel(fr, tile, l, px, k);
  return px < fr.width && k < fr.rows;
}

/* fragment:608-632 for one pixel: the samples added in order, averaged, times originalColor of the last sample (the shader's global still holds it after
 * the loop).  o: the pixel in the (stacked) planes, sampleStride: float4 between two samples' planes. */
__device__ __forceinline__ float4 resolve_pixel(const DeviceFrame &fr, const float4 *__restrict__ hits, const float4 *__restrict__ sampleRadiance, const float4 *__restrict__ lastOriginal,
                                                size_t o, size_t sampleStride, float *cost = nullptr /* -> the sum of the slots' w: what the pixel's paths cost (finalize_path) */) {
  float4 color = make_float4(0.f, 0.f, 0.f, 0.f);
  float c = 0.0f;
  if (__float_as_int(hits[o].w) != -1) {
    f3 finalColor = F3(0.0f, 0.0f, 0.0f);
    for (int s = 0; s < fr.samples; s++) {
      const float4 r = sampleRadiance[(size_t)s * sampleStride + o];
      finalColor = finalColor + F3(r.x, r.y, r.z);
      c += r.w;
    }
    if (cost) *cost = c;
    const float invSamples = 1.0f / (float)fr.samples;
    finalColor = finalColor * invSamples;
    const float4 oc = lastOriginal[o];
    finalColor = finalColor * F3(oc.x, oc.y, oc.z);
    if (fr.is_temporal == 1) color = make_float4(flx_fract(finalColor.x), flx_fract(finalColor.y), flx_fract(finalColor.z), 1.0f);
    else color = make_float4(finalColor.x, finalColor.y, finalColor.z, 1.0f);
  }
  return color;
}

/* rank of this lane among the set bits of `mask` below it */
__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool COUNT>
__device__ __forceinline__ void flush_counters(const WorkCounters &c, unsigned long long *out) {
  if (!COUNT) return;
  uint32_t v[8] = { c.primary_visits, c.closest_visits, c.shadow_visits, c.closest_walks, c.shadow_walks, c.shades, c.primary_hits, c.atlas_texels };
#pragma unroll
  for (int j = 0; j < 8; j++) {
    unsigned long long x = v[j];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63u) == 0u && x) atomicAdd(out + j, x);
  }
}

}  // namespace flx
#endif
