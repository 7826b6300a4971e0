/* Shared by the wavefront pipeline's kernels (flx_wavefront.hip, flx_walkq.hip): path-record flags, list constants,
 * the store of a finished path's radiance. */
#pragma once
#include "flx_kernels.h"
#include "flx_kernel_util.h"

namespace flx {

constexpr uint32_t WF_INVALID = 0xffffffffu;
constexpr uint32_t WF_IN_CHUNK = 256;       /* path ids a wave draws from the walk queue per atomic */
constexpr uint32_t WF_OUT_CHUNK = 256;      /* live-list slots a wave reserves per atomic */

/* states of a lane of the walk kernel (and of a suspended walk in the straggler list) */
enum { P_EMPTY = 0, P_WALKING = 1, P_DONE = 2, P_SWITCH = 3, P_SETUP = 4, P_RESUME = 5 };

/* record flags (q0.w as int bits) */
constexpr int RF_DEAD = 1, RF_DONT_FILTER = 2, RF_NEED_SHADOW = 4, RF_SHADOWED_NO_WALK = 8;
constexpr int RF_NO_CLOSEST = 16;      /* the loop guard ends the path after this bounce: its closest-hit walk would reach no output (nextBounceRuns) */

/* q0 origin.xyz flags | q1 nextDir.xyz shadowLen | q2 shadowOrigin.xyz baseLuminance  (after the walk: hit s,u,v,tri)
 * q3 shadowDir.xyz bounce | q4 litColor.xyz - | q5 finalColor.xyz - | q6 importancyFactor.xyz - | q7 originalColor.xyz - */

template <bool LV = false>
__device__ __forceinline__ void finalize_path(const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t pathId, f3 finalColor,
                                              f3 importancy, f3 originalColor, const FrameView *lv = nullptr, float cost = 1.0f /* the slot's w: what the path cost (k_resolve sums it per screen tile) */) {
  uint32_t px, k, s;
  item_pixel(fr, pathId, px, k, s);
  const size_t P = (size_t)fr.rows * fr.width;
  const size_t o = (size_t)k * fr.width + px;
  const f3 r = finalColor + importancy * view_ambient(view_at<LV>(fr, lv, frame_index(fr, k)));          /* fragment:598 */
  wb.sampleRadiance[(size_t)s * P + o] = make_float4(r.x, r.y, r.z, cost);
  if (s == (uint32_t)fr.samples - 1u) wb.lastOriginal[o] = make_float4(originalColor.x, originalColor.y, originalColor.z, 1.0f);
}


/* flx_walkcoop.hip */
void launch_walk_coop(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t compute_units, bool count, int b,
                      hipStream_t stream);
/* flx_walkq.hip */
void launch_walk_queue(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t compute_units, bool count, int b,
                       uint32_t total, hipStream_t stream);

}  // namespace flx
