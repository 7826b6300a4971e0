/* flx_kernels.h — host-visible launchers of the HIP kernels (flx_kernels.hip). */
#ifndef FLX_KERNELS_H
#define FLX_KERNELS_H

#include "flx_device.h"

namespace flx {

struct GBufferPtrs { float4 *color, *color_ip, *original_color, *id, *original_id; };

/* counters: 8 x u64 in flx_counters order, or nullptr (no counting code is compiled in). */
void launch_trace_pixels(const DeviceScene &sc, const DeviceFrame &fr, float4 *out, const GBufferPtrs &gb,
                         unsigned long long *counters, hipStream_t stream);
/* v2 pipeline: primary hits (float4 s,u,v,triangleId-as-bits per pixel) -> persistent path kernel -> resolve. */
uint32_t path_item_count(const DeviceFrame &fr);
void launch_primary(const DeviceScene &sc, const DeviceFrame &fr, float4 *hits, unsigned long long *counters, hipStream_t stream);
void launch_paths(const DeviceScene &sc, const DeviceFrame &fr, const float4 *hits, float4 *sampleRadiance, float4 *lastOriginal,
                  uint32_t *queue, uint32_t blocks, unsigned long long *counters, hipStream_t stream);
void launch_resolve(const DeviceFrame &fr, const float4 *hits, const float4 *sampleRadiance, const float4 *lastOriginal, float4 *out,
                    hipStream_t stream);
void launch_debug_math(int fn, const float *a, const float *b, float *out, uint32_t n, hipStream_t stream);

}  // namespace flx
#endif
