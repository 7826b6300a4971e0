/* flx_kernels.h — host-visible launchers of the HIP kernels (flx_kernels.hip). */
#ifndef FLX_KERNELS_H
#define FLX_KERNELS_H

#include "flx_device.h"

namespace flx {

struct GBufferPtrs { float4 *color, *color_ip, *original_color, *id, *original_id; };

/* counters: 8 x u64 in flx_counters order, or nullptr (no counting code is compiled in). */
void launch_trace_pixels(const DeviceScene &sc, const DeviceFrame &fr, float4 *out, const GBufferPtrs &gb,
                         unsigned long long *counters, hipStream_t stream);
void launch_debug_math(int fn, const float *a, const float *b, float *out, uint32_t n, hipStream_t stream);

}  // namespace flx
#endif
