// Issue rate of v_fma_f32 against v_pk_fma_f32 on one MI355X: 8 independent chains per lane, 1024 threads x 1024 workgroups.
// build: hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip ; prints Gop/s (instructions x 64 lanes) for both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a, float b) {
  if (PK) {
    v2f x[8];
    for (int j = 0; j < 8; j++) x[j] = v2f{ (float)threadIdx.x + j, (float)j };
    const v2f aa = { a, a }, bb = { b, b };
    for (int i = 0; i < iters; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = __builtin_elementwise_fma(x[j], aa, bb);
    float s = 0; for (int j = 0; j < 8; j++) s += x[j].x + x[j].y;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
  } else {
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)threadIdx.x + j;
    for (int i = 0; i < iters; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = __builtin_fmaf(x[j], a, b);
    float s = 0; for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
  }
}
int main() {
  float *out; hipMalloc(&out, 1024 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  for (int pk = 0; pk < 2; pk++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(1024), 0, 0, out, iters, 0.999f, 0.001f);
      else hipLaunchKernelGGL(k<0>, dim3(1024), dim3(1024), 0, 0, out, iters, 0.999f, 0.001f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.3f ms, %.1f G wave-instructions/s, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms,
                      1024.0 * 16 * iters * 8 / ms / 1e6, 1024.0 * 1024 * iters * 8 * (pk ? 4 : 2) / ms / 1e9);
    }
  }
  return 0;
}
