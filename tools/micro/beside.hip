// Does a small kernel on another stream run BESIDE a persistent launch that occupies G of the 256 CUs (1024 threads + 156 KB of LDS per workgroup: one per CU)?
// The persistent kernel spins for ~30 ms; the small kernel (256-thread blocks, no LDS) is launched 2 ms after it, on a stream of its own / of another priority.
// build: hipcc -O2 --offload-arch=gfx950 beside.hip -o beside
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k_hold(long long ticks, unsigned *xcc) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 1.0f;
  if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); atomicAdd(&xcc[id & 7u], 1u); }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (lds[threadIdx.x] == 2.0f) xcc[8] = 1;
}
__global__ void k_small(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 2.0f + 1.0f; }
int main() {
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t sHold, sSmall, sSmallHi;
  CK(hipStreamCreateWithPriority(&sHold, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithFlags(&sSmall, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sSmallHi, hipStreamNonBlocking, hi));
  CK(hipFuncSetAttribute((const void *)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  float *buf; CK(hipMalloc(&buf, 1 << 22)); CK(hipMemset(buf, 0, 1 << 22));
  unsigned *xcc; CK(hipMalloc(&xcc, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("stream priorities: lowest %d, highest %d\n", lo, hi);
  for (int G : { 256, 248, 240, 224, 192, 128 }) {
    for (int which = 0; which < 2; which++) {
      CK(hipMemset(xcc, 0, 64));
      hipLaunchKernelGGL(k_hold, dim3(G), dim3(1024), 156 * 1024, sHold, 3000000ll /* 30 ms at 100 MHz */, xcc);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      hipStream_t s = which ? sSmallHi : sSmall;
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_small, dim3(1024), dim3(256), 0, s, buf, 1 << 18);
      CK(hipStreamSynchronize(s));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      CK(hipStreamSynchronize(sHold));
      unsigned h[8]; CK(hipMemcpy(h, xcc, 32, hipMemcpyDeviceToHost));
      printf("persistent grid %3d (per XCC: %u %u %u %u %u %u %u %u), small kernel on a %s stream: done after %.3f ms -> %s\n", G, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7],
             which ? "highest-priority" : "normal", ms, ms < 20.0 ? "ran BESIDE" : "WAITED for the persistent launch");
    }
  }
  return 0;
}
