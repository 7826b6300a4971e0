// Can a RUNNING kernel see what the host posts?  Three mailboxes: (a) pinned host memory (hipHostMallocCoherent), written by a CPU store;
// (b) fine-grained device memory (hipExtMallocWithFlags hipDeviceMallocFinegrained), written by hipMemcpyAsync on a second stream;
// (c) ordinary device memory, written the same way.  The kernel polls with system-scope relaxed loads and gives up after ~50 ms.
// build: hipcc -O2 --offload-arch=gfx950 mailbox.hip -o mailbox
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_wait(const uint32_t *flag, uint32_t want, long long *out, long long budget) {
  const long long t0 = wall_clock64();
  long long polls = 0;
  uint32_t v = 0;
  for (;;) {
    v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    polls++;
    if (v == want) break;
    if (wall_clock64() - t0 > budget) break;
    __builtin_amdgcn_s_sleep(32);
  }
  out[0] = v; out[1] = wall_clock64() - t0; out[2] = polls;
}
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  long long *out; CK(hipHostMalloc(&out, 64, hipHostMallocDefault));
  uint32_t *stage; CK(hipHostMalloc(&stage, 64, hipHostMallocDefault));
  const long long budget = 100000000ll / 20;      // wall_clock64 ticks at 100 MHz: 50 ms
  for (int kind = 0; kind < 3; kind++) {
    uint32_t *box = nullptr;
    if (kind == 0) CK(hipHostMalloc(&box, 64, hipHostMallocCoherent | hipHostMallocMapped));
    if (kind == 1) { hipError_t e = hipExtMallocWithFlags((void **)&box, 64, hipDeviceMallocFinegrained); if (e != hipSuccess) { printf("finegrained device memory: %s\n", hipGetErrorString(e)); continue; } }
    if (kind == 2) CK(hipMalloc(&box, 64));
    for (int rep = 0; rep < 3; rep++) {
      const uint32_t want = 100u + rep;
      if (kind == 0) box[0] = 0; else { stage[0] = 0; CK(hipMemcpy(box, stage, 4, hipMemcpyHostToDevice)); }
      out[0] = out[1] = out[2] = -1;
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s1, box, want, out, budget);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      if (kind == 0) { __atomic_store_n(&box[0], want, __ATOMIC_RELEASE); }
      else { stage[0] = want; CK(hipMemcpyAsync(box, stage, 4, hipMemcpyHostToDevice, s2)); }
      CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
      printf("%-28s rep %d: kernel saw %lld (wanted %u) after %.3f ms, %lld polls -> %s\n",
             kind == 0 ? "pinned host, CPU store" : kind == 1 ? "fine-grained device, memcpy" : "ordinary device, memcpy", rep, out[0], want, out[1] / 100000.0, out[2],
             out[0] == (long long)want ? "SEEN" : "not seen");
    }
  }
  return 0;
}
