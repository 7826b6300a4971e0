// Issue rate of the vector instruction classes the walk trip is made of, on one MI355X: 8 independent chains per lane, 1024 threads (4 waves per SIMD) x 1024 workgroups.
// The question (profiles/r04_valu_rates.txt): bench.py prices the frame kernel's VALU issue against one wave64 instruction per 2 cycles and SIMD (v_fma_f32's rate,
// MI355X_MICROARCH.md) — do compares, selects, min / max, integer and logic operations issue at that rate too?
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; prints G wave-instructions/s and cycles per instruction and SIMD at 2.4 GHz for each class.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS8(OP)                                                                                 \
  for (int i = 0; i < iters; i++) {                                                                 \
    OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)                                          \
    OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)                                          \
  }

#define K(NAME, OP)                                                                                  \
  __global__ __launch_bounds__(1024) void NAME(float *out, int iters, float a, float b) {            \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    CHAINS8(OP)                                                                                      \
    out[blockIdx.x * 1024 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                     \
  }

#define OP_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define OP_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_SUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
#define OP_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define OP_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define OP_CMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
#define OP_CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
#define OP_CMPCND(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
#define OP_CMPS(x) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(x), "v"(a) : "s20", "s21");
#define OP_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define OP_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define OP_FMAABS(x) asm volatile("v_fma_f32 %0, -|%0|, %1, %0" : "+v"(x) : "v"(a));
#define OP_MBCNT(x) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(x));
#define OP_SUBREV(x) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(x) : "v"(b));
#define OP_BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(a), "v"(b));
#define OP_CNDS(x) asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(x) : "v"(a));
#define OP_ASHR(x) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x));
#define OP_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define OP_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(a));

K(k_fma, OP_FMA) K(k_mul, OP_MUL) K(k_add, OP_ADD) K(k_sub, OP_SUB) K(k_max, OP_MAX) K(k_min, OP_MIN) K(k_mov, OP_MOV) K(k_and, OP_AND) K(k_addu, OP_ADDU)
K(k_bfi, OP_BFI) K(k_cnds, OP_CNDS) K(k_ashr, OP_ASHR) K(k_max3, OP_MAX3) K(k_xor, OP_XOR)
K(k_lshl, OP_LSHL) K(k_cmp, OP_CMP) K(k_cnd, OP_CND) K(k_cmpcnd, OP_CMPCND) K(k_cmps, OP_CMPS) K(k_rcp, OP_RCP) K(k_med3, OP_MED3) K(k_fmaabs, OP_FMAABS) K(k_mbcnt, OP_MBCNT)

struct Case { const char *name; void (*fn)(float *, int, float, float); int per_op; };
int main() {
  float *out; hipMalloc(&out, 1024 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2048;
  const Case cases[] = { { "v_fma_f32", k_fma, 1 }, { "v_mul_f32", k_mul, 1 }, { "v_add_f32", k_add, 1 }, { "v_sub_f32", k_sub, 1 }, { "v_max_f32", k_max, 1 }, { "v_min_f32", k_min, 1 },
                         { "v_med3_f32", k_med3, 1 }, { "v_fma_f32 with -|x|", k_fmaabs, 1 }, { "v_mov_b32", k_mov, 1 }, { "v_and_b32", k_and, 1 }, { "v_add_u32", k_addu, 1 }, { "v_lshlrev_b32", k_lshl, 1 },
                         { "v_mbcnt_lo_u32_b32", k_mbcnt, 1 }, { "v_bfi_b32", k_bfi, 1 }, { "v_xor_b32", k_xor, 1 }, { "v_ashrrev_i32", k_ashr, 1 }, { "v_max3_f32", k_max3, 1 }, { "v_cndmask_b32 (sgpr mask)", k_cnds, 1 }, { "v_cmp_lt_f32 -> vcc", k_cmp, 1 }, { "v_cmp_lt_f32 -> sgpr pair", k_cmps, 1 }, { "v_cndmask_b32 (vcc)", k_cnd, 1 },
                         { "v_cmp_lt_f32 + v_cndmask_b32", k_cmpcnd, 2 }, { "v_rcp_f32", k_rcp, 1 } };
  int cus = 256; hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
  printf("%-32s %10s %22s %28s\n", "instruction (wave64)", "ms", "G wave-instructions/s", "cycles per instr and SIMD @2.4GHz");
  for (const Case &c : cases) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(c.fn, dim3(1024), dim3(1024), 0, 0, out, iters, 0.999f, 0.001f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double insts = 1024.0 * 16 * iters * 16 * c.per_op;       // workgroups x waves x iterations x 16 operations
    const double rate = insts / (best * 1e-3);
    printf("%-32s %10.3f %22.1f %28.2f\n", c.name, best, rate / 1e9, (double)cus * 4 * 2.4e9 / rate);
  }
  return 0;
}
