// Micro-benchmark: instruction issue on gfx950 — does scalar work overlap vector work across the waves
// of a SIMD?  Each wave runs ITER x (block of NV dependent-free VALU ops [+ NS SALU ops]).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
  uint32_t s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 * 3, s2 = s0 + 5, s3 = s0 ^ 7;
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE != 2) {   // 8 independent VALU
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(a2));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(a3)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(a4));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(a5)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a5) : "v"(a6));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(a7)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a7) : "v"(a0));
      }
      if (MODE != 0) {   // 8 SALU
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s3) : "scc"); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s3) : "s"(s0) : "scc");
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s3) : "scc"); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s3) : "s"(s0) : "scc");
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3;
}
template <int MODE> float run(int bpc, uint32_t iters) {
  uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256 * bpc, 256>>>(out, 10, 1); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<256 * bpc, 256>>>(out, iters, 1); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); hipFree(out); return ms;
}
int main() {
  const uint32_t it = 20000;
  for (int bpc : {1, 2, 4, 8}) {
    float v = run<0>(bpc, it), m = run<1>(bpc, it), s = run<2>(bpc, it);
    double nv = (double)it * 64;   // VALU (or SALU) instructions per wave
    // cycles per instruction per SIMD at 2.4 GHz: waves per SIMD = bpc
    printf("blocks/CU %d (waves/SIMD %d): VALU-only %.3f ms (%.2f cyc/instr/SIMD)  VALU+SALU %.3f ms  SALU-only %.3f ms (%.2f cyc/instr/CU-scalar)\n",
           bpc, bpc, v, v * 1e-3 * 2.4e9 / (nv * bpc), m, s, s * 1e-3 * 2.4e9 / (nv * bpc * 4));
  }
}
