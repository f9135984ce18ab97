// Micro-benchmark: issue cost of the instruction kinds the fused sweep is made of, 8 waves per SIMD (8 workgroups of 256 per
// CU), 8 independent chains per wave.  Prints cycles per instruction per SIMD (at the clock hipDeviceProp reports).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t iters, uint32_t seed) {
  uint32_t a[8], b = threadIdx.x * 7 + seed, c = threadIdx.x ^ seed;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * (i + 3) + seed;
  uint32_t s = __builtin_amdgcn_readfirstlane(seed) | 1;
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#define V_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define V_ADD_DPP(i) asm volatile("v_add_u32_dpp %0, %1, %0 row_newbcast:" #i " row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define V_ADD_DPPQ(i) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define V_ADD_S(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
#define V_ADDC(i) asm volatile("v_add_u32_e64 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
#define V_MIN3(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define V_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define V_MIN(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define V_CMP_CND(i) asm volatile("v_cmp_le_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define V_CMPS_CND(i) asm volatile("v_cmp_le_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c) : "s20", "s21");
#define V_RDLANE(i) { uint32_t t; asm volatile("v_readlane_b32 %0, %1, " #i : "=s"(t) : "v"(a[i])); s += t; }
#define V_RDLANE_ONLY(i) { uint32_t t; asm volatile("v_readlane_b32 %0, %1, " #i : "=s"(t) : "v"(a[i])); }
#define S_ADD(i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
#define S_NOP(i) asm volatile("s_nop 0");
#define S_WAIT(i) asm volatile("s_waitcnt vmcnt(0)");
      if (MODE == 0) { REP8(V_ADD) }
      if (MODE == 1) { REP8(V_ADD_DPP) }
      if (MODE == 2) { REP8(V_ADD_DPPQ) }
      if (MODE == 3) { REP8(V_ADD_S) }
      if (MODE == 4) { REP8(V_ADDC) }
      if (MODE == 5) { REP8(V_MIN3) }
      if (MODE == 6) { REP8(V_OR3) }
      if (MODE == 7) { REP8(V_MIN) }
      if (MODE == 8) { REP8(V_CMP_CND) }
      if (MODE == 9) { REP8(V_CMPS_CND) }
      if (MODE == 10) { REP8(V_RDLANE_ONLY) }
      if (MODE == 11) { REP8(S_ADD) }
      if (MODE == 12) { REP8(S_NOP) }
      if (MODE == 13) { REP8(S_WAIT) }
      if (MODE == 14) { REP8(V_ADD) REP8(S_ADD) }
      if (MODE == 15) { REP8(V_ADD) REP8(S_NOP) }
    }
  }
  uint32_t r = b + c + s;
  for (int i = 0; i < 8; ++i) r += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, int per_iter, double ghz) {
  const int bpc = 8; const uint32_t iters = 4000;
  uint32_t* out; hipMalloc(&out, 256 * bpc * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256 * bpc, 256>>>(out, 10, 1); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<256 * bpc, 256>>>(out, iters, 1); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); hipFree(out);
  const double n = (double)iters * 8 * per_iter;           // instructions per wave
  printf("%-44s %7.3f ms   %5.2f cycles per instruction per SIMD (8 waves per SIMD)\n", name, ms, ms * 1e-3 * ghz * 1e9 / (n * bpc));
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  printf("clock %.2f GHz\n", ghz);
  run<0>("v_add_u32", 8, ghz); run<1>("v_add_u32_dpp row_newbcast", 8, ghz); run<2>("v_add_u32_dpp quad_perm", 8, ghz);
  run<3>("v_add_u32 (SGPR operand)", 8, ghz); run<4>("v_add_u32_e64 clamp", 8, ghz); run<5>("v_min3_u32", 8, ghz); run<6>("v_or3_b32", 8, ghz);
  run<7>("v_min_u32", 8, ghz); run<8>("v_cmp (vcc) + v_cndmask [2 instr]", 16, ghz); run<9>("v_cmp_e64 (sgpr pair) + v_cndmask_e64 [2 instr]", 16, ghz);
  run<10>("v_readlane_b32", 8, ghz); run<11>("s_add_u32", 8, ghz); run<12>("s_nop 0", 8, ghz); run<13>("s_waitcnt vmcnt(0) (nothing pending)", 8, ghz);
  run<14>("v_add_u32 + s_add_u32 [2 instr]", 16, ghz); run<15>("v_add_u32 + s_nop [2 instr]", 16, ghz);
}
