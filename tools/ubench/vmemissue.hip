// Micro-benchmark: what does ONE vector memory instruction cost a CU when it hits in L1?  8 waves per SIMD, each wave issues
// groups of 16 loads of 64 lanes x W bytes from a small per-wave region (L1 resident), then one s_waitcnt.
// Prints cycles per instruction per CU: the floor the address unit (TA) / L1 set for a sweep that issues N such loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int W, int LANES, int SCAT>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ buf, uint32_t* out, uint32_t iters, uint32_t rows) {
  const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, (short)0, (int)(rows * 1024u), 0x00020000);
  uint32_t acc = 0;
  const uint32_t base = (wave % (rows / 64)) * 64;     // 16 rows per wave, reused every iteration: L1 hits
  // SCAT: the four 16-lane groups read four DIFFERENT rows (1 KiB apart x 5): the gather shape of k_fused_lean
  const uint32_t voff = SCAT ? ((lane >> 4) * 5u * 1024u + (lane & 15u) * W) : (LANES == 64 ? lane : (lane & (LANES - 1))) * W;
  for (uint32_t it = 0; it < iters; ++it) {
    if (W == 4) {
      uint32_t v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (base + ((j + it) & 15u)) * 1024u, 0);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += v[j];
    } else {
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (base + ((j + it) & 15u)) * 1024u, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].w;
    }
  }
  out[wave * 64 + lane] = acc;
}
template <int W, int LANES, int SCAT> void run(const char* name, int per_iter, double ghz) {
  const int bpc = 8; const uint32_t iters = 2000, rows = 1 << 14;
  uint32_t *buf, *out; hipMalloc(&buf, (size_t)rows * 1024); hipMemset(buf, 0, (size_t)rows * 1024); hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<W, LANES, SCAT><<<256 * bpc, 256>>>(buf, out, 10, rows); hipDeviceSynchronize();
  hipEventRecord(a); k<W, LANES, SCAT><<<256 * bpc, 256>>>(buf, out, iters, rows); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); hipFree(buf); hipFree(out);
  const double n = (double)iters * per_iter * 4 * bpc;      // instructions per CU
  printf("%-52s %7.3f ms   %5.2f cycles per instruction per CU\n", name, ms, ms * 1e-3 * ghz * 1e9 / n);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  run<4, 64, 0>("buffer_load_dword, 64 lanes (a 256-byte row)", 16, ghz);
  run<4, 16, 0>("buffer_load_dword, 16 distinct dwords x 4 (ELL record)", 16, ghz);
  run<16, 64, 0>("buffer_load_dwordx4, 64 lanes (1 KiB contiguous)", 8, ghz);
  run<16, 16, 0>("buffer_load_dwordx4, 16 distinct x 4", 8, ghz);
  run<16, 64, 1>("buffer_load_dwordx4, 4 groups x 256 B, 4 rows apart", 8, ghz);
  run<4, 64, 1>("buffer_load_dword, 4 groups x 64 B, 4 rows apart", 16, ghz);
}
