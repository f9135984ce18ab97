// Micro-benchmark: a workgroup (4 waves) gathers R random 256-byte rows (64 lanes x 4 B) into LDS, then reads them back —
// the shape of k_fused_stg's staging.  Modes: 0 = LDS-DMA dword (one row per instruction), 1 = VGPR loads + ds_write_b32,
// 2 = LDS-DMA dwordx4 (1 KiB = 4 consecutive rows per instruction), 3 = VGPR loads only (no LDS, the k_fused shape).
// usage: ldsgather     (prints a table: buffer MB x rows per block x mode -> TB/s gathered, us per 6250 workgroups)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
extern __shared__ uint32_t lds[];
template <int MODE, int RPW>      // RPW rows per wave (R = 4 RPW)
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ buf, uint32_t rows, const uint32_t* __restrict__ idx, uint32_t nidx,
                                         uint32_t iters, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane4 = lane * 4;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, (short)0, (int)(rows * 256u), 0x00020000);
  uint32_t acc = 0;
  uint32_t p = ((blockIdx.x * 4 + wave) * 977u) % nidx;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t iv = idx[(p + lane) % nidx];
    p = (p + 64 * 131) % nidx;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < RPW; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * RPW + j) * 64), 4, lane4,
                                                 __builtin_amdgcn_readlane(iv, j) << 8, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < RPW / 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * RPW + j * 4) * 64), 16, lane * 16,
                                                 (__builtin_amdgcn_readlane(iv, j) & ~3u) << 8, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
      uint32_t v[RPW];
#pragma unroll
      for (int j = 0; j < RPW; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4, __builtin_amdgcn_readlane(iv, j) << 8, 0);
#pragma unroll
      for (int j = 0; j < RPW; ++j) lds[(wave * RPW + j) * 64 + lane] = v[j];
    } else {
      uint32_t v[RPW];
#pragma unroll
      for (int j = 0; j < RPW; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4, __builtin_amdgcn_readlane(iv, j) << 8, 0);
#pragma unroll
      for (int j = 0; j < RPW; ++j) acc += v[j];
    }
    if (MODE != 3) {
      __syncthreads();
      // every wave reads 40 rows of the table back (10 links x 4 rows)
#pragma unroll 8
      for (int j = 0; j < 40; ++j) acc += lds[(((wave * 7 + j * 5) % (4 * RPW))) * 64 + lane];
      __syncthreads();
    }
  }
  if (acc == 0x12345678) out[0] = acc;
}
template <int MODE, int RPW> void run(size_t mb, int wgs, int iters, bool local) {
  const size_t rows = mb * 1024 * 1024 / 256;
  uint32_t* buf; hipMalloc(&buf, rows * 256); hipMemset(buf, 1, rows * 256);
  std::vector<uint32_t> h(1 << 20); uint64_t s = 88172645463325252ull;
  uint32_t base = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    if (local) { if ((i & 63) == 0) base = (uint32_t)(s % rows); h[i] = (uint32_t)((base + (s >> 40) % 700) % rows); }   // a workgroup's rows within ~700 rows
    else h[i] = (uint32_t)(s % rows);
  }
  uint32_t *idx, *out; hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMalloc(&out, 4);
  const size_t ldsb = (size_t)4 * RPW * 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, RPW><<<wgs, 256, ldsb>>>(buf, (uint32_t)rows, idx, (uint32_t)h.size(), 1, out);
  hipDeviceSynchronize();
  hipEventRecord(a); k<MODE, RPW><<<wgs, 256, ldsb>>>(buf, (uint32_t)rows, idx, (uint32_t)h.size(), iters, out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * iters * 4 * RPW * 256;
  printf("buf %3zu MB %s mode %d rows/wg %3d wgs %5d iters %d : %7.1f us  %6.2f TB/s  -> %6.1f us per 6250 workgroup-passes\n", mb, local ? "local " : "random", MODE,
         4 * RPW, wgs, iters, ms * 1e3, bytes / ms / 1e9, ms * 1e3 * 6250.0 / ((double)wgs * iters));
  hipFree(buf); hipFree(idx); hipFree(out);
}
int main() {
  for (int local = 0; local < 2; ++local)
    for (size_t mb : {2, 26}) {
      run<0, 24>(mb, 6250, 1, local); run<0, 24>(mb, 6250, 4, local); run<0, 24>(mb, 1280, 20, local);
      run<1, 24>(mb, 6250, 1, local); run<1, 24>(mb, 6250, 4, local); run<1, 24>(mb, 1280, 20, local);
      run<2, 24>(mb, 6250, 1, local); run<2, 24>(mb, 6250, 4, local);
      run<3, 24>(mb, 6250, 1, local); run<3, 24>(mb, 6250, 4, local); run<3, 12>(mb, 6250, 8, local);
      run<0, 16>(mb, 6250, 4, local); run<1, 16>(mb, 6250, 4, local);
    }
}
