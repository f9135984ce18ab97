// Micro-benchmark: how fast can a CU pull random coalesced rows (64 lanes x W bytes) out of L2 / MALL?
// usage: rowload <buffer MB> <bytes per lane: 4|8|16> <loads in flight per wave> <blocks per CU>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
template <typename T> __device__ inline uint32_t first_word(const T& v) { return (uint32_t)v; }
template <> __device__ inline uint32_t first_word<uint4>(const uint4& v) { return v.x; }
template <typename T, int K>
__global__ __launch_bounds__(256) void k(const T* __restrict__ buf, const uint32_t* __restrict__ idx, uint32_t nidx, uint32_t iters, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  uint32_t acc = 0;
  uint32_t p = (wave * 977u) % nidx;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t iv = idx[(p + lane) % nidx];      // 64 row indices per wave per iteration (coalesced)
    p = (p + 64) % nidx;
#pragma unroll
    for (int g = 0; g < 64 / K; ++g) {
      T v[K];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const uint32_t r = __builtin_amdgcn_readlane(iv, g * K + j);
        v[j] = buf[(size_t)r * 64 + lane];
      }
#pragma unroll
      for (int j = 0; j < K; ++j) acc += first_word(v[j]);
    }
  }
  if (acc == 0x12345678) out[0] = acc;
}
template <typename T, int K> double run(size_t mb, int bpc, int iters) {
  const size_t rows = mb * 1024 * 1024 / (64 * sizeof(T));
  T* buf; hipMalloc(&buf, rows * 64 * sizeof(T)); hipMemset(buf, 1, rows * 64 * sizeof(T));
  std::vector<uint32_t> h(1 << 20); uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % rows); }
  uint32_t *idx, *out; hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMalloc(&out, 4);
  const int blocks = 256 * bpc;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<T, K><<<blocks, 256>>>(buf, idx, (uint32_t)h.size(), 2, out);
  hipDeviceSynchronize();
  hipEventRecord(a); k<T, K><<<blocks, 256>>>(buf, idx, (uint32_t)h.size(), iters, out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * 4 * iters * 64 * 64 * sizeof(T);
  hipFree(buf); hipFree(idx); hipFree(out);
  return bytes / ms / 1e9;   // TB/s... bytes/ms/1e9 = TB/s
}
int main(int argc, char** argv) {
  for (size_t mb : {2, 13, 26, 51, 200}) for (int bpc : {4, 8}) {
    printf("buf %4zu MB  blocks/CU %d :  2B/lane K8 %6.2f K16 %6.2f | 4B/lane K4 %6.2f  K8 %6.2f K16 %6.2f | 8B/lane K4 %6.2f K8 %6.2f | 16B/lane K4 %6.2f K8 %6.2f  TB/s\n", mb, bpc,
      run<uint16_t,8>(mb,bpc,40), run<uint16_t,16>(mb,bpc,40),
      run<uint32_t,4>(mb,bpc,40), run<uint32_t,8>(mb,bpc,40), run<uint32_t,16>(mb,bpc,40), run<uint64_t,4>(mb,bpc,40), run<uint64_t,8>(mb,bpc,40), run<uint4,4>(mb,bpc,20), run<uint4,8>(mb,bpc,20));
  }
}
