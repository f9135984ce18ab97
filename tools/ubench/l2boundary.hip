// Micro-benchmark: what does a kernel boundary leave in the XCDs' L2s?
// Kernel R: every workgroup b reads ITS slice of a buffer (slice = b, so the same XCD touches the same bytes in every
// launch; total 16 MB = 2 MB per XCD) with dependent loads (latency-bound: one wave per workgroup walks its slice with a
// stride), and reports cycles per load.  Sequences: R after R (clean lines), R after W (lines written by the same XCD in
// the previous launch), R after a 1 GB flush kernel (cold).  An L2 hit is ~200-300 cycles, MALL / HBM 700-1000+.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void kR(const uint32_t* __restrict__ buf, uint32_t slice_words, uint32_t* out, uint64_t* cyc) {
  const uint32_t b = blockIdx.x;
  const uint32_t* p = buf + (size_t)b * slice_words;
  uint32_t idx = threadIdx.x;            // 64 lanes: a 256-byte row per step
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (uint32_t i = 0; i < 64; ++i) {     // 64 dependent row loads, stride 5 rows inside the slice
    const uint32_t v = p[(idx + acc * 0) % slice_words];
    acc += v;                             // data = 0: the dependence is real, the address unchanged
    idx = (idx + 5 * 64 + (acc & 0)) % slice_words;
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { cyc[b] = t1 - t0; }
  if (acc == 0x12345) out[0] = acc;
}
__global__ void kW(uint32_t* buf, uint32_t slice_words) {
  uint32_t* p = buf + (size_t)blockIdx.x * slice_words;
  for (uint32_t i = threadIdx.x; i < slice_words; i += blockDim.x) p[i] = 0;
}
__global__ void kFlush(const uint4* big, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += big[i].x;
  if (acc == 0x12345) out[0] = acc;
}
int main() {
  const uint32_t blocks = 2048, slice_words = 2048;      // 8 KB per workgroup, 16 MB total, 2 MB per XCD
  uint32_t *buf, *out; uint64_t* cyc; uint4* big;
  const size_t bign = (size_t)1 << 26;                    // 1 GB
  hipMalloc(&buf, (size_t)blocks * slice_words * 4); hipMemset(buf, 0, (size_t)blocks * slice_words * 4);
  hipMalloc(&out, 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&big, bign * 16); hipMemset(big, 0, bign * 16);
  std::vector<uint64_t> h(blocks);
  auto report = [&](const char* what) {
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    printf("%-44s %8.1f cycles per dependent row load (mean over %u workgroups)\n", what, s / blocks / 64.0, blocks);
  };
  for (int rep = 0; rep < 2; ++rep) {
    kFlush<<<4096, 256>>>(big, bign, out); kR<<<blocks, 64>>>(buf, slice_words, out, cyc); hipDeviceSynchronize(); report("R after a 1 GB flush (cold)");
    kR<<<blocks, 64>>>(buf, slice_words, out, cyc); hipDeviceSynchronize(); report("R after R (clean lines, previous launch)");
    kR<<<blocks, 64>>>(buf, slice_words, out, cyc); hipDeviceSynchronize(); report("R after R after R");
    kW<<<blocks, 64>>>(buf, slice_words); kR<<<blocks, 64>>>(buf, slice_words, out, cyc); hipDeviceSynchronize(); report("R after W (same workgroup -> same XCD wrote it)");
    kW<<<blocks - 7, 64>>>(buf + 7 * slice_words, slice_words); kR<<<blocks, 64>>>(buf, slice_words, out, cyc); hipDeviceSynchronize(); report("R after W shifted by 7 slices (another XCD wrote it)");
  }
  // within one launch: second pass over the same slice
  return 0;
}
