// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the engine's kernels use
// (the x2 FETCH_SIZE correction of the guide is quoted for 16-B/lane streaming reads; k_fused_lean
// issues 4-B/lane buffer loads = one 256-byte row per wave instruction, and gathers such rows from scattered places).
// Each kernel moves a KNOWN number of bytes from a buffer far larger than the 256 MB Infinity Cache:
//     calib_read4      every lane one dword, a wave instruction reads 256 consecutive bytes, rows streamed in order
//     calib_read16     every lane one uint4, a wave instruction reads 1 024 consecutive bytes
//     calib_gather256  every wave instruction reads one 256-byte row at a pseudo-random row of the buffer (each row once)
//     calib_read4_mall calib_read4 over the same 64 MB again and again (Infinity Cache hits: is the counter HBM or fabric?)
//     calib_write4     every lane stores one dword (256-byte rows, streamed)
// run under  rocprofv3 --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (tools/gpu_profile.sh, STAGE 2); the factor to apply to
// the counter is  bytes_moved / (counter_KiB * 1024).
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib && /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void calib_read4(const uint32_t *__restrict__ src, size_t n_words, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
    acc ^= src[i];
  if (acc == 0x9E3779B9u) sink[0] = acc;                  // (never: the loads must not be optimised away)
}

// the same loop over a 64 MB piece read again and again: larger than the eight L2s together (32 MB), a quarter of the
// Infinity Cache — if FETCH_SIZE counts these bytes in full, the counter sits on the fabric side of the L2, not at HBM
__global__ void calib_read4_mall(const uint32_t *__restrict__ src, size_t n_words, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
    acc ^= src[i];
  if (acc == 0x9E3779B9u) sink[0] = acc;
}

__global__ void calib_read16(const uint4 *__restrict__ src, size_t n_quads, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 q = src[i];
    acc ^= q.x ^ q.y ^ q.z ^ q.w;
  }
  if (acc == 0x9E3779B9u) sink[0] = acc;
}

// rows of 64 dwords; wave w of the grid reads rows perm(w), perm(w + waves), ...; perm = multiplication by an odd
// constant modulo the (power of two) row count: a bijection, neighbouring waves land megabytes apart
__global__ void calib_gather256(const uint32_t *__restrict__ src, uint32_t n_rows_log2, uint32_t *sink) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t waves = gridDim.x * (blockDim.x >> 6);
  const uint32_t n_rows = 1u << n_rows_log2;
  uint32_t acc = 0;
  for (uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < n_rows; r += waves) {
    const uint32_t row = (r * 2654435761u) & (n_rows - 1u);
    acc ^= src[(size_t)row * 64u + lane];
  }
  if (acc == 0x9E3779B9u) sink[0] = acc;
}

__global__ void calib_write4(uint32_t *__restrict__ dst, size_t n_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (uint32_t)i;
}

int main() {
  const uint32_t rows_log2 = 24;                                   // 2^24 rows x 256 B = 4 GiB
  const size_t bytes = ((size_t)1 << rows_log2) * 256u;
  uint32_t *buf = nullptr, *sink = nullptr;
  CK(hipMalloc((void **)&buf, bytes));
  CK(hipMalloc((void **)&sink, 64));
  CK(hipMemset(buf, 1, bytes));
  CK(hipDeviceSynchronize());
  const dim3 grid(256 * 16), block(256);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a)); hipLaunchKernelGGL(calib_read4, grid, block, 0, 0, buf, bytes / 4, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("calib_read4      %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
    CK(hipEventRecord(a)); hipLaunchKernelGGL(calib_read16, grid, block, 0, 0, (const uint4 *)buf, bytes / 16, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("calib_read16     %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
    CK(hipEventRecord(a)); hipLaunchKernelGGL(calib_gather256, grid, block, 0, 0, buf, rows_log2, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("calib_gather256  %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
    CK(hipEventRecord(a)); hipLaunchKernelGGL(calib_write4, grid, block, 0, 0, buf, bytes / 4); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("calib_write4     %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
  }
  const size_t small = (size_t)64 << 20;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(a)); hipLaunchKernelGGL(calib_read4_mall, grid, block, 0, 0, buf, small / 4, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("calib_read4_mall %zu bytes  %.3f ms  %.0f GB/s\n", small, ms, small / ms / 1e6);
  }
  printf("bytes_per_launch %zu (calib_read4_mall: %zu)\n", bytes, small);
  CK(hipFree(buf)); CK(hipFree(sink));
  return 0;
}
