// Where does kb_pa_shift (graph_patch.hip.h) spend its time?  The real kernel and stripped variants on a synthetic 1 M-link layout
// with 21 affected rows.   hipcc --offload-arch=gfx950 -O3 -I holo_amd/csrc -I include tools/ubench/pa_shift.hip -o /tmp/pa_shift
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "holo_spf_hip.h"
#include "graph_patch.hip.h"
using namespace hspf;

__global__ void __launch_bounds__(256) v_copy6(uint32_t kept, const uint32_t *a0, const uint32_t *a1, const uint32_t *a2, const uint32_t *a3, const uint32_t *a4, const uint32_t *a5,
                                               uint32_t *b0, uint32_t *b1, uint32_t *b2, uint32_t *b3, uint32_t *b4, uint32_t *b5) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kept) return;
  b0[k] = a0[k]; b1[k] = a1[k]; b2[k] = a2[k]; b3[k] = a3[k]; b4[k] = a4[k]; b5[k] = a5[k];
}
__global__ void __launch_bounds__(256) v_copy6x4(uint32_t kept4, const uint4 *a0, const uint4 *a1, const uint4 *a2, const uint4 *a3, const uint4 *a4, const uint4 *a5,
                                                 uint4 *b0, uint4 *b1, uint4 *b2, uint4 *b3, uint4 *b4, uint4 *b5) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kept4) return;
  b0[k] = a0[k]; b1[k] = a1[k]; b2[k] = a2[k]; b3[k] = a3[k]; b4[k] = a4[k]; b5[k] = a5[k];
}
__global__ void __launch_bounds__(256) v_copy1(uint32_t kept, const uint32_t *a0, uint32_t *b0) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kept) return;
  b0[k] = a0[k];
}

int main() {
  const uint32_t n = 100000, kept = 1000000, na = 21, na1 = na + 1;
  const size_t lb = (size_t)(kept + kept / 8 + 16) * 4;
  uint32_t *A[6], *B[6];
  for (int i = 0; i < 6; ++i) { hipMalloc(&A[i], lb); hipMalloc(&B[i], lb); hipMemset(A[i], 0, lb); hipMemset(B[i], 0, lb); }
  std::vector<uint32_t> aff(na), meta(PA_META * na1, 0);
  for (uint32_t j = 0; j < na; ++j) {
    aff[j] = 40000 + j * 37;
    meta[j] = aff[j] * 10; meta[na1 + j] = 10; meta[2 * na1 + j] = aff[j] * 10; meta[3 * na1 + j] = 10;
    meta[4 * na1 + j] = 10; meta[5 * na1 + j] = 10;
  }
  meta[4 * na1 + 3] = 9; meta[5 * na1 + 3] = 9;
  for (uint32_t j = 0; j <= na; ++j) { meta[6 * na1 + j] = j > 3 ? 0xFFFFFFFFu : 0u; meta[7 * na1 + j] = j > 3 ? 0xFFFFFFFFu : 0u; }
  uint32_t *d_aff, *d_meta, *st_in, *st_out, *in_ptr, *out_ptr; uint8_t *leaf; BuildInfo *info;
  hipMalloc(&d_aff, na * 4); hipMalloc(&d_meta, meta.size() * 4); hipMalloc(&st_in, na * 3 * PA_IN_STRIDE * 4); hipMalloc(&st_out, na * 3 * PA_OUT_STRIDE * 4);
  hipMalloc(&in_ptr, (n + 17) * 4); hipMalloc(&out_ptr, (n + 17) * 4); hipMalloc(&leaf, n); hipMalloc(&info, 32768);
  hipMemset(st_in, 0, na * 3 * PA_IN_STRIDE * 4); hipMemset(st_out, 0, na * 3 * PA_OUT_STRIDE * 4); hipMemset(in_ptr, 0, (n + 17) * 4); hipMemset(out_ptr, 0, (n + 17) * 4);
  hipMemset(leaf, 0, n); hipMemset(info, 0, 32768);
  hipMemcpy(d_aff, aff.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice);
  BuildInfo bi{}; bi.kept = kept - 1;
  hipMemcpy(info, &bi, sizeof bi, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char *name, auto fn) {
    for (int w = 0; w < 3; ++w) fn();
    hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    for (int r = 0; r < 10; ++r) {
      hipMemcpy(info, &bi, sizeof bi, hipMemcpyHostToDevice);
      hipEventRecord(e0, 0); fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    printf("%-28s best %.1f us  mean %.1f us\n", name, best * 1e3, sum * 100);
  };
  const dim3 g((kept + 255) / 256), b(256);
  timeit("kb_pa_shift A->B", [&] { hipLaunchKernelGGL(kb_pa_shift, g, b, 0, 0, n, na, d_aff, d_meta, st_in, st_out, A[0], A[1], A[2], A[3], A[4], A[5], B[0], B[1], B[2], B[3], B[4], B[5], in_ptr, out_ptr, leaf, info); });
  timeit("copy of 6 arrays, 4 B/lane", [&] { hipLaunchKernelGGL(v_copy6, g, b, 0, 0, kept, A[0], A[1], A[2], A[3], A[4], A[5], B[0], B[1], B[2], B[3], B[4], B[5]); });
  timeit("copy of 6 arrays, 16 B/lane", [&] { hipLaunchKernelGGL(v_copy6x4, dim3((kept / 4 + 255) / 256), b, 0, 0, kept / 4, (uint4 *)A[0], (uint4 *)A[1], (uint4 *)A[2], (uint4 *)A[3], (uint4 *)A[4], (uint4 *)A[5], (uint4 *)B[0], (uint4 *)B[1], (uint4 *)B[2], (uint4 *)B[3], (uint4 *)B[4], (uint4 *)B[5]); });
  timeit("copy of 1 array, 4 B/lane", [&] { hipLaunchKernelGGL(v_copy1, g, b, 0, 0, kept, A[0], B[0]); });
  timeit("6 x copy of 1 array", [&] { for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(v_copy1, g, b, 0, 0, kept, A[i], B[i]); });
  return 0;
}
