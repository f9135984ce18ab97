// What does a barrier among workgroups of ONE XCD cost?  (measure it before building the XCD-resident
// persistent sweep for mid-size graphs.)
// Workgroups b with b % 8 == x run on XCD x (observed placement, not a promise: every participant reports HW_REG_XCC_ID
// and the host checks).  The W participants iterate `iters` times over
//     [optional payload: each thread stores one 8-byte word, then loads 12 words other workgroups stored]  ->  barrier
// in three barrier forms:
//     flags   every workgroup publishes its iteration number with a PLAIN store (stays in the XCD's L2), one wave polls the
//             W flags with sc1 loads (L1 bypassed, L2 served): all-gather, no atomics
//     atomic  one monotonic counter, device-scope atomicAdd to arrive, sc1 load to poll (what a chip-wide barrier uses)
//     launch  no barrier: the same payload as `iters` dependent launches (the form the engine uses today)
// Every spin is bounded (a workgroup that is not resident or not on the expected XCD must not hang the box).
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u; }   // HW_REG_XCC_ID
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld_sc1(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Ctl { uint32_t counter; uint32_t abort; uint32_t bad; uint32_t pad[61]; uint32_t flags[256]; uint32_t xcc[256]; uint64_t ticks[256]; };

template <int FORM, bool PAYLOAD>
__global__ __launch_bounds__(256) void k_bar(Ctl *c, uint32_t W, uint32_t iters, uint32_t xcd, uint64_t *words, uint32_t n_words) {
  if ((blockIdx.x & 7u) != xcd) return;
  const uint32_t p = blockIdx.x >> 3, tid = threadIdx.x;
  if (p >= W) return;
  if (tid == 0) c->xcc[p] = xcc_id();
  __shared__ uint32_t s_ok;
  const uint64_t t0 = wall_clock64();
  uint64_t acc = 0;
  for (uint32_t it = 1; it <= iters; ++it) {
    if (PAYLOAD) {
      const uint32_t v = p * 256u + tid;
      words[v] = ((uint64_t)it << 32) | v;                                     // plain store: the line stays in this XCD's L2
      uint64_t q[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) q[k] = ld_sc1(words + (v * 2654435761u + k * 40503u) % n_words);   // other workgroups' words
#pragma unroll
      for (int k = 0; k < 12; ++k) acc += q[k] >> 32;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 64) {                                                             // wave 0 runs the barrier
      bool ok = true;
      if (FORM == 0) {
        if (tid == 0) { c->flags[p] = it; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        uint32_t spins = 0;
        for (;;) {
          uint32_t m = 0xFFFFFFFFu;
          for (uint32_t j = tid; j < W; j += 64) m = min(m, ld_sc1(&c->flags[j]));
          const bool mine = m >= it;
          if (__all(mine)) break;
          if (++spins > 2000000u || ld_sc1(&c->abort)) { ok = false; break; }
        }
      } else {
        if (tid == 0) atomicAdd(&c->counter, 1u);
        uint32_t spins = 0;
        while (ld_sc1(&c->counter) < it * W) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000u || ld_sc1(&c->abort)) { ok = false; break; }
        }
      }
      if (tid == 0) { s_ok = ok ? 1u : 0u; if (!ok) { atomicOr(&c->abort, 1u); atomicAdd(&c->bad, 1u); } }
    }
    __syncthreads();
    if (!s_ok) break;
  }
  if (tid == 0) c->ticks[p] = wall_clock64() - t0 + (acc == 0x123456789ull ? 1 : 0);
}

// the payload as one launch per iteration (every workgroup of the grid takes part: 8 x fewer vertices per CU)
__global__ __launch_bounds__(256) void k_step(uint32_t it, uint64_t *words, uint32_t n_words, uint64_t *sink) {
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  if (v >= n_words) return;
  words[v] = ((uint64_t)it << 32) | v;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k) acc += words[(v * 2654435761u + k * 40503u) % n_words] >> 32;
  if (acc == 0x123456789ull) sink[0] = acc;
}

int main() {
  Ctl *c = nullptr; uint64_t *words = nullptr;
  CK(hipMalloc((void **)&c, sizeof(Ctl)));
  CK(hipMalloc((void **)&words, 256 * 256 * 8 + 64));
  std::vector<Ctl> h(1);
  const uint32_t iters = 200;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int payload = 0; payload < 2; ++payload)
    for (int form = 0; form < 2; ++form)
      for (uint32_t W : {8u, 16u, 40u, 64u, 128u}) {
        for (uint32_t xcd : {0u, 5u}) {
          float best = 1e9f; uint32_t bad = 0, wrong = 0; uint64_t ticks = 0;
          for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(c, 0, sizeof(Ctl)));
            CK(hipEventRecord(a));
            const dim3 grid(8 * W), block(256);
            if (form == 0 && payload == 0) hipLaunchKernelGGL((k_bar<0, false>), grid, block, 0, 0, c, W, iters, xcd, words, W * 256u);
            if (form == 0 && payload == 1) hipLaunchKernelGGL((k_bar<0, true>), grid, block, 0, 0, c, W, iters, xcd, words, W * 256u);
            if (form == 1 && payload == 0) hipLaunchKernelGGL((k_bar<1, false>), grid, block, 0, 0, c, W, iters, xcd, words, W * 256u);
            if (form == 1 && payload == 1) hipLaunchKernelGGL((k_bar<1, true>), grid, block, 0, 0, c, W, iters, xcd, words, W * 256u);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            CK(hipMemcpy(h.data(), c, sizeof(Ctl), hipMemcpyDeviceToHost));
            best = std::min(best, ms); bad += h[0].bad;
            for (uint32_t j = 0; j < W; ++j) { if (h[0].xcc[j] != xcd) ++wrong; ticks = std::max<uint64_t>(ticks, h[0].ticks[j]); }
          }
          printf("%s %s W=%3u xcd=%u: %.2f us per iteration (event), %.2f us (device wall clock, last rep)  aborted=%u  off-XCD workgroups=%u\n",
                 form == 0 ? "flags " : "atomic", payload ? "payload" : "empty  ", W, xcd, best * 1000.f / iters, ticks / 100.0 / iters, bad, wrong);
        }
      }
  for (uint32_t W : {40u, 128u}) {                                         // the same payload as dependent launches
    uint64_t *sink = words + 256 * 256;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(a));
      for (uint32_t it = 1; it <= iters; ++it) hipLaunchKernelGGL(k_step, dim3(W), dim3(256), 0, 0, it, words, W * 256u, sink);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    printf("launch payload W=%3u: %.2f us per iteration (dependent launches on one stream)\n", W, best * 1000.f / iters);
  }
  return 0;
}
