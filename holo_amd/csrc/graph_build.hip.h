// graph_build.hip.h — gfx950 kernels that turn the caller's CSR (one row per LSA / LSP vertex, links in
// LSA order) into the layout the SPF kernels read, entirely in HBM.
//
// This is the step immediately before the path: the reference re-derives the graph from
// TLVs / LSAs on every link visit and repeats the two-way check per visit
// (holo-ospf/src/spf.rs:654-664, holo-isis/src/spf.rs:616-627); here it is done once per LSDB generation
// (hspf_graph_upload) or once per changed set of rows (hspf_graph_patch), by streaming passes over the links:
//
//   kb_links     one thread per link: owning row (binary search in row_ptr), range checks, two-way check
//                (row of the target lists the source), kept = two-way and the source can be expanded;
//                in-degree histogram.
//   scan         exclusive prefix sums: kept links -> positions of the forward (out) arrays and out_ptr;
//                in-degrees -> in_ptr.
//   kb_scatter   kept links into the forward arrays (stable: global link order) and, through a per-row
//                atomic cursor, into unsorted in-rows.
//   kb_rank      each in-link counts the in-links of its row that precede it in
//                (cost descending, source ascending, position ascending) — a total order, so the final
//                layout does not depend on the order the atomics happened to resolve in.
//   kb_rowflags  per-row static flags for the fused sweep; hop-count shape of the graph; pads.
//   kb_patch_raw (patch, graph_patch.hip.h) new raw CSR = old rows, except the replaced ones taken from the delta.
//
// Hub mode (some row lists more than HUB_DEG links — a LAN pseudonode with thousands of members): the per-link row scans
// of kb_links and kb_rank would be quadratic in such a row, so the same layout is derived from two stable device-wide
// radix sorts (hub_sort.h) instead:
//   kb_hub_keys     key (source, target) per link -> sorted: every row's targets ascending, rows in place
//   kb_hub_links    kb_links with the two-way check as a binary search in the target's sorted row
//   kb_hub_scatter  forward arrays as kb_scatter; key (target, ~cost) per kept link in forward order (= source, position
//                   ascending), dropped links keyed behind all kept ones -> stable sort = the in-row order of kb_rank
//   kb_hub_in_ptr   in-row bounds = first sorted key of every target (no in-degree histogram: no atomics on a hub's counter)
//   kb_hub_gather   in-link arrays from the sorted keys and the permutation
// Every per-link step is then O(log degree).  The layouts of the two modes are identical array by array.
//
// Everything is integer streaming work bound by HBM / L2 bandwidth; no MFMA, no LDS tiling beyond the scans.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spf_kernels.hip.h"

namespace hspf {

constexpr uint32_t GB_ERR_COL = 1u;      // a link targets a vertex >= n_vertices
constexpr uint32_t GB_ERR_METRIC = 2u;   // a link carries the reserved cost 0xFFFFFFFF

struct BuildInfo {       // device-resident summary of one build, copied back once
  uint32_t err;          // GB_ERR_*
  uint32_t kept;         // links that survive (two-way, source expandable)
  uint32_t wmax;         // largest kept cost
  uint32_t hc_bad;       // some kept link breaks the hop-count shape
  uint32_t hc_net;       // some network row has a kept in-link
  uint32_t xcd_start[9]; // GraphDev::xcd_start: work-balanced chunk ranges of the 8 XCDs (no heavy chunk), else even
                         // shares of the normal units
  uint32_t n_heavy;      // heavy 16-vertex chunks (a row of more than UNIT_HEAVY_DEG in-links): 4 work units each
  uint32_t max_in_deg;   // largest kept in-degree
  uint32_t any_rowflags; // OR of the rows' static flags (RF_*)
  uint32_t n_zero_rows;  // rows with RF_ZERO
  uint32_t n_bad_rows;   // rows with an in-link off the hop-count shape (hc_bad != 0 exactly when this is)
  uint32_t n_leaf;       // leaves (GraphDev::leaf)
};

// Counts that every row may add to (rows off the hop-count shape, RF_ZERO rows, leaves) do not go to BuildInfo directly: an
// atomic on ONE address costs ~10 ns whoever issues it (15 600 of them made kb_scatter 180 us, 100 000 made kb_rowflags
// 40-130 us), so each count has 64 counters in 64 different 128-byte lines behind BuildInfo, a block adds to counter
// blockIdx & 63, and the single-workgroup kernel at the end of the build (kb_units_small / kb_xcd) sums them into BuildInfo.
// The largest kept cost is spread the same way (GB_SC_WMAX: 64 maxima, folded into BuildInfo::wmax at the end): a streaming kernel
// starts with ~8 000 waves that all read "0 so far" and all raise ONE word — kb_pa_shift took 60-100 us in a patch's chain and
// 14.5 us on a layout of zero costs, where no wave had anything to raise (profiles/r06_notes.md r06zb).
constexpr uint32_t GB_SC_BAD = 0, GB_SC_ZERO = 1, GB_SC_LEAF = 2, GB_SC_COUNTS = 3, GB_SC_WMAX = 3, GB_SC_CLASSES = 4;
constexpr uint32_t GB_SC_WORDS = GB_SC_CLASSES * 64u * 32u;
__device__ __forceinline__ uint32_t *gb_spread(BuildInfo *info, uint32_t which) {
  return (uint32_t *)info + 32u + (which * 64u + (blockIdx.x & 63u)) * 32u;
}
// one wave (lanes 0..63 of `lane`): BuildInfo's counts from the spread counters
__device__ __forceinline__ void gb_counts_finish(BuildInfo *info, uint32_t lane) {
  uint32_t v[GB_SC_COUNTS];
  for (uint32_t c = 0; c < GB_SC_COUNTS; ++c) {
    v[c] = ((const uint32_t *)info)[32u + (c * 64u + lane) * 32u];
    for (int o = 32; o; o >>= 1) v[c] += (uint32_t)__shfl_xor((int)v[c], o);
  }
  uint32_t wm = ((const uint32_t *)info)[32u + (GB_SC_WMAX * 64u + lane) * 32u];
  for (int o = 32; o; o >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, o));
  if (lane == 0u) {
    info->n_bad_rows = v[GB_SC_BAD]; info->hc_bad = v[GB_SC_BAD] ? 1u : 0u;
    info->n_zero_rows = v[GB_SC_ZERO];
    info->n_leaf = v[GB_SC_LEAF];
    info->wmax = max(info->wmax, wm);                       // (hub mode: kb_hub_scatter raises BuildInfo::wmax itself, once per block)
  }
}
// a wave's largest cost into the spread maxima (lane 0 of the wave; a stale read costs one atomic too many, never a wrong maximum)
__device__ __forceinline__ void gb_raise_wmax(BuildInfo *info, uint32_t wmax) {
  uint32_t *slot = gb_spread(info, GB_SC_WMAX);
  if (wmax > *(volatile uint32_t *)slot) atomicMax(slot, wmax);
}

// rowaux bits (hspf_graph::d_rowaux): the per-row facts behind BuildInfo's hop-count summary, kept so that a structural
// patch can re-derive the summary from per-row arrays after rewriting only the affected rows (graph_patch.hip.h)
constexpr uint32_t RA_BAD = 1u;          // some kept in-link of the row is off the hop-count shape
constexpr uint32_t RA_NET_IN = 2u;       // a network row with a kept in-link

constexpr uint32_t HUB_DEG = 512;        // rows with more links than this: hub mode (HSPF_HUB_DEG)

constexpr int GB_BLOCK = 256;
constexpr int GB_ITEMS = 8;                       // scan: items per thread
constexpr int GB_TILE = GB_BLOCK * GB_ITEMS;      // scan: items per block

// Row that owns link k: the largest u < n with row_ptr[u] <= k (empty rows are skipped naturally).
__device__ __forceinline__ uint32_t gb_row_of(const uint32_t *__restrict__ row_ptr, uint32_t n, uint32_t k) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (row_ptr[mid] <= k) lo = mid; else hi = mid;
  }
  return lo;
}

// in_cnt[0 .. n] and BuildInfo + its spread counters, zeroed by ONE launch (two hipMemsetAsync of these sizes are six fill
// kernels of the runtime: aligned part + tails)
__global__ void __launch_bounds__(GB_BLOCK)
kb_clear(uint32_t *__restrict__ in_cnt, uint32_t n_cnt, uint32_t *__restrict__ info_words, uint32_t n_info) {
  const uint32_t i = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (i < n_cnt) in_cnt[i] = 0u;
  if (i < n_info) info_words[i] = 0u;
}

__global__ void __launch_bounds__(GB_BLOCK)
kb_links(uint32_t n, uint32_t e, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
         const uint32_t *__restrict__ metric, const uint8_t *__restrict__ vflags, uint32_t *__restrict__ src_of,
         uint8_t *__restrict__ twoway, uint8_t *__restrict__ keep, uint32_t *__restrict__ in_cnt,
         uint32_t *__restrict__ lslot, BuildInfo *__restrict__ info) {
  // The owning row: the block's 256 consecutive links span the rows [u0, u1] — two searches over all rows per BLOCK, the
  // bounds of those rows in LDS, a search over at most 257 of them per link there (a search over all rows per link was 17
  // dependent global loads).  Blocks whose links span more rows than fit (runs of empty rows) search globally as before.
  __shared__ uint32_t rp[GB_BLOCK + 2];
  __shared__ uint32_t s_u0, s_cnt;
  const uint32_t k0 = blockIdx.x * GB_BLOCK;
  if (threadIdx.x == 0) {
    const uint32_t u0 = gb_row_of(row_ptr, n, k0), u1 = gb_row_of(row_ptr, n, min(k0 + GB_BLOCK - 1u, e - 1u));
    s_u0 = u0; s_cnt = u1 - u0 + 1u;
  }
  __syncthreads();
  const uint32_t u0 = s_u0, cnt = s_cnt;
  const bool in_lds = cnt <= (uint32_t)GB_BLOCK + 1u;
  if (in_lds)
    for (uint32_t i = threadIdx.x; i < cnt; i += GB_BLOCK) rp[i] = row_ptr[u0 + i];
  __syncthreads();
  const uint32_t k = k0 + threadIdx.x;
  if (k >= e) return;
  uint32_t u;
  if (in_lds) {
    uint32_t lo = 0, hi = cnt;                          // largest i with rp[i] <= k (rp[0] <= k0 <= k)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (rp[mid] <= k) lo = mid; else hi = mid;
    }
    u = u0 + lo;
  } else {
    u = gb_row_of(row_ptr, n, k);
  }
  const uint32_t t = col[k];
  src_of[k] = u;
  uint32_t bad = 0;
  if (t >= n) bad |= GB_ERR_COL;
  if (metric[k] == INF) bad |= GB_ERR_METRIC;
  if (bad) {
    atomicOr(&info->err, bad);
    twoway[k] = 0; keep[k] = 0;
    return;
  }
  // two-way connectivity: the target's row lists the source, cost not compared
  bool two = false;
  const uint32_t b = row_ptr[t + 1];
  for (uint32_t k2 = row_ptr[t]; k2 < b && !two; k2 += 4u) {     // four entries per step: the loop waits for its loads once per four
    const uint32_t c0 = col[k2], c1 = k2 + 1u < b ? col[k2 + 1u] : INF, c2 = k2 + 2u < b ? col[k2 + 2u] : INF,
                   c3 = k2 + 3u < b ? col[k2 + 3u] : INF;          // (u < n <= 2^24: INF never matches)
    two = c0 == u || c1 == u || c2 == u || c3 == u;
  }
  const bool kp = two && !(vflags[u] & HSPF_VF_NO_EXPAND);
  twoway[k] = two ? 1 : 0;
  keep[k] = kp ? 1 : 0;
  if (kp) lslot[k] = atomicAdd(&in_cnt[t], 1u);      // the link's place in the target's in-row (any order: kb_rank fixes the
                                                     // final one); kb_scatter took it with a second atomic per link before
}

// ---- exclusive scan of m items (u8 or u32) into out[0..m], out[m] = total ----------------------------

__device__ __forceinline__ uint32_t gb_block_exclusive(uint32_t v, uint32_t *sh, uint32_t &total) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int d = 1; d < GB_BLOCK; d <<= 1) {
    const uint32_t add = tid >= d ? sh[tid - d] : 0u;
    __syncthreads();
    sh[tid] += add;
    __syncthreads();
  }
  total = sh[GB_BLOCK - 1];
  return sh[tid] - v;
}

template <typename T>
__global__ void __launch_bounds__(GB_BLOCK)
kb_scan_sums(const T *__restrict__ in, uint32_t m, uint32_t *__restrict__ sums) {
  __shared__ uint32_t sh[GB_BLOCK];
  const uint32_t base = blockIdx.x * GB_TILE + threadIdx.x * GB_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < GB_ITEMS; ++i)
    if (base + i < m) s += (uint32_t)in[base + i];
  uint32_t total;
  (void)gb_block_exclusive(s, sh, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one block: sums[0..nb) -> exclusive prefix in place, grand total -> *total (and info->kept when asked)
__global__ void __launch_bounds__(GB_BLOCK)
kb_scan_mid(uint32_t *__restrict__ sums, uint32_t nb, uint32_t *__restrict__ total_out) {
  __shared__ uint32_t sh[GB_BLOCK];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += GB_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? sums[i] : 0u;
    uint32_t total;
    const uint32_t ex = gb_block_exclusive(v, sh, total);
    if (i < nb) sums[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename T>
__global__ void __launch_bounds__(GB_BLOCK)
kb_scan_apply(const T *__restrict__ in, uint32_t m, const uint32_t *__restrict__ sums, uint32_t *__restrict__ out) {
  __shared__ uint32_t sh[GB_BLOCK];
  const uint32_t base = blockIdx.x * GB_TILE + threadIdx.x * GB_ITEMS;
  uint32_t v[GB_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < GB_ITEMS; ++i) {
    v[i] = base + i < m ? (uint32_t)in[base + i] : 0u;
    s += v[i];
  }
  uint32_t total;
  uint32_t run = sums[blockIdx.x] + gb_block_exclusive(s, sh, total);
#pragma unroll
  for (int i = 0; i < GB_ITEMS; ++i) {
    if (base + i <= m) out[base + i] = run;      // position m receives the grand total
    run += v[i];
  }
}

// out_ptr[u] = number of kept links before row u
__global__ void __launch_bounds__(GB_BLOCK)
kb_out_ptr(uint32_t n, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ kpre,
           uint32_t *__restrict__ out_ptr, BuildInfo *__restrict__ info, uint32_t e) {
  const uint32_t u = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (u <= n) out_ptr[u] = kpre[row_ptr[u]];
  if (u == 0) info->kept = kpre[e];
}

__global__ void __launch_bounds__(GB_BLOCK)
kb_scatter(uint32_t e, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
           const uint32_t *__restrict__ metric, const uint8_t *__restrict__ vflags,
           const uint32_t *__restrict__ src_of, const uint8_t *__restrict__ keep, const uint32_t *__restrict__ kpre,
           const uint32_t *__restrict__ in_ptr, const uint32_t *__restrict__ lslot,
           uint32_t *__restrict__ out_dst, uint32_t *__restrict__ out_w, uint32_t *__restrict__ out_fpos,
           uint4 *__restrict__ tmp, BuildInfo *__restrict__ info) {
  // the in-row copy of a link goes to a random place: ONE 16-byte record (cost, source | NT, position, target) instead of
  // four 4-byte stores to four arrays (4 M scattered stores at 1 M links were most of this kernel's 56 us)
  const uint32_t k = blockIdx.x * GB_BLOCK + threadIdx.x;
  uint32_t wmax = 0;
  if (k < e && keep[k]) {
    const uint32_t u = src_of[k], t = col[k], w = metric[k];
    const uint32_t fpos = k - row_ptr[u];
    const uint32_t o = kpre[k];
    out_dst[o] = t; out_w[o] = w; out_fpos[o] = fpos;
    const uint32_t i = in_ptr[t] + lslot[k];
    // the overload gate only exists for routers (holo-isis/src/spf.rs:568-574: `!vertex.id.is_pseudonode()`): the bit on
    // a network vertex is ignored, as k_exact and the oracle do
    const uint32_t uf = vflags[u];
    tmp[i] = make_uint4(w, u | (((uf & HSPF_VF_NO_TRANSIT) && !(uf & HSPF_VF_NETWORK)) ? SRC_NO_TRANSIT : 0u), fpos, t);
    wmax = w;
  }
  for (int o = 32; o; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o));
  if ((threadIdx.x & 63u) == 0u) gb_raise_wmax(info, wmax);
}

// In-links of a row by (cost descending, source ascending, position ascending): among tight links, i.e. equal
// dist[u] + cost, the first in row order has the smallest dist[u] and then the smallest u = the reference's
// first discoverer (earliest popped tight parent).  k_fused relies on it; k_dag / k_exact do not care.
__global__ void __launch_bounds__(GB_BLOCK)
kb_rank(uint32_t e, const BuildInfo *__restrict__ info, const uint32_t *__restrict__ in_ptr,
        const uint4 *__restrict__ tmp,
        uint32_t *__restrict__ in_src, uint32_t *__restrict__ in_w, uint32_t *__restrict__ in_fpos, uint32_t hub_deg) {
  const uint32_t i = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (i >= e || i >= info->kept) return;
  const uint4 me = tmp[i];                                // (cost, source | NT, position, target)
  const uint32_t t = me.w;
  const uint32_t a = in_ptr[t], b = in_ptr[t + 1];
  const uint32_t w = me.x, sraw = me.y, s = sraw & SRC_MASK, f = me.z;
  if (b - a > hub_deg) {
    // parallel links piled onto one row: the host sees max_in_deg and rebuilds in hub mode.  The row is still WRITTEN —
    // unsorted, at its scatter position —: the remaining kernels of this (discarded) pass index other arrays with its
    // sources, and whatever the freshly allocated arena held before would send them anywhere (a GPU memory fault when
    // that was the 0xFFFFFFFF of a freed state slab: round 4, tests/test_gpu_graph_build.py after the async tests).
    in_w[i] = w; in_src[i] = sraw; in_fpos[i] = f;
    return;
  }
  uint32_t rank = 0;
  for (uint32_t j = a; j < b; ++j) {
    const uint4 o = tmp[j];
    const uint32_t wj = o.x, sj = o.y & SRC_MASK, fj = o.z;
    const bool before = wj > w || (wj == w && (sj < s || (sj == s && fj < f)));
    rank += before ? 1u : 0u;
  }
  in_w[a + rank] = w;
  in_src[a + rank] = sraw;
  in_fpos[a + rank] = f;
}

// ---- hub mode ------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(GB_BLOCK)
kb_hub_keys(uint32_t n, uint32_t e, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
            uint32_t *__restrict__ src_of, uint64_t *__restrict__ key) {
  const uint32_t k = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (k >= e) return;
  const uint32_t u = gb_row_of(row_ptr, n, k);
  src_of[k] = u;
  key[k] = ((uint64_t)u << 32) | (uint64_t)col[k];        // an out-of-range target is reported by kb_hub_links
}

__global__ void __launch_bounds__(GB_BLOCK)
kb_hub_links(uint32_t n, uint32_t e, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
             const uint32_t *__restrict__ metric, const uint8_t *__restrict__ vflags,
             const uint32_t *__restrict__ src_of, const uint64_t *__restrict__ sorted, uint8_t *__restrict__ twoway,
             uint8_t *__restrict__ keep, BuildInfo *__restrict__ info) {
  const uint32_t k = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (k >= e) return;
  const uint32_t u = src_of[k], t = col[k];
  uint32_t bad = 0;
  if (t >= n) bad |= GB_ERR_COL;
  if (metric[k] == INF) bad |= GB_ERR_METRIC;
  if (bad) {
    atomicOr(&info->err, bad);
    twoway[k] = 0; keep[k] = 0;
    return;
  }
  // two-way connectivity: (t, u) is among the sorted keys of row t
  const uint64_t want = ((uint64_t)t << 32) | (uint64_t)u;
  uint32_t lo = row_ptr[t], hi = row_ptr[t + 1];
  const uint32_t end = hi;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (sorted[mid] < want) lo = mid + 1; else hi = mid;
  }
  const bool two = lo < end && sorted[lo] == want;
  const bool kp = two && !(vflags[u] & HSPF_VF_NO_EXPAND);
  twoway[k] = two ? 1 : 0;
  keep[k] = kp ? 1 : 0;            // no in-degree histogram here (400 000 atomics on one counter): kb_hub_in_ptr
}

// in_ptr[t] = kept links whose target precedes t = first sorted (target << 32 | ~cost) key >= t << 32; dropped links sort
// behind every target, so in_ptr[n] = kept
__global__ void __launch_bounds__(GB_BLOCK)
kb_hub_in_ptr(uint32_t n, uint32_t e, const uint64_t *__restrict__ sorted, uint32_t *__restrict__ in_ptr) {
  const uint32_t t = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (t > n) return;
  const uint64_t want = (uint64_t)t << 32;
  uint32_t lo = 0, hi = e;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (sorted[mid] < want) lo = mid + 1; else hi = mid;
  }
  in_ptr[t] = lo;
}

__global__ void __launch_bounds__(GB_BLOCK)
kb_hub_scatter(uint32_t e, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
               const uint32_t *__restrict__ metric, const uint8_t *__restrict__ vflags,
               const uint32_t *__restrict__ src_of, const uint8_t *__restrict__ keep, const uint32_t *__restrict__ kpre,
               uint32_t *__restrict__ out_dst, uint32_t *__restrict__ out_w, uint32_t *__restrict__ out_fpos,
               uint64_t *__restrict__ key, uint32_t *__restrict__ val, uint32_t *__restrict__ srcflag,
               uint64_t dropped, BuildInfo *__restrict__ info) {
  __shared__ uint32_t bmax;
  if (threadIdx.x == 0) bmax = 0;
  __syncthreads();
  const uint32_t k = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (k < e) {
    if (keep[k]) {
      const uint32_t u = src_of[k], t = col[k], w = metric[k];
      const uint32_t fpos = k - row_ptr[u];
      const uint32_t o = kpre[k];
      out_dst[o] = t; out_w[o] = w; out_fpos[o] = fpos;
      const uint32_t uf = vflags[u];                       // see kb_scatter
      srcflag[o] = u | (((uf & HSPF_VF_NO_TRANSIT) && !(uf & HSPF_VF_NETWORK)) ? SRC_NO_TRANSIT : 0u);
      key[k] = ((uint64_t)t << 32) | (uint64_t)(~w);       // cost descending
      val[k] = o;
      atomicMax(&bmax, w);
    } else {
      key[k] = dropped; val[k] = 0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && bmax) gb_raise_wmax(info, bmax);
}

__global__ void __launch_bounds__(GB_BLOCK)
kb_hub_gather(uint32_t e, const BuildInfo *__restrict__ info, const uint64_t *__restrict__ sorted,
              const uint32_t *__restrict__ perm, const uint32_t *__restrict__ srcflag, const uint32_t *__restrict__ out_fpos,
              uint32_t *__restrict__ in_src, uint32_t *__restrict__ in_w, uint32_t *__restrict__ in_fpos) {
  const uint32_t i = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (i >= e || i >= info->kept) return;
  const uint32_t o = perm[i];
  in_w[i] = ~(uint32_t)sorted[i];
  in_src[i] = srcflag[o];
  in_fpos[i] = out_fpos[o];
}

// Static reasons why a row needs the general fused routine (RF_*), the hop-count shape of the graph
// (MetricMode::HopCount graphs, holo-isis/src/spf.rs:1131-1146: cost 0 into a pseudonode, 1 into a router), and the
// 16 zero entries behind every array that the kernels' fixed-size fetches may touch.
__global__ void __launch_bounds__(GB_BLOCK)
kb_rowflags(uint32_t n, const uint32_t *__restrict__ in_ptr, const uint32_t *__restrict__ in_src,
            const uint32_t *__restrict__ in_w, const uint8_t *__restrict__ vflags, uint8_t *__restrict__ rowflags,
            uint8_t *__restrict__ rowaux, BuildInfo *__restrict__ info, uint32_t giant_deg) {
  // sixteen lanes per row (a wave = four rows): a row's links arrive in one coalesced load per array and sixteen links,
  // all rows of the wave in flight together (one thread per row walked its links one dependent load after the other:
  // 50-130 us for 100 000 rows of ten links); a row of more than 256 in-links is walked by its whole wave afterwards (a
  // LAN with thousands of members would otherwise be sixteen lanes' serial loop)
  const uint32_t gt = blockIdx.x * GB_BLOCK + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t t = gt >> 4, j = gt & 15u;
  const bool valid = t < n;
  const uint32_t a = valid ? in_ptr[t] : 0u, b = valid ? in_ptr[t + 1] : 0u;
  const bool net = valid && (vflags[t] & HSPF_VF_NETWORK);
  const bool wide = b - a > 256u;
  uint32_t f = 0;
  bool bad = false;
  auto link = [&](uint32_t i, uint32_t row, bool row_net, uint32_t &ff, bool &bb) {
    const uint32_t sraw = in_src[i], u = sraw & SRC_MASK, w = in_w[i];
    if (sraw & SRC_NO_TRANSIT) ff |= RF_NT;
    if (w == 0u && u >= row) ff |= RF_ZERO;
    if (row_net) bb |= !(w == 0u && !(vflags[u] & HSPF_VF_NETWORK) && u > row);
    else bb |= w != 1u;
  };
  if (!wide)
    for (uint32_t i = a + j; i < b; i += 16u) link(i, t, net, f, bad);
  for (int o = 8; o; o >>= 1) {                            // the row's sixteen lanes (xor below 16 stays inside the group)
    const int ob = __shfl_xor((int)bad, o);                // (every lane shuffles: no short-circuit around it)
    f |= (uint32_t)__shfl_xor((int)f, o);
    bad = bad || ob != 0;
  }
  for (uint64_t todo = __ballot(wide && j == 0u); todo != 0ull; todo &= todo - 1ull) {
    const int l = __ffsll((unsigned long long)todo) - 1;
    const uint32_t row = __shfl(t, l), ra = __shfl(a, l), rb = __shfl(b, l);
    const bool row_net = __shfl((int)net, l) != 0;
    uint32_t ff = 0;
    bool bb = false;
#pragma unroll 4
    for (uint32_t i = ra + lane; i < rb; i += 64u) link(i, row, row_net, ff, bb);
    const uint32_t all = (__ballot(ff & RF_NT) ? RF_NT : 0u) | (__ballot(ff & RF_ZERO) ? RF_ZERO : 0u);
    const bool any_bad = __ballot(bb) != 0ull;
    if ((int)lane == l) { f |= all; bad = any_bad; }
  }
  f |= (b - a > 16u ? RF_MANY : 0u) | (b - a > giant_deg ? RF_GIANT : 0u);
  const bool head = valid && j == 0u;
  if (head) {
    rowflags[t] = (uint8_t)f;
    rowaux[t] = (uint8_t)((bad ? RA_BAD : 0u) | ((net && b > a) ? RA_NET_IN : 0u));   // what the summary counts, per row (graph_patch.hip.h)
  }
  // the summary: reduced over the wave, then over the block; maximum and OR only touch BuildInfo when they would change
  // it, the two counts go to the spread counters (gb_spread)
  __shared__ uint32_t red[4][5];
  uint32_t mx = head ? b - a : 0u, fo = head ? f : 0u;
  for (int o = 32; o; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); fo |= (uint32_t)__shfl_xor((int)fo, o); }
  const uint64_t m_bad = __ballot(head && bad), m_zero = __ballot(head && (f & RF_ZERO)), m_net = __ballot(head && net && b > a);
  if (lane == 0u) {
    uint32_t *r = red[threadIdx.x >> 6];
    r[0] = mx; r[1] = fo; r[2] = (uint32_t)__popcll(m_bad); r[3] = (uint32_t)__popcll(m_zero); r[4] = m_net ? 1u : 0u;
  }
  __syncthreads();
  if (threadIdx.x != 0u) return;
  uint32_t bmx = 0, bfo = 0, nbad = 0, nzero = 0, bnet = 0;
  for (int w = 0; w < GB_BLOCK / 64; ++w) { bmx = max(bmx, red[w][0]); bfo |= red[w][1]; nbad += red[w][2]; nzero += red[w][3]; bnet |= red[w][4]; }
  if (bmx > *(volatile uint32_t *)&info->max_in_deg) atomicMax(&info->max_in_deg, bmx);
  if (bfo & ~*(volatile uint32_t *)&info->any_rowflags) atomicOr(&info->any_rowflags, bfo);
  if (nbad) atomicAdd(gb_spread(info, GB_SC_BAD), nbad);
  if (nzero) atomicAdd(gb_spread(info, GB_SC_ZERO), nzero);
  if (bnet && !*(volatile uint32_t *)&info->hc_net) info->hc_net = 1u;   // plain store: every writer stores the same value
}

// Work units (GraphDev::unit_first): heavy flag per 16-vertex chunk, then (after a scan of the flags) the unit table
// [4 units per heavy chunk | 1 unit per other chunk], each class in vertex order.
__global__ void __launch_bounds__(GB_BLOCK)
kb_unit_count(uint32_t n, const uint32_t *__restrict__ in_ptr, uint32_t *__restrict__ hf, uint32_t heavy_deg) {
  const uint32_t c = blockIdx.x * GB_BLOCK + threadIdx.x;
  const uint32_t nb = (n + 15u) / 16u;
  if (c >= nb) return;
  bool heavy = false;
  for (uint32_t v = c * 16u; v < min(c * 16u + 16u, n); ++v) heavy = heavy || (in_ptr[v + 1] - in_ptr[v] > heavy_deg);
  hf[c] = heavy ? 1u : 0u;
}

__global__ void __launch_bounds__(GB_BLOCK)
kb_unit_fill(uint32_t n, const uint32_t *__restrict__ hf, const uint32_t *__restrict__ hpos, uint32_t *__restrict__ unit_first,
             BuildInfo *__restrict__ info) {
  const uint32_t c = blockIdx.x * GB_BLOCK + threadIdx.x;
  const uint32_t nb = (n + 15u) / 16u;
  const uint32_t nh = hpos[nb];                       // heavy chunks
  if (c == 0) info->n_heavy = nh;
  if (c >= nb) return;
  const uint32_t hb = hpos[c];                        // heavy chunks before c
  if (hf[c]) for (uint32_t k = 0; k < 4u; ++k) unit_first[4u * hb + k] = min(c * 16u + 4u * k, n) | UNIT_SPLIT;
  else unit_first[4u * nh + (c - hb)] = c * 16u;
}

// scan + kb_unit_fill + kb_xcd + kb_pads in ONE workgroup behind kb_unit_count, for graphs of up to 65 536 chunks (1 M
// vertices): six dependent launches of a few microseconds of work each were a fifth of a structural patch's launch chain.
// Thread t owns the chunks [t ipt, (t + 1) ipt): heavy flags in a bit mask, heavy chunks in front of its range by a block scan.
constexpr int GB_UNITS_THREADS = 1024;
constexpr uint32_t GB_UNITS_MAX_CHUNKS = 65536;
__device__ __forceinline__ void kb_xcd_body(uint32_t x, uint32_t n, const uint32_t *__restrict__ in_ptr, BuildInfo *__restrict__ info,
                                            uint32_t row_cost, uint32_t n_heavy);
__device__ __forceinline__ void kb_xcd_wave(uint32_t x, uint32_t lane, uint32_t n, const uint32_t *__restrict__ in_ptr, BuildInfo *info,
                                            uint32_t row_cost, uint32_t n_heavy);
__global__ void __launch_bounds__(GB_UNITS_THREADS)
kb_units_small(uint32_t n, const uint32_t *__restrict__ in_ptr_c, const uint32_t *__restrict__ hf, uint32_t *__restrict__ unit_first,
               BuildInfo *info, uint32_t row_cost, uint32_t *in_ptr, uint32_t *out_ptr,
               uint32_t *a0, uint32_t *a1, uint32_t *a2, uint32_t *a3, uint32_t *a4, uint32_t *a5, BuildInfo *host_info) {
  __shared__ uint32_t sh[GB_UNITS_THREADS];
  const uint32_t t = threadIdx.x;
  const uint32_t nb = (n + 15u) / 16u;
  const uint32_t ipt = (nb + GB_UNITS_THREADS - 1u) / GB_UNITS_THREADS;     // <= 64
  const uint32_t c0 = min(t * ipt, nb), c1 = min(c0 + ipt, nb);
  uint64_t heavy = 0ull;                                   // the flags of kb_unit_count (the pass over the rows stays a grid's work)
  for (uint32_t c = c0; c < c1; ++c)
    if (hf[c]) heavy |= 1ull << (c - c0);
  const uint32_t mine = (uint32_t)__popcll(heavy);
  sh[t] = mine;
  __syncthreads();
  for (int d = 1; d < GB_UNITS_THREADS; d <<= 1) {
    const uint32_t add = (int)t >= d ? sh[t - d] : 0u;
    __syncthreads();
    sh[t] += add;
    __syncthreads();
  }
  const uint32_t nh = sh[GB_UNITS_THREADS - 1];
  uint32_t hb = sh[t] - mine;                                    // heavy chunks before c0
  for (uint32_t c = c0; c < c1; ++c) {
    if ((heavy >> (c - c0)) & 1ull) {
      for (uint32_t k = 0; k < 4u; ++k) unit_first[4u * hb + k] = min(c * 16u + 4u * k, n) | UNIT_SPLIT;
      ++hb;
    } else {
      unit_first[4u * nh + (c - hb)] = c * 16u;
    }
  }
  if (t == 0u) info->n_heavy = nh;
  if (t < 576u) kb_xcd_wave(t >> 6, t & 63u, n, in_ptr_c, info, row_cost, nh);      // waves 0 .. 8: one XCD bound each
  if (t >= 640u && t < 704u) gb_counts_finish(info, t - 640u);
  if (t >= 576u && t < 592u) {                                   // kb_pads (info->kept was written by kb_out_ptr, launches ago)
    const uint32_t i = t - 576u, kept = info->kept;
    in_ptr[n + 1 + i] = 0; out_ptr[n + 1 + i] = 0;
    a0[kept + i] = 0; a1[kept + i] = 0; a2[kept + i] = 0; a3[kept + i] = 0; a4[kept + i] = 0; a5[kept + i] = 0;
  }
  if (host_info) {                                               // the summary straight into the caller's page-locked block (a copy packet less on a patch's chain)
    __syncthreads();
    __threadfence();
    if (t < sizeof(BuildInfo) / 4u) ((volatile uint32_t *)host_info)[t] = ((volatile uint32_t *)info)[t];
  }
}

// Work-balanced XCD ranges for graphs without heavy chunks (GraphDev::xcd_start): cost of a 16-vertex chunk prefix k =
// in-links of the first 16k vertices + 8 per vertex (a row costs about 8 links' worth of fixed work; HSPF_XCD_ROW_COST);
// XCD x starts at the first chunk whose prefix reaches x/8 of the total.  Seven binary searches, one thread each.
// With heavy chunks: even shares of the normal units (the heavy ones are split evenly by the host, GraphDev::xcd_heavy).
__device__ __forceinline__ void kb_xcd_body(uint32_t x, uint32_t n, const uint32_t *__restrict__ in_ptr, BuildInfo *__restrict__ info,
                                            uint32_t row_cost, uint32_t n_heavy) {
  const uint32_t nb = (n + 15u) / 16u;
  if (n_heavy) {
    const uint32_t nn = nb - n_heavy;
    info->xcd_start[x] = (uint32_t)((uint64_t)nn * x / 8ull);
    return;
  }
  auto cost = [&](uint32_t k) -> uint64_t { const uint32_t v = min(k * 16u, n); return (uint64_t)in_ptr[v] + (uint64_t)row_cost * v; };
  const uint64_t total = cost(nb), want = total * x / 8ull;
  uint32_t lo = 0, hi = nb;                       // smallest k with cost(k) >= want
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (cost(mid) >= want) hi = mid; else lo = mid + 1;
  }
  info->xcd_start[x] = x == 8u ? nb : lo;
}
// The same bound found by a whole wave: 64 probes per round instead of one (a binary search over 6 250 chunks is 13 dependent
// trips to the L2, ~8 us of a structural patch's chain; three rounds of 64 probes are ~2).  Smallest k in [0, nb] with
// cost(k) >= want; cost(nb) = total >= want.
__device__ __forceinline__ void kb_xcd_wave(uint32_t x, uint32_t lane, uint32_t n, const uint32_t *__restrict__ in_ptr, BuildInfo *info,
                                            uint32_t row_cost, uint32_t n_heavy) {
  const uint32_t nb = (n + 15u) / 16u;
  if (n_heavy) {
    if (lane == 0u) info->xcd_start[x] = (uint32_t)((uint64_t)(nb - n_heavy) * x / 8ull);
    return;
  }
  auto cost = [&](uint32_t k) -> uint64_t { const uint32_t v = min(k * 16u, n); return (uint64_t)in_ptr[v] + (uint64_t)row_cost * v; };
  const uint64_t total = cost(nb), want = total * x / 8ull;
  uint32_t lo = 0, hi = nb;                        // the answer is in [lo, hi]; cost(hi) >= want
  while (lo < hi) {
    const uint32_t step = max((hi - lo + 62u) / 63u, 1u);                 // lanes probe lo, lo + step, ..: lane 63 reaches hi
    const uint32_t k = min(lo + lane * step, hi);
    const uint64_t m = __ballot(cost(k) >= want);                         // monotone: a run of lanes from the first true one (lane 63 probes hi: never empty)
    const int f = __ffsll((unsigned long long)m) - 1;
    const uint32_t kf = min(lo + (uint32_t)f * step, hi);
    const uint32_t kprev = f ? min(lo + (uint32_t)(f - 1) * step, hi) : lo;
    hi = kf;
    lo = f ? kprev + 1u : lo;
    if (f == 0) break;                                                      // cost(lo) >= want: lo is the answer (hi = lo)
  }
  if (lane == 0u) info->xcd_start[x] = x == 8u ? nb : hi;
}
__global__ void kb_xcd(uint32_t n, const uint32_t *__restrict__ in_ptr, BuildInfo *__restrict__ info, uint32_t row_cost) {
  if (threadIdx.x <= 8u) kb_xcd_body(threadIdx.x, n, in_ptr, info, row_cost, info->n_heavy);
  gb_counts_finish(info, threadIdx.x);                      // (launched with 64 threads)
}

// Leaves (GraphDev::leaf): exactly one kept in-link, and the kept out-links (at most one) lead back to its source.
__global__ void __launch_bounds__(GB_BLOCK)
kb_leaf_mark(uint32_t n, const uint32_t *__restrict__ in_ptr, const uint32_t *__restrict__ in_src,
             const uint32_t *__restrict__ out_ptr, const uint32_t *__restrict__ out_dst, uint8_t *__restrict__ leaf,
             BuildInfo *__restrict__ info) {
  const uint32_t v = blockIdx.x * GB_BLOCK + threadIdx.x;
  bool is = false;
  if (v < n) {
    const uint32_t e0 = in_ptr[v], o0 = out_ptr[v], od = out_ptr[v + 1] - o0;
    is = in_ptr[v + 1] - e0 == 1u && (od == 0u || (od == 1u && out_dst[o0] == (in_src[e0] & SRC_MASK)));
    leaf[v] = is ? 1u : 0u;
  }
  const uint64_t m = __ballot(is);                         // (a fat-tree has 250 000 leaves: not one atomic each on one word)
  if ((threadIdx.x & 63u) == 0u && m) atomicAdd(gb_spread(info, GB_SC_LEAF), (uint32_t)__popcll(m));
}

// ... and SRC_LEAF on every link that comes from one
__global__ void __launch_bounds__(GB_BLOCK)
kb_leaf_links(uint32_t e, const BuildInfo *__restrict__ info, uint32_t *__restrict__ in_src, const uint8_t *__restrict__ leaf) {
  const uint32_t i = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (i >= e || i >= info->kept) return;
  const uint32_t s = in_src[i];
  if (leaf[s & SRC_MASK]) in_src[i] = s | SRC_LEAF;
}

// ---- hspf_graph_upload_keyed: LSDB records -> CSR on the device ---------------------------------------------
// The caller hands over what sits in its LSDB: per vertex a 64-bit KEY (ascending key order = the reference's VertexId order)
// and its links as (target key, cost) in LSA / LSP link order — vertices in ANY order, targets unresolved.  What the host twins
// did per link on one core (a binary search among 100 000 vertex ids, a million times: 148 ms) happens here: the keys are
// ranked by a stable radix sort (hub_sort.h), every link's target is looked up in the sorted keys, links to vertices that are
// not in the LSDB are dropped (vertex_lsa_links / vertex_edges do not yield them), rows are laid out in rank order.
__global__ void __launch_bounds__(GB_BLOCK)
kb_kx_iota(uint32_t n, uint32_t *__restrict__ v) {
  const uint32_t i = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ void __launch_bounds__(GB_BLOCK)
kb_kx_rank(uint32_t n, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ skey, const uint8_t *__restrict__ vfl_in,
           uint32_t *__restrict__ rank, uint8_t *__restrict__ vflags, uint32_t *__restrict__ err) {
  const uint32_t r = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (r >= n) return;
  const uint32_t i = perm[r];
  rank[i] = r;
  vflags[r] = vfl_in[i];
  if (r && skey[r] == skey[r - 1]) atomicOr(err, 1u);               // the same vertex twice
}
__global__ void __launch_bounds__(GB_BLOCK)
kb_kx_resolve(uint32_t m, const uint64_t *__restrict__ tkey, const uint64_t *__restrict__ skey, uint32_t n, uint32_t *__restrict__ keep,
              uint32_t *__restrict__ tidx) {
  const uint32_t j = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (j >= m) return;
  const uint64_t k = tkey[j];
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skey[mid] < k) lo = mid + 1; else hi = mid; }
  const bool found = lo < n && skey[lo] == k;
  keep[j] = found ? 1u : 0u;
  tidx[j] = lo;
}
__global__ void __launch_bounds__(GB_BLOCK)
kb_kx_deg(uint32_t n, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ vrow, const uint32_t *__restrict__ kpre, uint32_t *__restrict__ deg) {
  const uint32_t r = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (r >= n) return;
  const uint32_t i = perm[r];
  deg[r] = kpre[vrow[i + 1]] - kpre[vrow[i]];
}
__global__ void __launch_bounds__(GB_BLOCK)
kb_kx_scatter(uint32_t n, const uint32_t *__restrict__ rank, const uint32_t *__restrict__ vrow, const uint32_t *__restrict__ keep,
              const uint32_t *__restrict__ kpre, const uint32_t *__restrict__ tidx, const uint32_t *__restrict__ tmet,
              const uint32_t *__restrict__ row_ptr, uint32_t *__restrict__ col, uint32_t *__restrict__ metric) {
  // 16 lanes per input vertex: its links in their order, the kept ones at row_ptr[rank] + (kept links before them in the row)
  const uint32_t t = blockIdx.x * GB_BLOCK + threadIdx.x, i = t >> 4, sub = t & 15u;
  if (i >= n) return;
  const uint32_t a = vrow[i], b = vrow[i + 1], base = row_ptr[rank[i]], k0 = kpre[a];
  for (uint32_t j = a + sub; j < b; j += 16u)
    if (keep[j]) { const uint32_t d = base + (kpre[j] - k0); col[d] = tidx[j]; metric[d] = tmet[j]; }
}

// GraphDev::zcyc — which vertices may lie on a CYCLE of zero-cost kept links.  Trimming: a vertex stays alive while it has a
// zero-cost in-link from an alive vertex AND a zero-cost out-link to an alive vertex; vertices on a cycle never die, whatever
// survives GB_ZC_ROUNDS rounds is a superset of them (long zero-cost chains end up in it too: conservative).  Used by the sweep
// kernels to decide which zero-cost links from higher-numbered sources may feed hops / masks (spf_kernels.hip.h finish_row_z).
constexpr int GB_ZC_ROUNDS = 8;
__global__ void __launch_bounds__(GB_BLOCK)
kb_zc_init(uint32_t n, const uint32_t *__restrict__ in_ptr, const uint32_t *__restrict__ in_w, const uint32_t *__restrict__ out_ptr,
           const uint32_t *__restrict__ out_w, uint8_t *__restrict__ alive) {
  const uint32_t v = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (v >= n) return;
  bool zi = false, zo = false;
  for (uint32_t e = in_ptr[v], e1 = in_ptr[v + 1]; e < e1 && !zi; ++e) zi = in_w[e] == 0u;
  if (zi) for (uint32_t k = out_ptr[v], k1 = out_ptr[v + 1]; k < k1 && !zo; ++k) zo = out_w[k] == 0u;
  alive[v] = (zi && zo) ? 1 : 0;
}
__global__ void __launch_bounds__(GB_BLOCK)
kb_zc_round(uint32_t n, const uint32_t *__restrict__ in_ptr, const uint32_t *__restrict__ in_src, const uint32_t *__restrict__ in_w,
            const uint32_t *__restrict__ out_ptr, const uint32_t *__restrict__ out_dst, const uint32_t *__restrict__ out_w,
            const uint8_t *__restrict__ a, uint8_t *__restrict__ b) {
  const uint32_t v = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (v >= n) return;
  bool zi = false, zo = false;
  if (a[v]) {
    for (uint32_t e = in_ptr[v], e1 = in_ptr[v + 1]; e < e1 && !zi; ++e) zi = in_w[e] == 0u && a[in_src[e] & SRC_MASK];
    if (zi) for (uint32_t k = out_ptr[v], k1 = out_ptr[v + 1]; k < k1 && !zo; ++k) zo = out_w[k] == 0u && a[out_dst[k]];
  }
  b[v] = (zi && zo) ? 1 : 0;
}

// Fixed-stride (ELL) copy of the link records for k_fused_lean: 16 entries per vertex, rows 0 .. n (row n = all pad).
//   ell_so[16 v + j] = byte offset of the source's state row (source << 8); j >= in-degree: the pad row n (never reached);
//                      the low byte of entry 0 carries  in-degree (0 .. 16; more: 0x1F) | (more than 16 out-links) << 5 |
//                      network << 7
//   ell_w [16 v + j] = cost; pad entries 0
//   ell_od[16 v + j] = byte offset of out-neighbour j's activation stamp (target << 2); j >= out-degree: 0xFFFFFFFF (a
//                      buffer store at that offset is out of range and dropped)
// Only rows of at most 16 in-links are read through it (longer rows carry RF_MANY and take the general routine).
__global__ void __launch_bounds__(GB_BLOCK)
kb_ell(uint32_t n, const uint32_t *__restrict__ in_ptr, const uint32_t *__restrict__ in_src, const uint32_t *__restrict__ in_w,
       const uint32_t *__restrict__ out_ptr, const uint32_t *__restrict__ out_dst, const uint8_t *__restrict__ vflags,
       uint32_t *__restrict__ ell_so, uint32_t *__restrict__ ell_w, uint32_t *__restrict__ ell_od) {
  const uint32_t t = blockIdx.x * GB_BLOCK + threadIdx.x;
  const uint32_t v = t >> 4, j = t & 15u;
  if (v > n) return;
  uint32_t so = n << 8, w = 0u, od = 0xFFFFFFFFu, info = 0u;
  if (v < n) {
    const uint32_t e0 = in_ptr[v], deg = in_ptr[v + 1] - e0;
    const uint32_t o0 = out_ptr[v], odeg = out_ptr[v + 1] - o0;
    info = (deg <= 16u ? deg : 0x1Fu) | (odeg > 16u ? 0x20u : 0u) | ((vflags[v] & HSPF_VF_NETWORK) ? 0x80u : 0u);
    if (deg <= 16u && j < deg) { so = (in_src[e0 + j] & SRC_MASK) << 8; w = in_w[e0 + j]; }
    if (j < odeg) od = out_dst[o0 + j] << 2;
  }
  if (j == 0u) so |= info;
  ell_so[t] = so;
  ell_w[t] = w;
  ell_od[t] = od;
}

__global__ void kb_pads(uint32_t n, const BuildInfo *__restrict__ info, uint32_t *in_ptr, uint32_t *out_ptr,
                        uint32_t *a0, uint32_t *a1, uint32_t *a2, uint32_t *a3, uint32_t *a4, uint32_t *a5) {
  const uint32_t i = threadIdx.x;
  if (i >= 16) return;
  const uint32_t kept = info->kept;
  in_ptr[n + 1 + i] = 0; out_ptr[n + 1 + i] = 0;
  a0[kept + i] = 0; a1[kept + i] = 0; a2[kept + i] = 0; a3[kept + i] = 0; a4[kept + i] = 0; a5[kept + i] = 0;
}

// ---- patch: replace whole rows of the raw CSR: kb_patch_raw (graph_patch.hip.h) writes the new row bounds (old ones + the length
// changes of the replaced rows in front: no 400 KB upload of the bounds per patch), the replaced rows' flags and the spliced
// targets / costs (a link's source is found among the REPLACED rows, not among all rows) in one launch.

// ---- patch, fast path: the replaced rows list the SAME targets in the same order with the same flags, only costs differ
// (a metric change: the commonest reason an LSP / LSA is re-originated; holo-isis/src/spf.rs:144, 733-735 `trigger_lsps`).
// Nothing moves: kept sets, degrees, row bounds, work units and XCD ranges stay; what changes is the cost of the
// replaced rows' links in the raw CSR, the forward arrays and the in-rows of their targets, whose ORDER (cost descending,
// source ascending, position ascending) then has to be restored, together with the targets' row flags and ELL records.
// Work: O(links of the replaced rows x in-degree of their targets), two launches.  The layout equals a fresh upload's.
struct PatchInfo { uint32_t wmax; uint32_t stale; int32_t d_zero; int32_t d_bad; };

// one workgroup per replaced row b (vertex changed[b]): raw costs and forward costs
__global__ void __launch_bounds__(256)
kb_pc_apply(const uint32_t *__restrict__ changed, const uint32_t *__restrict__ dptr, const uint32_t *__restrict__ dmet,
            const uint32_t *__restrict__ row_ptr, uint32_t *__restrict__ metric, const uint32_t *__restrict__ out_ptr,
            uint32_t *__restrict__ out_w, const uint32_t *__restrict__ out_fpos) {
  const uint32_t b = blockIdx.x, u = changed[b];
  const uint32_t d0 = dptr[b], len = dptr[b + 1] - d0, r0 = row_ptr[u];
  for (uint32_t k = threadIdx.x; k < len; k += 256u) metric[r0 + k] = dmet[d0 + k];
  for (uint32_t o = out_ptr[u] + threadIdx.x; o < out_ptr[u + 1]; o += 256u) out_w[o] = dmet[d0 + out_fpos[o]];
}

// one workgroup per affected target (in-degree <= 256): the new costs of its entries (sources looked up in the sorted
// changed list), their order (cost descending, then source, then position: kb_rank / kb_hub_gather), the row's flags
// (kb_rowflags) and ELL records (kb_ell), and what the summary of the build needs: largest new cost, whether a link that
// carried the old maximum got cheaper, and the change in the counts of RF_ZERO rows and of rows off the hop-count shape
__global__ void __launch_bounds__(256)
kb_pc_resort(uint32_t n_changed, const uint32_t *__restrict__ changed, const uint32_t *__restrict__ dptr,
             const uint32_t *__restrict__ dmet, const uint32_t *__restrict__ targets, const uint32_t *__restrict__ in_ptr,
             uint32_t *__restrict__ in_src, uint32_t *__restrict__ in_w, uint32_t *__restrict__ in_fpos,
             const uint8_t *__restrict__ vflags, uint8_t *__restrict__ rowflags, uint8_t *__restrict__ rowaux, uint32_t *__restrict__ ell_so,
             uint32_t *__restrict__ ell_w, uint32_t giant_deg, uint32_t wmax_old, PatchInfo *__restrict__ pinfo) {
  __shared__ uint32_t s_src[256], s_w[256], s_pos[256];
  const uint32_t t = targets[blockIdx.x], a = in_ptr[t], d = in_ptr[t + 1] - a, i = threadIdx.x;
  const bool row_net = (vflags[t] & HSPF_VF_NETWORK) != 0;
  uint32_t sraw = 0, w = 0, f = 0, src = 0;
  bool bad_old = false, bad_new = false, zero_new = false, nt = false;
  if (i < d) {
    sraw = in_src[a + i]; w = in_w[a + i]; f = in_fpos[a + i]; src = sraw & SRC_MASK;
    uint32_t lo = 0, hi = n_changed;                      // first index with changed[index] >= src
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (changed[mid] < src) lo = mid + 1; else hi = mid; }
    uint32_t wn = w;
    if (lo < n_changed && changed[lo] == src) {
      wn = dmet[dptr[lo] + f];
      atomicMax(&pinfo->wmax, wn);
      if (w == wmax_old && wn < w) pinfo->stale = 1u;
    }
    const bool src_net = (vflags[src] & HSPF_VF_NETWORK) != 0;
    bad_old = row_net ? !(w == 0u && !src_net && src > t) : w != 1u;
    bad_new = row_net ? !(wn == 0u && !src_net && src > t) : wn != 1u;
    zero_new = wn == 0u && src >= t;
    nt = (sraw & SRC_NO_TRANSIT) != 0u;
    w = wn;
    s_src[i] = src; s_w[i] = w; s_pos[i] = f;
  }
  __syncthreads();
  uint32_t rank = 0;
  if (i < d) {
    for (uint32_t j = 0; j < d; ++j) {
      const uint32_t wj = s_w[j], sj = s_src[j], fj = s_pos[j];
      rank += (wj > w || (wj == w && (sj < src || (sj == src && fj < f)))) ? 1u : 0u;
    }
    in_src[a + rank] = sraw; in_w[a + rank] = w; in_fpos[a + rank] = f;
  }
  const int any_zero = __syncthreads_or(zero_new ? 1 : 0), any_nt = __syncthreads_or(nt ? 1 : 0);
  const int any_bad_old = __syncthreads_or(bad_old ? 1 : 0), any_bad_new = __syncthreads_or(bad_new ? 1 : 0);
  if (i == 0) {
    const uint32_t was = rowflags[t];
    const uint32_t fl = (d > 16u ? RF_MANY : 0u) | (d > giant_deg ? RF_GIANT : 0u) | (any_nt ? RF_NT : 0u) | (any_zero ? RF_ZERO : 0u);
    rowflags[t] = (uint8_t)fl;
    rowaux[t] = (uint8_t)((any_bad_new ? RA_BAD : 0u) | ((row_net && d > 0u) ? RA_NET_IN : 0u));   // (kb_rowflags; read by the next structural patch's summary)
    const int dz = (int)((fl & RF_ZERO) != 0u) - (int)((was & RF_ZERO) != 0u), db = (any_bad_new != 0) - (any_bad_old != 0);
    if (dz) atomicAdd(&pinfo->d_zero, dz);
    if (db) atomicAdd(&pinfo->d_bad, db);
  }
  if (d <= 16u && i < d) {
    const uint32_t info = d | (row_net ? 0x80u : 0u) | (ell_so[(size_t)t * 16u] & 0x20u);   // bit 5 (out-links) is untouched
    ell_so[(size_t)t * 16u + rank] = (src << 8) | (rank == 0u ? info : 0u);
    ell_w[(size_t)t * 16u + rank] = w;
  }
}

// largest kept cost, for the patch that made a link carrying the old maximum cheaper
__global__ void __launch_bounds__(GB_BLOCK)
kb_pc_wmax(uint32_t e_kept, const uint32_t *__restrict__ out_w, PatchInfo *__restrict__ pinfo) {
  uint32_t m = 0;
  for (uint32_t k = blockIdx.x * GB_BLOCK + threadIdx.x; k < e_kept; k += gridDim.x * GB_BLOCK) m = max(m, out_w[k]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63u) == 0u && m) atomicMax(&pinfo->wmax, m);
}

}  // namespace hspf
