// spf_repair.hip.h — roots whose pop order is DYNAMIC (zero-cost router links), resolved in parallel.
//
// The reference's candidate list is an ordered map keyed (distance, VertexId) that a vertex enters when its first parent is
// popped (holo-isis/src/spf.rs:629-704, holo-ospf/src/spf.rs:666-719).  With positive costs every parent of v is popped
// before anything at v's distance, the pop order is the static (distance, index) order and the sweep kernels are exact.  A
// zero-cost link u -> v with dist[u] == dist[v] breaks that: v enters the list only when u is popped, so a v with a LOWER
// index than u is popped after it, and whether another equal-distance vertex counts as a parent of v (`nexthops.extend`,
// first discoverer for `hops`) depends on who was on the list when.  Distances do not depend on the order; the sweep kernels
// store them for every root, flag such a root LF_DYN and leave placeholders where a vertex' only way in is a zero-cost link
// from a higher-numbered source (finish_row).  Until round 6 the whole root was then re-run by k_exact — one GPU THREAD walking
// a binary heap over the whole graph: seconds per batch at 100 k vertices.  This file replaces that for LF_DYN roots.
//
// The order, in closed form.  Fix a root and a distance level (all vertices with dist == d).  SEEDS are the level's vertices
// with a tight parent at a lower level (or the root): they are on the candidate list when the level starts.  The level is
// popped lowest index first among what is on the list; popping u releases its zero-cost tight children.  Hence:
//   R(v)   = min over release paths (seed -> ... -> v along zero-cost tight links) of the LARGEST index on the path
//            (seeds: R = v).  Everything with R <= m is popped before anything with an index > m, so the level is popped in
//            ascending R; R(v) == v for every vertex that no higher-numbered vertex has to release ("natural": all but a few);
//   group  = the vertices with the same R = y other than y itself: they are released (directly or through each other) by y,
//            have smaller indices than y and are popped right after y, before anything else — again lowest index first among
//            the released ones: a priority walk over the handful of group members, sequential, per group;
//   pos(v) = position inside its group (y: 0).  The pop order of the root is the order of the keys (dist, R, pos).
// Then   parents(v) = tight in-links whose source is expanded and precedes v in that order,
//        hops(v)    = hops(first parent in that order) + is_router(v),      mask(v) = OR over parents (slot bit | mask(parent)).
// For natural vertices whose parents are natural this IS the static rule — the emitted rows are right already; k_repair
// re-evaluates the vertices with a zero-cost tight link from a higher-numbered source and the tight children of non-natural
// vertices, and from there whatever changes (worklist sweeps, one workgroup per root: the roots are independent).
//
// Cost follows the number of zero-cost links and the size of the changed cones, not the size of the graph.  A root whose
// group does not fit the walk's fixed-size heap, or that has no hops / mask arrays to repair, still goes to k_exact.
#pragma once
#include "spf_kernels.hip.h"

namespace hspf {

constexpr uint32_t RP_THREADS = 1024u;
constexpr uint32_t RP_HEAP = 48u;               // members of one group that can be on the walk's list at once
constexpr uint32_t RP_UNRES = 0xFFFFFFFFu;      // R of a vertex no release path has reached yet
constexpr uint32_t RP_INHEAP = 0xFFFFFFFFu;     // pos of a group member that is on the walk's list
constexpr uint32_t RP_ST_FAIL = 1u;             // status bit: the root must go to k_exact

// ---- vertices with a zero-cost kept in-link (the only ones whose R can differ from their index) -----------------------------
__global__ __launch_bounds__(256) void kr_zmark(uint32_t e, const uint32_t *__restrict__ out_dst, const uint32_t *__restrict__ out_w,
                                                uint8_t *__restrict__ zflag) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k < e && out_w[k] == 0u) zflag[out_dst[k]] = 1;
}
__global__ __launch_bounds__(256) void kr_zcompact(uint32_t n, const uint8_t *__restrict__ zflag, uint32_t *__restrict__ zl, uint32_t *__restrict__ nz) {
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  const bool z = v < n && zflag[v] != 0;
  const uint64_t b = __ballot(z);
  if (b == 0ull) return;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(nz, (uint32_t)__popcll(b));
  base = __builtin_amdgcn_readfirstlane(base);
  if (z) zl[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = v;
}

// packed words of the listed rows -> the row-major staging tables (the inverse of k_pack_full; one mask word)
template <int WB>
__global__ __launch_bounds__(256) void kr_unpack_rows(uint32_t n, uint32_t n_rows, const uint32_t *__restrict__ rows, const void *__restrict__ packed,
                                                      FusedParams P, uint32_t *__restrict__ dist, uint16_t *__restrict__ hops,
                                                      uint16_t *__restrict__ flags, uint64_t *__restrict__ mask) {
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  if (v >= n || blockIdx.y >= n_rows) return;
  const size_t idx = (size_t)rows[blockIdx.y] * n + v;
  const uint32_t mm = (1u << P.mbits) - 1u;
  if (WB == 8) {
    const uint64_t w = ((const uint64_t *)packed)[idx];
    const bool in = w != ~0ull;
    dist[idx] = in ? (uint32_t)(w >> 32) : INF;
    hops[idx] = in ? (uint16_t)(((uint32_t)w >> P.mbits) & P.hmax) : (uint16_t)0;
    flags[idx] = in ? 1 : 0;
    mask[idx] = in ? (uint64_t)((uint32_t)w & mm) : 0ull;
  } else {
    const uint32_t w = ((const uint32_t *)packed)[idx];
    const bool in = w < P.inf_t;
    dist[idx] = in ? (w >> P.sh) : INF;
    hops[idx] = in ? (uint16_t)((w >> P.mbits) & P.hmax) : (uint16_t)0;
    flags[idx] = in ? 1 : 0;
    mask[idx] = in ? (uint64_t)(w & mm) : 0ull;
  }
}

struct RepairArgs {
  GraphDev g;
  const uint32_t *root_list;   // [n_dyn] indices into roots[] (= the root's slot in the run's slot tables)
  const uint32_t *roots;
  uint32_t n_dyn, net_nexthops, ignore_ovl;
  SlotTabs tabs;
  uint32_t *dist; uint16_t *hops; uint16_t *flags; uint64_t *mask; uint32_t words;
  const uint32_t *row_map;
  const uint8_t *zflag;        // [n] 1 = has a zero-cost kept in-link
  const uint32_t *zl;          // [nz] those vertices
  const uint32_t *nz;          // device counter
  uint32_t *R, *pos;           // [n_dyn][n]; only entries of zl vertices are ever written or read
  uint32_t *stamp;             // [n_dyn][n], zero on entry: id of the last sweep a vertex was put on a worklist for
  uint32_t *wl;                // [n_dyn][2][n] worklists
  uint32_t *status;            // [n_dyn] RP_ST_FAIL | sweeps << 8;  [n_dyn + j]: vertices evaluated;  [2 n_dyn + j]: groups walked | largest << 16;
                               // [3 n_dyn + 8 j ..]: 100 MHz ticks of the phases (seeds, R, walks, first worklist, sweeps), nz
  uint32_t *pop_rank;          // may be null (HSPF_RUN_POP_RANK: written by kr_rank_fix)
};

struct RpCtx {
  const GraphDev &g;
  const uint32_t *D; const uint16_t *H; const uint64_t *M;
  const uint8_t *zflag; const uint32_t *R, *P;
  uint32_t root, ignore_ovl;
};

// the source of in-link `raw` counts for this root (overload gate, holo-isis/src/spf.rs:568-574)
__device__ __forceinline__ bool rp_src_ok(const RpCtx &c, uint32_t raw) {
  return c.ignore_ovl || !(raw & SRC_NO_TRANSIT) || (raw & SRC_MASK) == c.root;
}
// vertex m releases / feeds its out-neighbours (the same gate seen from the source; NO_EXPAND vertices keep no out-links)
__device__ __forceinline__ bool rp_expands(const RpCtx &c, uint32_t m) {
  const uint32_t vf = c.g.vflags[m];
  return m == c.root || (vf & 1u) || c.ignore_ovl || !(vf & 2u);
}
__device__ __forceinline__ bool rp_tight(uint32_t du, uint32_t w, uint32_t dv) {
  return du != INF && (uint64_t)du + w == (uint64_t)dv;            // (dv != INF; a saturated sum is never tight)
}
__device__ __forceinline__ uint64_t rp_ord(const RpCtx &c, uint32_t u) {     // (R, pos) of u; natural vertices: (u, 0)
  return c.zflag[u] ? (((uint64_t)c.R[u] << 32) | c.P[u]) : ((uint64_t)u << 32);
}

// hops and mask word q..: one vertex in the true order.  Returns false when v has no parent (cannot happen for a vertex of the SPT).
template <int WMAX>
__device__ __forceinline__ bool rp_eval(const RpCtx &c, const RepairArgs &a, uint32_t ri, uint32_t v, uint32_t W, uint32_t &nh, uint64_t (&nm)[WMAX]) {
  const GraphDev &g = c.g;
  const uint32_t dv = c.D[v];
  const uint64_t ov = rp_ord(c, v);
  const bool v_router = !(g.vflags[v] & 1u);
  uint32_t bd = INF, bhops = 0; uint64_t bo = ~0ull; bool have = false;
#pragma unroll
  for (int q = 0; q < WMAX; ++q) nm[q] = 0ull;
  for (uint32_t e = g.in_ptr[v], e1 = g.in_ptr[v + 1]; e < e1; ++e) {
    const uint32_t raw = g.in_src[e], w = g.in_w[e], u = raw & SRC_MASK;
    if (!rp_src_ok(c, raw)) continue;
    const uint32_t du = c.D[u];
    if (!rp_tight(du, w, dv)) continue;
    uint64_t ou = 0;
    if (w == 0u) { ou = rp_ord(c, u); if (ou >= ov) continue; }     // same level: a parent only if it is popped before v
    else if (c.zflag[u]) ou = rp_ord(c, u);
    else ou = (uint64_t)u << 32;
    const uint32_t hu = c.H[u];
    if (!have || du < bd || (du == bd && ou < bo)) { have = true; bd = du; bo = ou; bhops = hu; }
    if (hu == 0u) {                                                 // parent: the root or a hops-0 network -> the link's own slot
      if (v_router || a.net_nexthops) {
        const uint32_t base_s = (u == c.root) ? 0u : slot_base_of(a.tabs, ri, u);
        if (base_s != 0xFFFFFFFFu) {
          const uint32_t sidx = base_s + g.in_fpos[e];
          if ((sidx >> 6) < W && (sidx >> 6) < (uint32_t)WMAX) nm[sidx >> 6] |= 1ull << (sidx & 63u);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < WMAX; ++q) if ((uint32_t)q < W) nm[q] |= c.M[(size_t)u * W + q];
    }
  }
  nh = min(bhops + (v_router ? 1u : 0u), 0xFFFFu);                  // u16 saturating_add
  return have;
}

// One workgroup per dynamic root.  WMAX >= mask words of the run (1, 2, 4, 16).
template <int WMAX>
__global__ __launch_bounds__(RP_THREADS) void k_repair(RepairArgs a) {
  __shared__ uint32_t s_cnt[2], s_flag, s_fail, s_evals, s_groups, s_gmax;
  const uint32_t j = blockIdx.x, tid = threadIdx.x;
  const uint32_t ri = a.root_list[j];
  const uint32_t root = a.roots[ri];
  const GraphDev &g = a.g;
  const uint32_t n = g.n, W = a.words;
  const size_t orow = a.row_map ? a.row_map[ri] : ri;
  uint32_t *D = a.dist + orow * n;
  uint16_t *H = a.hops + orow * n;
  uint16_t *F = a.flags + orow * n;
  uint64_t *M = a.mask + orow * (size_t)n * W;
  uint32_t *R = a.R + (size_t)j * n, *P = a.pos + (size_t)j * n, *ST = a.stamp + (size_t)j * n;
  uint32_t *WL0 = a.wl + (size_t)j * 2u * n, *WL1 = WL0 + n;
  const uint32_t nz = *a.nz;
  const RpCtx c{g, D, H, M, a.zflag, R, P, root, a.ignore_ovl};
  if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_flag = 0; s_fail = 0; s_evals = 0; s_groups = 0; s_gmax = 0; }
  uint32_t *prof = a.status + 3u * a.n_dyn + 8u * j;
  uint64_t t_mark = wall_clock64();
  auto lap = [&](uint32_t k) { if (tid == 0) { const uint64_t t = wall_clock64(); prof[k] = (uint32_t)(t - t_mark); t_mark = t; } };
  if (tid == 0) prof[5] = nz;
  // ---- 1. seeds: R = own index; every other vertex of the list waits for a release path
  for (uint32_t i = tid; i < nz; i += RP_THREADS) {
    const uint32_t v = a.zl[i], dv = D[v];
    bool seed = dv == INF || v == root;
    if (!seed)
      for (uint32_t e = g.in_ptr[v], e1 = g.in_ptr[v + 1]; e < e1 && !seed; ++e) {
        const uint32_t raw = g.in_src[e], w = g.in_w[e];
        seed = w != 0u && rp_src_ok(c, raw) && rp_tight(D[raw & SRC_MASK], w, dv);
      }
    R[v] = seed ? v : RP_UNRES;
    P[v] = 0u;
  }
  __syncthreads();
  lap(0);
  // ---- 2. R = min over zero-cost tight parents of max(R(parent), own index): monotone, to the fixed point
  for (uint32_t it = 0;; ++it) {
    bool ch = false;
    for (uint32_t i = tid; i < nz; i += RP_THREADS) {
      const uint32_t v = a.zl[i], rv = R[v];
      if (rv == v) continue;                                         // a seed, or as low as it can get
      const uint32_t dv = D[v];
      uint32_t cand = rv;
      for (uint32_t e = g.in_ptr[v], e1 = g.in_ptr[v + 1]; e < e1; ++e) {
        const uint32_t raw = g.in_src[e], u = raw & SRC_MASK;
        if (g.in_w[e] != 0u || !rp_src_ok(c, raw) || D[u] != dv || u == v) continue;
        const uint32_t ru = a.zflag[u] ? R[u] : u;
        if (ru != RP_UNRES) cand = min(cand, max(ru, v));
      }
      if (cand < rv) { R[v] = cand; ch = true; }
    }
    if (ch) s_flag = 1;
    __syncthreads();
    const bool again = s_flag != 0;
    __syncthreads();
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (!again) break;
    if (it > 4u * n + 64u) { if (tid == 0) s_fail = 1; break; }
  }
  lap(1);
  // ---- 3. the groups: members of group y (R == y, other than y) in the order of a lowest-index-first walk from y
  for (uint32_t i = tid; i < nz; i += RP_THREADS) {
    const uint32_t v = a.zl[i], y = R[v], dv = D[v];
    if (y == v || dv == INF) continue;
    if (y == RP_UNRES) { s_fail = 1; continue; }                     // (in the SPT without a release path: cannot happen)
    if (!rp_expands(c, y)) continue;
    // the walk is done by the thread of y's lowest-numbered direct child in the group
    uint32_t minc = INF; bool direct = false;
    for (uint32_t k = g.out_ptr[y], k1 = g.out_ptr[y + 1]; k < k1; ++k) {
      const uint32_t x = g.out_dst[k];
      if (g.out_w[k] != 0u || x == y || D[x] != dv || !a.zflag[x] || R[x] != y) continue;
      minc = min(minc, x); direct = direct || x == v;
    }
    if (!direct || minc != v) continue;
    uint32_t heap[RP_HEAP]; uint32_t hn = 0, p = 0; bool fail = false;
    auto push = [&](uint32_t x) {
      if (P[x] != 0u) return;                                        // on the list or popped already (parallel links, several releasers)
      if (hn == RP_HEAP) { fail = true; return; }
      P[x] = RP_INHEAP;
      uint32_t q = hn++;
      while (q > 0 && heap[q - 1] < x) { heap[q] = heap[q - 1]; --q; }     // kept sorted, largest first: the pop is heap[--hn]
      heap[q] = x;
    };
    for (uint32_t k = g.out_ptr[y], k1 = g.out_ptr[y + 1]; k < k1; ++k) {
      const uint32_t x = g.out_dst[k];
      if (g.out_w[k] == 0u && x != y && D[x] == dv && a.zflag[x] && R[x] == y) push(x);
    }
    while (hn && !fail) {
      const uint32_t m = heap[--hn];
      P[m] = ++p;
      if (!rp_expands(c, m)) continue;
      for (uint32_t k = g.out_ptr[m], k1 = g.out_ptr[m + 1]; k < k1; ++k) {
        const uint32_t x = g.out_dst[k];
        if (g.out_w[k] == 0u && x != y && x != m && D[x] == dv && a.zflag[x] && R[x] == y) push(x);
      }
    }
    if (fail) s_fail = 1;
    atomicAdd(&s_groups, 1u); atomicMax(&s_gmax, p);
  }
  __syncthreads();
  lap(2);
  if (s_fail) {                                                       // nothing was written to the result rows: k_exact redoes the root
    if (tid == 0) { a.status[j] = RP_ST_FAIL; a.status[a.n_dyn + j] = 0; a.status[2u * a.n_dyn + j] = s_groups | (min(s_gmax, 0xFFFFu) << 16); }
    return;
  }
  // a member that no walk reached (its releaser does not expand, ...) cannot exist: every member has a release path from y
  // ---- 4. the rows are those of a root with a dynamic order (HSPF_RF_EXACT: "ask for pop_rank if the order matters");
  //         first worklist: zero-cost tight link from a higher-numbered source; tight children of non-natural vertices
  for (uint32_t v = tid; v < n; v += RP_THREADS) if (F[v] & 1u) F[v] = (uint16_t)(F[v] | 2u);
  auto wake = [&](uint32_t x, uint32_t sweep, uint32_t *list, uint32_t which) {
    if (x == root) return;
    if (atomicExch(&ST[x], sweep) == sweep) return;
    list[atomicAdd(&s_cnt[which], 1u)] = x;
  };
  for (uint32_t i = tid; i < nz; i += RP_THREADS) {
    const uint32_t v = a.zl[i], dv = D[v];
    if (dv == INF || v == root) continue;
    if (P[v] == RP_INHEAP) { s_fail = 1; continue; }
    bool due = false;
    for (uint32_t e = g.in_ptr[v], e1 = g.in_ptr[v + 1]; e < e1 && !due; ++e) {
      const uint32_t raw = g.in_src[e], u = raw & SRC_MASK;
      due = g.in_w[e] == 0u && u >= v && rp_src_ok(c, raw) && D[u] == dv;
    }
    if (due) wake(v, 1u, WL0, 0u);
    if (R[v] != v && rp_expands(c, v))
      for (uint32_t k = g.out_ptr[v], k1 = g.out_ptr[v + 1]; k < k1; ++k) {
        const uint32_t x = g.out_dst[k];
        if (rp_tight(dv, g.out_w[k], D[x]) && D[x] != INF) wake(x, 1u, WL0, 0u);
      }
  }
  __syncthreads();
  lap(3);
  // ---- 5. sweeps: evaluate the worklist in the true order; whatever changes wakes its tight children
  uint32_t sweep = 1u, cur = 0u, evals = 0u;
  for (;;) {
    const uint32_t cnt = s_cnt[cur];
    if (cnt == 0u || s_fail) break;
    const uint32_t *list = cur ? WL1 : WL0;
    uint32_t *next = cur ? WL0 : WL1;
    for (uint32_t i = tid; i < cnt; i += RP_THREADS) {
      const uint32_t v = list[i];
      uint32_t nh; uint64_t nm[WMAX];
      if (!rp_eval<WMAX>(c, a, ri, v, W, nh, nm)) { s_fail = 1; continue; }
      ++evals;
      bool ch = nh != H[v];
#pragma unroll
      for (int q = 0; q < WMAX; ++q) if ((uint32_t)q < W) ch = ch || nm[q] != M[(size_t)v * W + q];
      if (!ch) continue;
      H[v] = (uint16_t)nh;
#pragma unroll
      for (int q = 0; q < WMAX; ++q) if ((uint32_t)q < W) M[(size_t)v * W + q] = nm[q];
      if (!rp_expands(c, v)) continue;
      const uint32_t dv = D[v];
      for (uint32_t k = g.out_ptr[v], k1 = g.out_ptr[v + 1]; k < k1; ++k) {
        const uint32_t x = g.out_dst[k];
        if (D[x] != INF && rp_tight(dv, g.out_w[k], D[x])) wake(x, sweep + 1u, next, cur ^ 1u);
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) s_cnt[cur] = 0;
    cur ^= 1u; ++sweep;
    __syncthreads();
    if (sweep > 4u * n + 64u) { if (tid == 0) s_fail = 1; __syncthreads(); break; }
  }
  atomicAdd(&s_evals, evals);
  __syncthreads();
  lap(4);
  if (tid == 0) {
    a.status[j] = (s_fail ? RP_ST_FAIL : 0u) | (min(sweep - 1u, 0xFFFFFFu) << 8);
    a.status[a.n_dyn + j] = s_evals;
    a.status[2u * a.n_dyn + j] = s_groups | (min(s_gmax, 0xFFFFu) << 16);
  }
}

}  // namespace hspf
