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
// vertices, and from there whatever changes (worklist sweeps).
//
// Cost follows the number of zero-cost links and the size of the changed cones, not the size of the graph.  A root whose
// group does not fit the walk's fixed-size heap, whose R needs more than RP_RELAX rounds, or that has no hops / mask arrays to
// repair, still goes to k_exact.
//
// Shape (round 6, second form).  The first form was ONE launch, one workgroup of 1 024 threads per root: correct, and bound by
// the latency of its serial per-link loops on ONE compute unit per root — isis-100k with 1 % zero-cost links, 64 roots: 2.2 ms
// (6 700 evaluations per root, ~30 us each), a single root 1.6 ms with 255 CUs idle (profiles/r06_notes.md).  Now every phase
// is a chip-wide launch over (roots x items) and an item is worked on by a GROUP OF 16 LANES, one in-link per lane: the
// dependent loads of a row (record -> source's distance -> source's order / hops / mask) are three round trips per item instead
// of three per link.  Sweeps are launched ahead in chunks; a sweep whose predecessor woke nobody returns at once.
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
  // one atomic per BLOCK on the list's counter (one per wave: 1 600 of them on one word were 13 of this kernel's 19 us at 100 000
  // vertices); the order of the list does not matter to anyone
  __shared__ uint32_t s_wave[4], s_base;
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  const bool z = v < n && zflag[v] != 0;
  const uint64_t b = __ballot(z);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = t ? atomicAdd(nz, t) : 0u;
  }
  __syncthreads();
  uint32_t base = s_base;
  for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
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

constexpr uint32_t RP_RELAX = 4u;                // rounds of the R relaxation that are launched (a root still changing in the last one: k_exact)
constexpr uint32_t RP_MAX_SWEEPS = 4096u;        // worklist sweeps a repair may take
constexpr uint32_t RP_GX = 64u;                  // blocks per root of a phase launch (256 threads = 16 groups of 16 lanes)

// control block (u32 words, zero on entry).  Per-root counters sit RP_PAD words apart: atomics on one cache LINE retire one after
// the other whoever issues them (64 roots' counters side by side made the first-worklist launch 1 ms: profiles/r06_notes.md).
//   cnt[3][n_dyn] worklist lengths (sweep s reads s % 3, fills (s + 1) % 3, clears (s + 2) % 3) | nsc[n_dyn] vertices that wait for a
//   release path | fail[n_dyn] (1: R did not settle in the rounds launched, 2: anything else) | rlast[n_dyn] last relaxation round that
//   changed the root | pend[RP_MAX_SWEEPS + 2] "sweep s has work" | per root, RP_PAD apart: evaluations, groups, largest deep group
//   (one atomic per BLOCK on the root's own line: a per-thread atomic on one word made the first sweep 240 us instead of 80)
constexpr uint32_t RP_PAD = 32u;
struct RpCtl {
  uint32_t *base; uint32_t nd;
  __device__ __host__ uint32_t *cnt(uint32_t which, uint32_t j) const { return base + ((size_t)which * nd + j) * RP_PAD; }
  __device__ __host__ uint32_t *nsc(uint32_t j) const { return base + ((size_t)3u * nd + j) * RP_PAD; }
  __device__ __host__ uint32_t *fail() const { return base + (size_t)4u * nd * RP_PAD; }
  __device__ __host__ uint32_t *rlast() const { return fail() + nd; }
  __device__ __host__ uint32_t *pend() const { return fail() + 2u * (size_t)nd; }
  __device__ __host__ uint32_t *tot(uint32_t j) const { return pend() + RP_MAX_SWEEPS + 2u + (size_t)j * RP_PAD; }
  static size_t words(uint32_t nd) { return (size_t)5u * nd * RP_PAD + 2u * (size_t)nd + RP_MAX_SWEEPS + 2u + 4u; }
};

struct RepairArgs {
  GraphDev g;
  const uint32_t *root_list;   // [n_dyn] indices into roots[] (= the root's slot in the run's slot tables)
  const uint32_t *roots;
  uint32_t n_dyn, net_nexthops, ignore_ovl, xcd_map;
  SlotTabs tabs;
  uint32_t *dist; uint16_t *hops; uint16_t *flags; uint64_t *mask; uint32_t words;
  const uint32_t *row_map;
  const uint8_t *zflag;        // [n] 1 = has a zero-cost kept in-link
  const uint32_t *zl;          // [nz] those vertices
  const uint32_t *nz;          // device counter
  uint32_t *R, *pos;           // [n_dyn][n]; only entries of zl vertices are ever written or read
  uint32_t *stamp;             // [n_dyn][n], zero on entry: id of the last sweep a vertex was put on a worklist for
  uint32_t *wl;                // [n_dyn][2][n] worklists
  RpCtl ctl;
  // HSPF_REPAIR_TRACE=root:v0,v1,v2,v3 (debugging): evaluations of and wake-ups for these vertices of that root, recorded in order
  uint32_t *trace; uint32_t trace_root; uint32_t trace_v[4];
};
#define RP_TRACE_CAP 4096u
__device__ __forceinline__ bool rp_traced(const RepairArgs &a, uint32_t root, uint32_t v) {
  return a.trace && root == a.trace_root && (v == a.trace_v[0] || v == a.trace_v[1] || v == a.trace_v[2] || v == a.trace_v[3]);
}
__device__ __forceinline__ void rp_trace(const RepairArgs &a, uint32_t code, uint32_t x, uint32_t y, uint32_t z) {
  const uint32_t i = atomicAdd(a.trace, 1u);
  if (i < RP_TRACE_CAP) { uint32_t *t = a.trace + 4u + 4u * i; t[0] = code; t[1] = x; t[2] = y; t[3] = z; }
}

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
__device__ __forceinline__ bool rp_any16(bool p) {                 // over the 16 lanes of the group
  return ((__ballot(p) >> (threadIdx.x & 48u)) & 0xFFFFull) != 0ull;
}

// The per-root view of a phase kernel: block bx of root j works on the root's items bx, bx + gx, ... in groups of 16 lanes.
// Placement (a.xcd_map): consecutive workgroups go round-robin to the 8 XCDs, so with the plain (x, j) grid every root's blocks sat on
// every XCD and each of the eight L2s saw all roots' tables (64 x ~2.5 MB of random 4-byte reads: nothing stayed).  Mapped, workgroup
// L = x + gridDim.x * y belongs to XCD L % 8 and that XCD works through the roots j = L % 8, L % 8 + 8, ... one after the other,
// gridDim.x blocks each: a root's distances, hops, masks, order and stamps meet ONE L2, phase after phase (a kernel boundary does
// not empty it).  Speed only: any placement gives the same result.  The grid is (blocks per root, roots rounded up to 8).
struct RpRoot {
  uint32_t j, ri, root, n, W, sub, grp, ngrp, bx, gx; bool live;
  uint32_t *D; uint16_t *H, *F; uint64_t *M; uint32_t *R, *P, *ST, *WL0, *WL1;
  __device__ __forceinline__ RpRoot(const RepairArgs &a) {
    gx = gridDim.x;
    if (a.xcd_map) {
      const uint32_t L = blockIdx.x + gridDim.x * blockIdx.y, slot = L >> 3;
      j = (L & 7u) + 8u * (slot / gx); bx = slot % gx;
    } else { j = blockIdx.y; bx = blockIdx.x; }
    live = j < a.n_dyn;
    if (!live) j = 0u;
    ri = a.root_list[j]; root = a.roots[ri]; n = a.g.n; W = a.words;
    sub = threadIdx.x & 15u; grp = bx * (blockDim.x >> 4) + (threadIdx.x >> 4); ngrp = gx * (blockDim.x >> 4);
    const size_t orow = a.row_map ? a.row_map[ri] : ri;
    D = a.dist + orow * n; H = a.hops + orow * n; F = a.flags + orow * n; M = a.mask + orow * (size_t)n * W;
    R = a.R + (size_t)j * n; P = a.pos + (size_t)j * n; ST = a.stamp + (size_t)j * n;
    WL0 = a.wl + (size_t)j * 2u * n; WL1 = WL0 + n;
  }
  __device__ __forceinline__ RpCtx ctx(const RepairArgs &a) const { return RpCtx{a.g, D, H, M, a.zflag, R, P, root, a.ignore_ovl}; }
};

// "This root has failed" read ONCE per block: other blocks of the root may raise the flag while this kernel runs, and the blocks
// end on block-wide barriers (rp_stat, __syncthreads_or) — a block whose threads disagreed on the early return would hang there.
__device__ __forceinline__ bool rp_failed(const RepairArgs &a, uint32_t j) {
  __shared__ uint32_t s_failed;
  if (threadIdx.x == 0) s_failed = a.ctl.fail()[j];
  __syncthreads();
  return s_failed != 0u;
}
// One statistics counter of the block's root: the threads' values summed (or maxed) in LDS, ONE global atomic per block.
__device__ __forceinline__ void rp_stat(uint32_t *word, uint32_t mine, bool is_max) {
  __shared__ uint32_t s_acc;
  if (threadIdx.x == 0) s_acc = 0u;
  __syncthreads();
  if (mine) { if (is_max) atomicMax(&s_acc, mine); else atomicAdd(&s_acc, mine); }
  __syncthreads();
  if (threadIdx.x == 0 && s_acc) { if (is_max) atomicMax(word, s_acc); else atomicAdd(word, s_acc); }
}
// Append x to a list of the block's root — every lane of the wave calls it (want = false: nothing to append): ONE atomic per wave.
__device__ __forceinline__ void rp_append(uint32_t *cnt, uint32_t *list, uint32_t x, bool want) {
  const uint64_t b = __ballot(want);
  if (b == 0ull) return;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == (uint32_t)__builtin_ctzll(b)) base = atomicAdd(cnt, (uint32_t)__popcll(b));
  base = (uint32_t)__shfl((int)base, __builtin_ctzll(b), 64);
  if (want) list[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = x;
}
// ... onto the worklist of `sweep`, once per sweep and vertex (the stamp decides who appends)
__device__ __forceinline__ bool rp_wake(const RpRoot &r, uint32_t *cnt, uint32_t *list, uint32_t x, uint32_t sweep, bool want,
                                        const RepairArgs *a = nullptr, uint32_t waker = 0) {
  const bool asked = want;
  want = want && x != r.root && atomicExch(&r.ST[x], sweep) != sweep;
  if (a && asked && rp_traced(*a, r.root, x)) rp_trace(*a, 0x10000000u | sweep, x, waker, want ? 1u : 0u);
  rp_append(cnt, list, x, want);
  return want;
}

// ---- 1. seeds: R = own index; every other vertex of the list waits for a release path.  Also: HSPF_RF_EXACT on the root's rows.
__global__ __launch_bounds__(256) void kr_seed(RepairArgs a) {
  const RpRoot r(a);
  if (!r.live) return;
  const GraphDev &g = a.g;
  const RpCtx c = r.ctx(a);
  // (the rows are those of a root with a dynamic order: "ask for pop_rank if the order matters")
  for (uint32_t v = r.bx * blockDim.x + threadIdx.x; v < r.n; v += r.gx * blockDim.x) if (r.F[v] & 1u) r.F[v] = (uint16_t)(r.F[v] | 2u);
  const uint32_t nz = *a.nz;
  for (uint32_t i = r.grp; i < nz; i += r.ngrp) {
    const uint32_t v = a.zl[i], dv = r.D[v];
    bool seed = false;
    if (dv != INF && v != r.root)
      for (uint32_t e = g.in_ptr[v] + r.sub, e1 = g.in_ptr[v + 1]; rp_any16(e < e1); e += 16u) {
        if (e < e1) {
          const uint32_t raw = g.in_src[e], w = g.in_w[e];
          seed = seed || (w != 0u && rp_src_ok(c, raw) && rp_tight(r.D[raw & SRC_MASK], w, dv));
        }
      }
    seed = rp_any16(seed) || dv == INF || v == r.root;
    if (r.sub == 0u) { r.R[v] = seed ? v : RP_UNRES; r.P[v] = 0u; }
    rp_append(a.ctl.nsc(r.j), r.WL0, v, r.sub == 0u && !seed);       // the only vertices the next two phases look at (WL0 is free until sweep 1)
  }
}

// ---- 2. R = min over zero-cost tight parents of max(R(parent), own index): monotone, to the fixed point (one round per launch)
__global__ __launch_bounds__(256) void kr_relax(RepairArgs a, uint32_t round) {
  const RpRoot r(a);
  if (!r.live) return;
  uint32_t *rlast = a.ctl.rlast();
  if (round > 0u && rlast[r.j] + 1u < round) return;               // this root stopped changing (uniform per block)
  const GraphDev &g = a.g;
  const RpCtx c = r.ctx(a);
  const uint32_t ns = *a.ctl.nsc(r.j);
  bool ch = false;
  for (uint32_t i = r.grp; i < ns; i += r.ngrp) {
    const uint32_t v = r.WL0[i], rv = r.R[v];
    if (rv == v) continue;                                           // as low as it can get (uniform in the group)
    const uint32_t dv = r.D[v];
    uint32_t cand = rv;
    for (uint32_t e = g.in_ptr[v] + r.sub, e1 = g.in_ptr[v + 1]; rp_any16(e < e1); e += 16u) {
      if (e < e1) {
        const uint32_t raw = g.in_src[e], u = raw & SRC_MASK;
        if (g.in_w[e] == 0u && rp_src_ok(c, raw) && u != v && r.D[u] == dv) {
          const uint32_t ru = a.zflag[u] ? r.R[u] : u;
          if (ru != RP_UNRES) cand = min(cand, max(ru, v));
        }
      }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) cand = min(cand, (uint32_t)__shfl_xor((int)cand, off, 16));
    if (cand < rv) { if (r.sub == 0u) r.R[v] = cand; ch = true; }
  }
  if (__syncthreads_or(ch ? 1 : 0) && threadIdx.x == 0) rlast[r.j] = round;     // (one store per block)
}

// ---- 3. the groups: members of group y (R == y, other than y) in the order of a lowest-index-first walk from y.
// Nearly every group is FLAT — y's direct zero-cost children, none of which releases a member of its own —, and then the walk is
// just the children in index order: pos = 1 + the number of links from y to lower-numbered direct children, which every member counts for itself
// in y's out-row (16 lanes, one out-link each; kr_walks).  A member that is NOT a direct child of y marks the group (stamp of y:
// free until the first worklist is made) and kr_walks_deep redoes such a group with the real walk, one thread per group.
constexpr uint32_t RP_DEEP = 0xFFFFFFFEu;
__global__ __launch_bounds__(256) void kr_walks(RepairArgs a, uint32_t rounds) {
  const RpRoot r(a);
  if (!r.live) return;
  const GraphDev &g = a.g;
  const RpCtx c = r.ctx(a);
  if (a.ctl.rlast()[r.j] + 1u >= rounds) { if (r.bx == 0 && threadIdx.x == 0) a.ctl.fail()[r.j] = 1u; return; }   // R did not settle in the rounds launched
  const uint32_t ns = *a.ctl.nsc(r.j);
  uint32_t groups = 0;
  for (uint32_t i = r.grp; i < ns; i += r.ngrp) {
    const uint32_t v = r.WL0[i], y = r.R[v], dv = r.D[v];
    if (y == v) continue;                                            // released in index order: natural
    if (y == RP_UNRES) { if (r.sub == 0u) a.ctl.fail()[r.j] = 2u; continue; }   // (in the SPT without a release path: cannot happen)
    uint32_t below = 0; bool direct = false;
    if (rp_expands(c, y))
      for (uint32_t k = g.out_ptr[y] + r.sub, k1 = g.out_ptr[y + 1]; rp_any16(k < k1); k += 16u) {
        if (k < k1) {
          const uint32_t x = g.out_dst[k];
          if (g.out_w[k] == 0u && x != y && r.D[x] == dv && a.zflag[x] && r.R[x] == y) {
            direct = direct || x == v;
            below += x < v ? 1u : 0u;                                // (parallel links count twice: pos only has to ORDER the members)
          }
        }
      }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) below += (uint32_t)__shfl_xor((int)below, off, 16);
    direct = rp_any16(direct);
    if (r.sub != 0u) continue;
    if (direct) { r.P[v] = below + 1u; if (below == 0u) ++groups; }
    else r.ST[y] = RP_DEEP;                                           // released by another member: the group needs the real walk
  }
  rp_stat(a.ctl.tot(r.j) + 1, groups, false);
}
__global__ __launch_bounds__(256) void kr_walks_deep(RepairArgs a) {
  const RpRoot r(a);
  if (!r.live) return;
  const GraphDev &g = a.g;
  const RpCtx c = r.ctx(a);
  if (rp_failed(a, r.j)) return;
  const uint32_t ns = *a.ctl.nsc(r.j);
  uint32_t gmax = 0;
  for (uint32_t i = r.bx * blockDim.x + threadIdx.x; i < ns; i += r.gx * blockDim.x) {
    const uint32_t v = r.WL0[i], y = r.R[v], dv = r.D[v];
    if (y == v || y == RP_UNRES || r.P[v] != 1u || r.ST[y] != RP_DEEP) continue;     // the walk: the thread of y's lowest direct child
    const uint32_t k0 = g.out_ptr[y], k1 = g.out_ptr[y + 1];
    for (uint32_t k = k0; k < k1; ++k) {                               // the flat positions are void
      const uint32_t x = g.out_dst[k];
      if (g.out_w[k] == 0u && x != y && r.D[x] == dv && a.zflag[x] && r.R[x] == y) r.P[x] = 0u;
    }
    uint32_t heap[RP_HEAP]; uint32_t hn = 0, p = 0; bool fail = false;
    auto push = [&](uint32_t x) {
      if (r.P[x] != 0u) return;                                      // on the list or popped already (parallel links, several releasers)
      if (hn == RP_HEAP) { fail = true; return; }
      r.P[x] = RP_INHEAP;
      uint32_t q = hn++;
      while (q > 0 && heap[q - 1] < x) { heap[q] = heap[q - 1]; --q; }     // kept sorted, largest first: the pop is heap[--hn]
      heap[q] = x;
    };
    for (uint32_t k = k0; k < k1; ++k) {
      const uint32_t x = g.out_dst[k];
      if (g.out_w[k] == 0u && x != y && r.D[x] == dv && a.zflag[x] && r.R[x] == y) push(x);
    }
    while (hn && !fail) {
      const uint32_t m = heap[--hn];
      r.P[m] = ++p;
      if (!rp_expands(c, m)) continue;
      for (uint32_t k = g.out_ptr[m], ke = g.out_ptr[m + 1]; k < ke; ++k) {
        const uint32_t x = g.out_dst[k];
        if (g.out_w[k] == 0u && x != y && x != m && r.D[x] == dv && a.zflag[x] && r.R[x] == y) push(x);
      }
    }
    if (fail) a.ctl.fail()[r.j] = 2u;
    gmax = max(gmax, p);
  }
  rp_stat(a.ctl.tot(r.j) + 2, gmax, true);
}

// ---- 4. first worklist: a zero-cost tight link from a higher-numbered source; the tight children of non-natural vertices
__global__ __launch_bounds__(256) void kr_due(RepairArgs a) {
  const RpRoot r(a);
  if (!r.live) return;
  const GraphDev &g = a.g;
  const RpCtx c = r.ctx(a);
  if (rp_failed(a, r.j)) return;
  const uint32_t nz = *a.nz;
  uint32_t *cnt = a.ctl.cnt(1u, r.j);                                // sweep 1 reads counter 1, list 1
  bool any = false;
  for (uint32_t i = r.grp; i < nz; i += r.ngrp) {
    const uint32_t v = a.zl[i], dv = r.D[v];
    if (dv == INF || v == r.root) continue;
    const uint32_t rv = r.R[v];
    if (rv != v && (r.P[v] == 0u || r.P[v] == RP_INHEAP)) { if (r.sub == 0u) a.ctl.fail()[r.j] = 2u; continue; }   // a member no walk reached: cannot happen
    bool due = false;
    for (uint32_t e = g.in_ptr[v] + r.sub, e1 = g.in_ptr[v + 1]; rp_any16(e < e1); e += 16u) {
      if (e < e1) {
        const uint32_t raw = g.in_src[e], u = raw & SRC_MASK;
        due = due || (g.in_w[e] == 0u && u >= v && rp_src_ok(c, raw) && r.D[u] == dv);
      }
    }
    any = rp_wake(r, cnt, r.WL1, v, 1u, rp_any16(due) && r.sub == 0u, &a, 0xFFFFFFFFu) || any;
    if (rv != v && rp_expands(c, v))
      for (uint32_t k = g.out_ptr[v] + r.sub, k1 = g.out_ptr[v + 1]; rp_any16(k < k1); k += 16u) {
        const uint32_t x = k < k1 ? g.out_dst[k] : v;
        const bool t = k < k1 && r.D[x] != INF && rp_tight(dv, g.out_w[k], r.D[x]);
        any = rp_wake(r, cnt, r.WL1, x, 1u, t, &a, v) || any;
      }
  }
  if (__syncthreads_or(any ? 1 : 0) && threadIdx.x == 0) a.ctl.pend()[1] = 1u;
}

// ---- 5. one sweep: evaluate the worklist in the true order; whatever changes wakes its tight children
// hops and mask of v: 16 lanes, one in-link each.  `have` false: v has no parent (cannot happen for a vertex of the SPT).
template <int WMAX>
__global__ __launch_bounds__(256) void kr_sweep(RepairArgs a, uint32_t sweep) {
  uint32_t *pend = a.ctl.pend();
  if (pend[sweep] == 0u) return;                                     // the sweep before woke nobody: the repair is over
  const RpRoot r(a);
  if (!r.live) return;
  const GraphDev &g = a.g;
  const RpCtx c = r.ctx(a);
  const uint32_t W = r.W;
  uint32_t *cnt_cur = a.ctl.cnt(sweep % 3u, r.j), *cnt_next = a.ctl.cnt((sweep + 1u) % 3u, r.j);
  if (r.bx == 0 && threadIdx.x == 0) *a.ctl.cnt((sweep + 2u) % 3u, r.j) = 0u;   // read by the sweep before, filled by the next one
  if (rp_failed(a, r.j)) return;
  const uint32_t cnt = *cnt_cur;
  const uint32_t *list = (sweep & 1u) ? r.WL1 : r.WL0;
  uint32_t *next = (sweep & 1u) ? r.WL0 : r.WL1;
  bool any = false; uint32_t evals = 0;
  for (uint32_t i = r.grp; i < cnt; i += r.ngrp) {
    const uint32_t v = list[i];
    const uint32_t dv = r.D[v];
    const uint64_t ov = rp_ord(c, v);
    const bool v_router = !(g.vflags[v] & 1u);
    uint32_t bd = INF, bh = 0; uint64_t bo = ~0ull; bool have = false;
    uint64_t nm[WMAX];
#pragma unroll
    for (int q = 0; q < WMAX; ++q) nm[q] = 0ull;
    for (uint32_t e = g.in_ptr[v] + r.sub, e1 = g.in_ptr[v + 1]; rp_any16(e < e1); e += 16u) {
      if (e >= e1) continue;
      const uint32_t raw = g.in_src[e], w = g.in_w[e], u = raw & SRC_MASK;
      if (!rp_src_ok(c, raw)) continue;
      const uint32_t du = r.D[u];
      if (!rp_tight(du, w, dv)) continue;
      const uint64_t ou = rp_ord(c, u);
      if (w == 0u && ou >= ov) continue;                              // same level: a parent only if it is popped before v
      const uint32_t hu = r.H[u];
      if (!have || du < bd || (du == bd && (ou < bo || (ou == bo && hu < bh)))) { have = true; bd = du; bo = ou; bh = hu; }
      if (hu == 0u) {                                                 // parent: the root or a hops-0 network -> the link's own slot
        if (v_router || a.net_nexthops) {
          const uint32_t base_s = (u == r.root) ? 0u : slot_base_of(a.tabs, r.ri, u);
          if (base_s != 0xFFFFFFFFu) {
            const uint32_t sidx = base_s + g.in_fpos[e];
            if ((sidx >> 6) < W && (sidx >> 6) < (uint32_t)WMAX) nm[sidx >> 6] |= 1ull << (sidx & 63u);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < WMAX; ++q) if ((uint32_t)q < W) nm[q] |= r.M[(size_t)u * W + q];
      }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {                          // the first parent in the order (dist, R, pos); the union of the masks
      const uint32_t obd = (uint32_t)__shfl_xor((int)bd, off, 16), obh = (uint32_t)__shfl_xor((int)bh, off, 16);
      const uint64_t obo = (uint64_t)__shfl_xor((long long)bo, off, 16);
      const bool ohv = __shfl_xor(have ? 1 : 0, off, 16) != 0;
      // (dist, order, hops) as ONE total order: two lanes can hold the same parent with different hop counts — parallel links
      // from a parent that another group is re-evaluating in this very sweep, read in different trips of the loop above — and
      // with the hops left out of the comparison each of them kept its own: the group then disagreed on `ch` below, only the
      // lanes that saw a change went on to wake the children, and a child behind one of the other lanes' links was never
      // woken (round 6, tools/debug/dyn_fuzz_seed.py 487107: one wrong hop count in ~40 runs with three processes on the GPU).
      if (ohv && (!have || obd < bd || (obd == bd && (obo < bo || (obo == bo && obh < bh))))) { have = true; bd = obd; bo = obo; bh = obh; }
#pragma unroll
      for (int q = 0; q < WMAX; ++q) nm[q] |= (uint64_t)__shfl_xor((long long)nm[q], off, 16);
    }
    bh = (uint32_t)__shfl((int)bh, 0, 16);                          // (the group's lane 0 decides: `ch` and what is stored must be one value)
    if (!have) { if (r.sub == 0u) a.ctl.fail()[r.j] = 1u; continue; }
    if (r.sub == 0u) ++evals;
    const uint32_t nh = min(bh + (v_router ? 1u : 0u), 0xFFFFu);     // u16 saturating_add
    bool ch = nh != r.H[v];
#pragma unroll
    for (int q = 0; q < WMAX; ++q) if ((uint32_t)q < W) ch = ch || nm[q] != r.M[(size_t)v * W + q];
    if (r.sub == 0u && rp_traced(a, r.root, v)) rp_trace(a, 0x20000000u | sweep, v, (bh << 16) | (nh & 0xFFFFu), ((uint32_t)r.H[v] << 16) | (ch ? 1u : 0u) | ((uint32_t)(bo >> 32) << 1 & 0xFFFEu));
    if (!ch) continue;                                                // (uniform in the group: every lane compared the same values)
    // every lane has read the old values: lane 0 stores, the group wakes the tight children up
    if (r.sub == 0u) {
      r.H[v] = (uint16_t)nh;
#pragma unroll
      for (int q = 0; q < WMAX; ++q) if ((uint32_t)q < W) r.M[(size_t)v * W + q] = nm[q];
    }
    if (!rp_expands(c, v)) continue;
    for (uint32_t k = g.out_ptr[v] + r.sub, k1 = g.out_ptr[v + 1]; rp_any16(k < k1); k += 16u) {
      const uint32_t x = k < k1 ? g.out_dst[k] : v;
      const bool t = k < k1 && r.D[x] != INF && rp_tight(dv, g.out_w[k], r.D[x]);
      any = rp_wake(r, cnt_next, next, x, sweep + 1u, t, &a, v) || any;
    }
  }
  if (__syncthreads_or(any ? 1 : 0) && threadIdx.x == 0) pend[sweep + 1u] = 1u;
  rp_stat(a.ctl.tot(r.j), evals, false);
}


// ---- pop_rank (HSPF_RUN_POP_RANK) of the repaired roots: the position of every vertex in the order of the keys (dist, R, pos).
// Two stable device-wide radix sorts (hub_sort.h): by R, then by (root, dist); the few runs of equal (dist, R) — a group and the
// vertex that released it — are put in pos order by their own members.  Vertices off the SPT: 0xFFFFFFFF.
__global__ __launch_bounds__(256) void kr_rank_keys1(RepairArgs a, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t j = blockIdx.y, n = a.g.n;
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  if (v >= n) return;
  const size_t item = (size_t)j * n + v;
  keys[item] = a.zflag[v] ? a.R[item] : v;
  vals[item] = (uint32_t)item;
}
__global__ __launch_bounds__(256) void kr_rank_keys2(RepairArgs a, const uint32_t *__restrict__ items, uint64_t *__restrict__ keys, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const uint32_t item = items[i], n = a.g.n, j = item / n, v = item - j * n;
  const uint32_t ri = a.root_list[j];
  const size_t orow = a.row_map ? a.row_map[ri] : ri;
  keys[i] = ((uint64_t)j << 32) | a.dist[orow * n + v];
}
__global__ __launch_bounds__(256) void kr_rank_write(RepairArgs a, const uint32_t *__restrict__ items, uint32_t *__restrict__ pop_rank, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const uint32_t n = a.g.n, item = items[i], j = item / n, v = item - j * n;
  const uint32_t ri = a.root_list[j];
  const size_t orow = a.row_map ? a.row_map[ri] : ri;
  const uint32_t *D = a.dist + orow * n, *R = a.R + (size_t)j * n, *P = a.pos + (size_t)j * n;
  uint32_t *out = pop_rank + orow * n;
  const uint32_t dv = D[v];
  if (dv == INF) { out[v] = INF; return; }
  const uint32_t rv = a.zflag[v] ? R[v] : v, pv = a.zflag[v] ? P[v] : 0u;
  // the run of equal (dist, R) around i: the members before me in the run, and those of them with a smaller pos
  uint32_t before = 0, smaller = 0;
  const size_t row0 = (size_t)j * n;
  for (size_t k = i; k > row0; --k) {
    const uint32_t x = items[k - 1] - j * n;
    if (D[x] != dv || (a.zflag[x] ? R[x] : x) != rv) break;
    ++before; smaller += (a.zflag[x] ? P[x] : 0u) < pv ? 1u : 0u;
  }
  for (size_t k = i + 1; k < row0 + n; ++k) {
    const uint32_t x = items[k] - j * n;
    if (D[x] != dv || (a.zflag[x] ? R[x] : x) != rv) break;
    smaller += (a.zflag[x] ? P[x] : 0u) < pv ? 1u : 0u;
  }
  out[v] = (uint32_t)(i - row0) - before + smaller;
}

}  // namespace hspf
