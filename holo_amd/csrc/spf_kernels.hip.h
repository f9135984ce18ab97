// spf_kernels.hip.h — gfx950 (CDNA4) kernels of the batched SPF engine.
//
// Layout ("lane = root"): one wavefront carries 64 SPF roots of the same graph.  All per-run
// state is [batch][vertex][64 lanes], so the 64 roots' values of one vertex are one 256-byte row:
// every access to a neighbour's state is a single fully coalesced wave load, the link record
// (source, cost) is wave-uniform and comes through the scalar cache, and no atomics are needed
// because the wave that owns vertex v is the only writer of row v (pull / in-edge formulation).
// No MFMA: this is an irregular integer path bound by cache / HBM bandwidth.
//
// Semantics restated on device (see include/holo_spf_hip.h for the reference file:line map):
//   k_relax   label-correcting fixed point of   dist[v] = min over in-links (u,w) of dist[u] (+) w
//             under the reference's gates (holo-isis/src/spf.rs:557-604, 637-647).
//   k_dag     hops (first-discoverer rule, holo-ospf/src/spf.rs:700-703 / holo-isis :674-676) and
//             the ECMP first-hop set (holo-ospf/src/spf.rs:733-767, holo-isis/src/spf.rs:680-704)
//             over the tight-edge DAG, valid when the reference's pop order equals the static
//             (distance, vertex) order; every root for which that is not provable is flagged and
//             re-done by k_exact.
//   k_exact   one lane = one root, literal sequential restatement of the reference loop with a
//             binary heap in HBM (slow; only for roots with zero-cost plateaus that make the pop
//             order dynamic, or u32 saturation).
//   k_emit    LDS-tiled transpose of the lane-major state into the row-major [root][vertex]
//             result arrays of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace hspf {

constexpr uint32_t INF = 0xFFFFFFFFu;
constexpr uint32_t SRC_NO_TRANSIT = 0x80000000u;   // bit31 of in_src: source has the overload bit
constexpr uint32_t SRC_LEAF = 0x40000000u;         // bit30 of in_src: source is a leaf (GraphDev::leaf); read by k_fw<.., LEAF> only
constexpr uint32_t SRC_MASK = 0x3FFFFFFFu;

// hv word: [15:0] hops, [31:16] epoch.  epoch == 0: not final yet; epoch == e: became final in the
// DAG launch with epoch e.  A reader in launch E only trusts rows with 0 < epoch < E, i.e. rows
// finalised by an EARLIER launch: the row's mask words were written by that launch too and are
// visible after the kernel boundary, so no in-kernel release/acquire is needed between the two
// stores (MI355X_MICROARCH.md: per-XCD L2s are not coherent inside a launch).
constexpr uint32_t HV_EPOCH_SHIFT = 16;
constexpr uint32_t HV_EPOCH_MAX = 0xFFFFu;

// per-root (lane) status bits
constexpr uint32_t LF_NEED_EXACT = 1u;   // static pop order not provable / saturation: use k_exact
constexpr uint32_t LF_OVERFLOW = 2u;     // narrow fused state could not hold a value: redo the run wide
constexpr uint32_t LF_DYN = 4u;          // the pop order of this root is dynamic (a vertex whose only way in is a zero-cost link from a
                                         // higher-numbered source): distances are final, hops / masks are repaired by k_repair (spf_repair.hip.h)

constexpr int WAVES_PER_BLOCK = 4;
constexpr int VPW = 4;                    // vertices per wave per block
constexpr int VPB = WAVES_PER_BLOCK * VPW;

struct GraphDev {
  uint32_t n;            // vertices
  uint32_t e_in;         // kept in-links
  const uint32_t *in_ptr;   // [n+1]  transposed CSR of the links that survive the two-way check
  const uint32_t *in_src;   // [e_in] source vertex | SRC_NO_TRANSIT
  const uint32_t *in_w;     // [e_in] cost
  const uint32_t *in_fpos;  // [e_in] position of the link inside its source row (for slots)
  const uint8_t *vflags;    // [n]
  const uint8_t *rowflags;  // [n] RF_* : static reasons why a row needs the general fused routine
  // Leaves: a vertex with exactly one kept in-link whose kept out-links (at most one) lead back to that link's source — a
  // single-homed host, a stub LAN's pseudonode.  No shortest path runs THROUGH a leaf, so it takes no part in the fixed
  // point: its row is evaluated once, from its neighbour's final state, inside the emit (k_emit<W, true>), and a link
  // FROM a leaf only counts in the lane whose root the leaf is (k_fw<.., LEAF>).  The definition does not depend on costs.
  const uint8_t *leaf;      // [n] 1 = leaf
  // zcyc[v] = 1: v may lie on a CYCLE of zero-cost kept links (it survives the rounds of kb_zc_round, graph_build.hip.h);
  // null: the graph has no zero-cost link from a higher- or equal-numbered source.  A zero-cost link u -> v, u >= v, is SAFE
  // unless zcyc[u] && zcyc[v]: a safe one may feed v's hops / mask when it is v's only way in (finish_row_z) without ever
  // closing a cycle of dependencies among the rows of a sweep.
  const uint8_t *zcyc;
  // forward CSR (kept links only) for k_exact
  const uint32_t *out_ptr;  // [n+1]
  const uint32_t *out_dst;  // [e_in]
  const uint32_t *out_w;    // [e_in]
  const uint32_t *out_fpos; // [e_in]
  // XCD x (blocks with blockIdx.x % 8 == x, MI355X_MICROARCH.md) owns the 16-vertex chunks xcd_start[x] ..
  // xcd_start[x+1]: contiguous, so that the rows a wave touches stay in that XCD's L2, and cut at equal WORK
  // (in-links + a per-row constant), not equal vertex counts: in a fat-tree all 12 500 hundred-link switch rows come
  // first in VertexId order and would otherwise all land on XCD 0.
  uint32_t xcd_start[9];
  // Work units (null: a unit is a 16-vertex chunk and xcd_start is all there is).  A wave walks the in-links of its rows
  // one after the other in groups of a few requests, so a chunk of 100-link rows (fat-tree switches, LAN pseudonodes) is
  // one long latency chain on four waves while the rest of the chip idles.  Graphs with such chunks are cut into HEAVY
  // units — a chunk that holds a row of more than UNIT_HEAVY_DEG in-links becomes four units of 4 vertices, ONE row per
  // wave — and normal units (16 vertices, four rows per wave): unit_first = [heavy units | normal units], each class in
  // vertex order, unit_first[u] = first vertex | UNIT_SPLIT.  Every XCD gets an equal, contiguous share of BOTH classes
  // (xcd_heavy / xcd_start index into their class) and runs its heavy units first: one cost model cannot balance rows
  // that are bound by their link count against rows that are bound by the fixed cost of a wave (measured, fat-tree:
  // every weighting of a single range loses to the even split; profiles/r02_notes.md r02m).
  const uint32_t *unit_first;
  uint32_t n_heavy_units;
  uint32_t xcd_heavy[9];
  // Giant rows (more than GIANT_DEG in-links: the pseudonode of a LAN with hundreds or thousands of routers).  Even one
  // row per wave is then a chain of hundreds of dependent round trips per visit (isis-100k plus ONE 1 000-router LAN: 3.1
  // instead of 0.9 ms per 64-root run, 5 000 routers: 12 ms).  Such a row is cut into slices of GIANT_SLICE links;
  // k_giant_part evaluates the slices of the due giant rows in parallel before every sweep (one workgroup per slice) and
  // the sweep's own visit of the row merges the slices' accumulators in row order instead of walking the links.
  const uint32_t *giant_vtx;     // [n_giant] ascending
  const uint32_t *giant_slice0;  // [n_giant + 1] first slice of every giant row; [n_giant] = number of slices
  uint32_t n_giant;
};
constexpr uint32_t UNIT_SPLIT = 0x80000000u;
constexpr uint32_t UNIT_HEAVY_DEG = 32u;
constexpr uint32_t GIANT_DEG = 256u;
constexpr uint32_t GIANT_SLICE = 256u;          // links per slice: one workgroup, 64 links per wave
constexpr uint32_t GIANT_WORDS = 8u;            // accumulator words per lane and slice (AnyAcc)

struct OutDev {
  uint32_t *dist; uint16_t *hops; uint16_t *flags; uint64_t *mask; uint32_t out_words;
  const uint32_t *row_map;   // output row of the run's root r (null: r itself) — runs regrouped by state class
  void *packed;              // hspf_run_packed (ABI 7): the fused emit writes the state WORD itself, row-major, instead of the four arrays
  __device__ __forceinline__ size_t row(uint32_t r) const { return row_map ? row_map[r] : r; }
};

// rowflags bits (set at upload); RF_HNB is added per batch by k_init_fused
constexpr uint32_t RF_MANY = 1u;      // more than 16 kept in-links
constexpr uint32_t RF_NT = 2u;        // an in-link from an overloaded (no-transit) source
constexpr uint32_t RF_ZERO = 4u;      // a zero-cost in-link from a higher- or equal-numbered source
constexpr uint32_t RF_HNB = 8u;       // an in-neighbour that can have hops == 0 for some root of the batch
constexpr uint32_t RF_HNBN = 32u;     // ... and not only because that in-neighbour IS a root: a row behind a hops-0 network (k_init_fused)
constexpr uint32_t RF_ROOT = 64u;     // the row of a root of the batch (its own lane stays 0)
constexpr uint32_t RF_GIANT = 16u;    // more than GIANT_DEG kept in-links: the row is evaluated in slices (k_giant_part)

struct SlotTabs {           // per root: H vertices and their slot bases (include/holo_spf_hip.h)
  const uint32_t *ptr;      // [n_root_slots+1]
  const uint32_t *vtx;
  const uint32_t *base;
};

// The dispatcher places consecutive workgroups round-robin on the 8 XCDs (block b -> XCD b%8,
// MI355X_MICROARCH.md "Workgroup dispatch").  Give every XCD one contiguous eighth of the vertex
// range so that the rows a wave touches (its neighbours in a spatially numbered LSDB) stay in
// that XCD's private 4 MiB L2.  Speed only: any placement gives the same result.
// Returns the chunk of block bx, or 0xFFFFFFFF when this XCD's range is shorter than the grid's share (the grid is
// 8 x the longest range).
__device__ __forceinline__ uint32_t xcd_chunk(const uint32_t *xcd_start, uint32_t bx) {
  const uint32_t x = bx & 7u, j = bx >> 3;
  const uint32_t s0 = xcd_start[x], s1 = xcd_start[x + 1];
  return j < s1 - s0 ? s0 + j : 0xFFFFFFFFu;
}

// Rows of wave `wave` of block bx: first vertex and how many consecutive rows (4, or 1 in a heavy unit); false = nothing.
__device__ __forceinline__ bool wave_rows(const GraphDev &g, uint32_t bx, uint32_t wave, uint32_t &vbeg, uint32_t &nrows) {
  if (!g.unit_first) {
    const uint32_t u = xcd_chunk(g.xcd_start, bx);
    if (u == 0xFFFFFFFFu) return false;
    nrows = (uint32_t)VPW;
    vbeg = u * (uint32_t)VPB + wave * (uint32_t)VPW;
    return vbeg < g.n;
  }
  const uint32_t x = bx & 7u;
  uint32_t j = bx >> 3, u;
  const uint32_t h0 = g.xcd_heavy[x], nh = g.xcd_heavy[x + 1] - h0;
  if (j < nh) u = h0 + j;
  else {
    j -= nh;
    const uint32_t s0 = g.xcd_start[x];
    if (j >= g.xcd_start[x + 1] - s0) return false;
    u = g.n_heavy_units + s0 + j;
  }
  const uint32_t uf = g.unit_first[u];
  nrows = (uf & UNIT_SPLIT) ? 1u : (uint32_t)VPW;
  vbeg = (uf & ~UNIT_SPLIT) + wave * nrows;
  return vbeg < g.n;
}

__device__ __forceinline__ uint32_t slot_base_of(const SlotTabs &t, uint32_t root_slot, uint32_t u) {
  const uint32_t a = t.ptr[root_slot], b = t.ptr[root_slot + 1];
  for (uint32_t i = a; i < b; ++i)
    if (t.vtx[i] == u) return t.base[i];
  return 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------
// init: dist = INF except dist[root] = 0; hv = 0; mask = 0 are memsets; this sets the roots.
__global__ void k_init_roots(uint32_t n, uint32_t *dist, const uint32_t *roots, uint32_t n_lanes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  const uint32_t r = roots[i];
  if (r == INF) return;
  const uint32_t batch = i >> 6, lane = i & 63;
  dist[((size_t)batch * n + r) * 64 + lane] = 0;
}

// ---------------------------------------------------------------------------------------------
// Inner-loop design (measured, profiles/r01_pmc_notes.md): a CU has ONE scalar ALU for its four
// SIMDs, and a first version that kept the wave-uniform link data (source, cost, addresses,
// bounds) on the scalar side was issue-bound on it (1350 SALU + 1010 VALU instructions per wave,
// memory wait only 22 % of wave cycles).  So per link the kernels do exactly:
//     v_readlane (source)  v_readlane (cost)  v_lshl_add (row offset)  global_load (saddr form)
//     v_add_u32 clamp      v_min / v_cmp
// with COMPILE-TIME lane numbers (fully unrolled groups of 4 links), a 32-bit VGPR byte offset on
// top of an SGPR base (needs n * 256 <= 2^32), and everything rare (overload gating, padding,
// zero-cost ordering) folded beforehand into the per-lane link vectors instead of per-link tests.
// One coalesced vector load brings up to 64 links of the row (lane j = link j).

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ uint32_t add_sat(uint32_t a, uint32_t b) {   // v_add_u32 ... clamp
  return __builtin_elementwise_add_sat(a, b);
}
__device__ __forceinline__ uint32_t ld_row(const uint32_t *base, uint32_t byte_off) {
  return *(const uint32_t *)((const char *)base + byte_off);           // saddr + 32-bit voffset
}
__device__ __forceinline__ uint64_t ld_row64(const uint64_t *base, size_t byte_off) {
  return *(const uint64_t *)((const char *)base + byte_off);
}
__device__ __forceinline__ uint64_t ld_row64s(const uint64_t *base, uint32_t byte_off) {
  return *(const uint64_t *)((const char *)base + byte_off);           // saddr + 32-bit voffset
}

__device__ __forceinline__ uint32_t in_fpos_of(const GraphDev &g, uint32_t e) { return g.in_fpos[e]; }

constexpr int GRP = 4;            // links per unrolled group
constexpr int NGRP = 64 / GRP;

// Distance relaxation sweep.  grid = (ceil8(ceil(n / VPB)), n_batches), block = 256.
// MAXINF: max_path_metric == 0xFFFFFFFF (OSPF): u32 saturation must be detected.
template <bool MAXINF>
__global__ __launch_bounds__(256) void k_relax(GraphDev g, uint32_t *__restrict__ dist,
                                               const uint32_t *__restrict__ roots,
                                               uint32_t maxpath, uint32_t ignore_ovl,
                                               int *changed, int sweep,
                                               uint32_t *lane_flags) {
  if (sweep > 0 && changed[sweep - 1] == 0) return;      // converged in an earlier launch
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t batch = blockIdx.y;
  const uint32_t n = g.n;
  uint32_t vbeg, nrows;
  if (!wave_rows(g, blockIdx.x, wave, vbeg, nrows)) return;
  const uint32_t *__restrict__ in_ptr = g.in_ptr;
  const uint32_t *__restrict__ in_src = g.in_src;
  const uint32_t *__restrict__ in_w = g.in_w;
  const uint32_t my_root = roots[batch * 64 + lane];
  uint32_t *D = dist + (size_t)batch * n * 64;
  const uint32_t lane4 = lane * 4u;
  // row bounds of the VPW vertices of this wave: one vector load, lanes 0..VPW
  const uint32_t pv = in_ptr[min(vbeg + min(lane, (uint32_t)VPW), n)];
  bool any = false, sat = false;
#pragma unroll
  for (int i = 0; i < VPW; ++i) {
    const uint32_t v = vbeg + i;
    if (v >= n || (uint32_t)i >= nrows) break;
    const uint32_t e0 = rdlane(pv, i), e1 = rdlane(pv, i + 1);
    const uint32_t d_old = ld_row(D, v * 256u + lane4);
    uint32_t best = d_old;
    for (uint32_t eb = e0; eb < e1; eb += 64) {
      const uint32_t cnt = min(64u, e1 - eb);
      // padding lanes: (own row, cost INF) -> saturates to INF, never improves anything
      uint32_t sv = lane < cnt ? in_src[eb + lane] : v;
      const uint32_t wv = lane < cnt ? in_w[eb + lane] : INF;
      // overload gating is rare: only rows that list an overloaded source take the slow path
      const bool has_nt = !ignore_ovl && __ballot((sv & SRC_NO_TRANSIT) != 0) != 0ull;
      if (!has_nt) sv &= SRC_MASK;
#pragma unroll
      for (int gi = 0; gi < NGRP; ++gi) {
        if (cnt <= (uint32_t)(gi * GRP)) break;
        uint32_t du[GRP];
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
          const uint32_t u = rdlane(sv, gi * GRP + k) & SRC_MASK;
          du[k] = ld_row(D, u * 256u + lane4);
        }
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
          const uint32_t w = rdlane(wv, gi * GRP + k);
          uint32_t d = du[k];
          if (has_nt) {                                           // uniform
            const uint32_t sw = rdlane(sv, gi * GRP + k);
            if ((sw & SRC_NO_TRANSIT) && (sw & SRC_MASK) != my_root) d = INF;
          }
          const uint32_t c = add_sat(d, w);                       // INF stays INF
          if (MAXINF && c == INF && d != INF && w != INF) sat = true;   // genuine u32 saturation
          best = min(best, c);        // candidates above max_path_metric are rejected at the end
        }
      }
    }
    // min over all candidates: if the smallest exceeds max_path_metric, every one does
    if (best < d_old && best <= maxpath) {
      D[(size_t)v * 64 + lane] = best;
      any = true;
    }
  }
  if (__ballot(any) != 0ull && lane == 0) changed[sweep] = 1;
  if (MAXINF && sat) atomicOr(&lane_flags[batch * 64 + lane], LF_NEED_EXACT);
}

// ---------------------------------------------------------------------------------------------
// SPT-DAG sweep: hops + first-hop mask.  Same grid as k_relax.  W = mask words (template).
// GS (only W == 1): accept parents finalised in THIS launch too.  Safe without fences because a
// final row with hops != 0 always has a non-zero mask word (every non-first-hop vertex inherits
// at least one slot), the mask array starts at 0 and is written exactly once: a reader that sees
// a stale or not-yet-written mask sees 0 and simply treats the parent as pending.
// In-links of a row are stored by (cost descending, source ascending) — graph_build.hip.h, kb_rank —, so among the
// tight links the first one in row order has the smallest parent distance and, on ties, the smallest parent index:
// the earliest-popped tight parent (first discoverer).
template <int W, bool GS>
__global__ __launch_bounds__(256) void k_dag(GraphDev g, const uint32_t *__restrict__ dist,
                                             uint32_t *__restrict__ hv, uint64_t *__restrict__ mask,
                                             const uint32_t *__restrict__ roots, SlotTabs tabs,
                                             uint32_t net_nexthops, uint32_t ignore_ovl,
                                             int *changed, int sweep, uint32_t epoch,
                                             uint32_t *lane_flags, uint32_t hc, uint32_t *__restrict__ act) {
  // links per group = neighbour rows requested together: more for narrow masks (a 100-link row of a fat-tree switch is
  // 2 dependent round trips per group), fewer for wide ones (2 * DG * W mask registers)
  constexpr int DG = W <= 2 ? 8 : (W <= 4 ? 4 : 2);
  constexpr int NDG = 64 / DG;
  if (sweep > 0 && changed[sweep - 1] == 0) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t batch = blockIdx.y;
  const uint32_t n = g.n;
  uint32_t vbeg, nrows;
  if (!wave_rows(g, blockIdx.x, wave, vbeg, nrows)) return;
  const uint32_t *__restrict__ in_ptr = g.in_ptr;
  const uint32_t *__restrict__ in_src = g.in_src;
  const uint32_t *__restrict__ in_w = g.in_w;
  const uint32_t root_slot = batch * 64 + lane;
  const uint32_t my_root = roots[root_slot];
  const uint32_t *D = dist + (size_t)batch * n * 64;
  uint32_t *H = hv + (size_t)batch * n * 64;
  uint64_t *M = mask + (size_t)batch * n * 64 * W;
  const uint32_t lane4 = lane * 4u, lane8 = lane * 8u;
  const uint32_t pv = in_ptr[min(vbeg + min(lane, (uint32_t)VPW), n)];
  // Push activation, as in k_fused: a row can only make progress after one of its in-neighbours became final for
  // some root, and a row that does so in sweep s stamps s + 1 on its out-neighbours; every row is due in sweep 0
  // (stamps start at 0).  Without it every pending row re-walks all its links in every sweep.
  uint32_t *A = act + (size_t)batch * n;
  const uint32_t av = A[min(vbeg + min(lane, (uint32_t)VPW - 1u), n - 1)];
  const uint32_t po = g.out_ptr[min(vbeg + min(lane, (uint32_t)VPW), n)];
  bool any = false;
#pragma unroll 1
  for (int i = 0; i < VPW; ++i) {
    const uint32_t v = vbeg + i;
    if (v >= n || (uint32_t)i >= nrows) break;
    if (rdlane(av, i) < (uint32_t)sweep) continue;       // none of its in-neighbours changed in the last sweep
    const uint32_t cur = ld_row(H, v * 256u + lane4);
    const bool need0 = (cur >> HV_EPOCH_SHIFT) == 0;
    if (__ballot(need0) == 0ull) continue;               // whole row already final
    const uint32_t dv = ld_row(D, v * 256u + lane4);
    uint32_t out_hv = 0;
    bool done = need0 && (dv == INF || v == my_root);    // not in SPT, or the root (hops 0)
    const bool need = need0 && !done;
    uint64_t m[W];
#pragma unroll
    for (int k = 0; k < W; ++k) m[k] = 0;
    if (__ballot(need) != 0ull) {
      const bool v_router = !(g.vflags[v] & 1u);
      bool pending = false, anyp = false;
      bool zdone = false;                 // hop-count-like graphs: the one zero-cost parent has been met
      uint32_t bk_d = INF, p0h = 0;
      const uint32_t e0 = rdlane(pv, i), e1 = rdlane(pv, i + 1);
      for (uint32_t eb = e0; eb < e1; eb += 64) {
        const uint32_t cnt = min(64u, e1 - eb);
        uint32_t sv = lane < cnt ? in_src[eb + lane] : v;
        uint32_t wv = lane < cnt ? in_w[eb + lane] : 1u;
        const bool has_nt = !ignore_ovl && __ballot((sv & SRC_NO_TRANSIT) != 0) != 0ull;
        if (!has_nt) sv &= SRC_MASK;
        // A zero-cost link from a HIGHER-numbered source is never a static-order parent; padding
        // lanes likewise: rewrite both as (own row, cost 1), which can never be tight.  Exception:
        // hop-count-like graphs (see fused_row_any): there the FIRST tight one of them, in row order
        // = lowest source, is the vertex' only parent.
        const uint32_t zvec = (lane < cnt && wv == 0u && (sv & SRC_MASK) >= v) ? 1u : 0u;
        const bool has_zl = hc != 0u && __ballot(zvec != 0u) != 0ull;
        if (lane >= cnt || (zvec && !hc)) { sv = v; wv = 1u; }
#pragma unroll
        for (int gi = 0; gi < NDG; ++gi) {
          if (cnt <= (uint32_t)(gi * DG)) break;
          uint32_t du[DG];
          bool tight[DG];
#pragma unroll
          for (int k = 0; k < DG; ++k) {                          // no requests for the padding of a short row
            const uint32_t u = rdlane(sv, gi * DG + k) & SRC_MASK;
            du[k] = (uint32_t)(gi * DG + k) < cnt ? ld_row(D, u * 256u + lane4) : INF;
          }
          bool anyt = false;
#pragma unroll
          for (int k = 0; k < DG; ++k) {
            const uint32_t w = rdlane(wv, gi * DG + k);
            uint32_t d = du[k];
            if (has_nt) {
              const uint32_t sw = rdlane(sv, gi * DG + k);
              if ((sw & SRC_NO_TRANSIT) && (sw & SRC_MASK) != my_root) d = INF;
            }
            // tight parent (d + w == dv without saturation; dv != INF for `need` lanes)
            tight[k] = need && d != INF && add_sat(d, w) == dv;
            if (has_zl && rdlane(zvec, gi * DG + k) != 0u) {    // uniform, hop-count-like graphs only
              tight[k] = tight[k] && !zdone;
              zdone = zdone || tight[k];
            }
            anyt = anyt || tight[k];
          }
          if (__ballot(anyt) == 0ull) continue;
          uint32_t hu[DG];
          uint64_t mu[DG][W];
#pragma unroll
          for (int k = 0; k < DG; ++k) {                          // hops / masks only of links tight for some root
            const uint32_t u = rdlane(sv, gi * DG + k) & SRC_MASK;
            const bool tk = __ballot(tight[k]) != 0ull;
            hu[k] = tk ? ld_row(H, u * 256u + lane4) : 0u;
#pragma unroll
            for (int q = 0; q < W; ++q) mu[k][q] = tk ? ld_row64(M, ((size_t)u * W + q) * 512u + lane8) : 0ull;
          }
#pragma unroll
          for (int k = 0; k < DG; ++k) {
            const uint32_t hh = hu[k] & 0xFFFFu;
            const uint32_t eu = hu[k] >> HV_EPOCH_SHIFT;
            bool ready;
            if (GS) {
              bool mz = true;
#pragma unroll
              for (int q = 0; q < W; ++q) mz = mz && (mu[k][q] == 0);
              ready = tight[k] && eu != 0 && (hh == 0 || !mz);
            } else {
              ready = tight[k] && eu != 0 && eu < epoch;         // final since an earlier launch
            }
            anyp = anyp || tight[k];
            pending = pending || (tight[k] && !ready);
            if (ready && du[k] < bk_d) { bk_d = du[k]; p0h = hh; }
            const bool direct = ready && hh == 0;                // parent: root or hops-0 network
            if (__ballot(direct) != 0ull) {
              const uint32_t u = rdlane(sv, gi * DG + k) & SRC_MASK;
              const uint32_t fpos = g.in_fpos[eb + gi * DG + k];
              if (direct && (v_router || net_nexthops)) {
                const uint32_t base_s = (u == my_root) ? 0u : slot_base_of(tabs, root_slot, u);
                const uint32_t sidx = base_s + fpos;
#pragma unroll
                for (int q = 0; q < W; ++q)
                  if ((sidx >> 6) == (uint32_t)q) m[q] |= 1ull << (sidx & 63u);
              }
            }
            const bool inherit = ready && hh != 0;
#pragma unroll
            for (int q = 0; q < W; ++q) m[q] |= inherit ? mu[k][q] : 0ull;
          }
        }
      }
      if (need && !anyp) {
        // In the SPT but no parent precedes it in static order: the reference's pop order is
        // dynamic here (zero-cost plateau).  Placeholder; k_repair recomputes hops / masks of the root in the true order.
        atomicOr(&lane_flags[root_slot], LF_DYN);
        out_hv = 0; done = true;
#pragma unroll
        for (int k = 0; k < W; ++k) m[k] = 0;
      } else if (need && !pending) {
        uint32_t hops = p0h + (v_router ? 1u : 0u);
        if (hops > 0xFFFFu) hops = 0xFFFFu;              // u16 saturating_add
        out_hv = hops; done = true;
      }
    }
    if (done) {
#pragma unroll
      for (int k = 0; k < W; ++k) M[((size_t)v * W + k) * 64 + lane] = m[k];
      H[(size_t)v * 64 + lane] = out_hv | (epoch << HV_EPOCH_SHIFT);
      any = true;
    }
    if (__ballot(done) != 0ull) {                        // wake the out-neighbours up for the next sweep
      const uint32_t o0 = rdlane(po, i), o1 = rdlane(po, i + 1);
      for (uint32_t ob = o0 + lane; ob < o1; ob += 64) A[g.out_dst[ob]] = (uint32_t)sweep + 1u;
    }
  }
  if (__ballot(any) != 0ull && lane == 0) changed[sweep] = 1;
}

// ---------------------------------------------------------------------------------------------
// Fused sweep (the fast path when every root of the run has <= 24 first-hop slots): ONE
// label-correcting fixed point over a packed per-(vertex, root) state instead of a distance phase
// followed by a DAG phase.  Two state widths, same code (template parameter ST):
//   wide   (uint64_t)  [63:32] dist   [31:M] hops   [M-1:0] first-hop mask   (M = 16, or up to 24 for runs whose
//          roots have 17-24 slots: 8-15 hop bits, LF_OVERFLOW -> redone on the two-phase path)
//   narrow (uint32_t)  [31:sh] dist   [sh-1:M] hops  [M-1:0] mask      (M = slots of the run,
//          H = sh-M hop bits; chosen by the host when the graph's costs fit; a lane that gets
//          within one link cost of the field limit, or too many hops, raises LF_OVERFLOW and the
//          whole run is redone wide -> results never depend on the width)
// all ones = not reached.  Every recomputation of a lane is a pure function of its in-neighbours':
//     dist  = min over in-links of dist[u] (+) w           (gates as in k_relax)
//     mask  = OR over the links attaining the min of (hops[u] == 0 ? slot bit : mask[u])
//     hops  = hops[p0] + is_router(v),  p0 = first link (ascending source) with the smallest dist[u]
// so the unique fixed point is the reference's result whenever its pop order is the static
// (dist, index) order (same proof obligation as k_dag; lanes that violate it are flagged for
// k_exact).  A state is read and written by ONE access per lane, so a reader can never see a
// distance with somebody else's mask.  Stale reads inside a launch are harmless: memory is
// monotone across launches and the run ends only after a launch in which nothing changed.
//
// Arithmetic is done in "key space": dkey = the distance field left in place (low bits zero),
// wkey = cost << sh, candidate = v_add_u32 clamp: a sum that leaves the field saturates to all
// ones, which is >= inf_t, i.e. "not reached"; comparisons are plain u32 compares.
//
// Row skipping by PUSH activation: act[v] = id of the latest sweep in which row v has to be
// recomputed.  A wave that changes row u in sweep c stamps c+1 on u's out-neighbours (<= deg
// scattered 4-byte stores), so the next sweep's "is there anything to do here" test is ONE load
// per wave, and a sweep with nothing to do costs a few microseconds.  A stale or racing stamp can
// only cause an extra recomputation, never a missed one: every change made in sweep c is seen
// (kernel boundary) by all its dependents in sweep c+1.  isis-100k: ~21 of 34 row-sweeps visited.
//
// Memory-level parallelism: the link vectors of all VPW vertices are fetched up front and ALL
// neighbour rows of a row are requested before the first one is consumed (profiles/r01b notes:
// the first versions, 4 rows per round trip, ran at exactly waves/resident x 12 x loaded latency).
// rows_done: [256] work counter of a run launched with HSPF_RUN_COUNT_ROWS (the COUNT instantiation of k_fused): rows
// evaluated, spread over 256 words and summed by the host (hspf_stats.rows_recomputed).  Not touched otherwise: even
// one fire-and-forget atomic per wave behind a scalar pointer load measurably lengthens the ~10 us life of a wave
// (profiles/r02_notes.md: 0.93 vs 0.85 ms per batch).
// giant_part: [tags: batches x n_giant, padded to 64 words | accumulators: batches x slices x GIANT_WORDS x 64 lanes]; a tag
// holds sweep + 1 of the sweep whose k_giant_part filled the row's slices for that batch (0: never)
// dyn_part (round 6): LF_DYN of the lane = root sweeps goes to one of DYN_PARTS partial status arrays [DYN_PARTS][dyn_L] (null: straight
// to lane_flags): on a graph with 1 % zero-cost links 5 000 waves per pass each raised the SAME 64 words — atomics on one address
// retire at ~10 ns apiece, whoever issues them, and a 20 us dense pass took 80.  k_lf_reduce folds them into lane_flags.
constexpr uint32_t DYN_PARTS = 64u;
struct FusedGraph { GraphDev g; SlotTabs tabs; uint32_t *rows_done; uint32_t *giant_part; uint32_t *dyn_part; uint32_t dyn_L; };   // device-resident descriptor of one run
__device__ __forceinline__ void raise_dyn(const FusedGraph *gp, uint32_t *lane_flags, uint32_t root_slot, uint32_t wave_id) {
  uint32_t *dp = gp->dyn_part;
  if (dp) {
    uint32_t *w = dp + (size_t)(wave_id & (DYN_PARTS - 1u)) * gp->dyn_L + root_slot;
    if (!(*(volatile uint32_t *)w & LF_DYN)) atomicOr(w, LF_DYN);
  } else atomicOr(&lane_flags[root_slot], LF_DYN);
}
__global__ void k_lf_reduce(uint32_t *__restrict__ lane_flags, const uint32_t *__restrict__ part, uint32_t L) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  uint32_t x = 0;
  for (uint32_t k = 0; k < DYN_PARTS; ++k) x |= part[(size_t)k * L + i];
  if (x) lane_flags[i] |= x;
}

struct FusedParams {
  uint32_t sh;        // bit position of the dist field (narrow) / 0 (wide: dist is the high word)
  uint32_t mbits;     // mask field width
  uint32_t hmax;      // largest representable hops value
  uint32_t inf_t;     // dkey >= inf_t  <=> not reached
  uint32_t maxkey;    // dkey >  maxkey <=> beyond max_path_metric
  uint32_t ovf_t;     // narrow: dkey >= ovf_t (and reached) -> a later sum could leave the field
  uint32_t hc;        // 1: hop-count-like graph (every link into a network costs 0, into a router 1)
  uint32_t infw;      // narrow: the word stored for "not reached" (all ones; inf_t for the lean sweep, whose sums must not wrap)
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Buffer resource of one batch's state slab, built from values forced into SGPRs: a descriptor
// that the compiler keeps in VGPRs turns every load into a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const void *S, uint32_t bytes) {
  const uint64_t a = (uint64_t)S;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// One row element through a raw buffer load: address = rsrc.base + voff (lane * sizeof(ST), VGPR)
// + soff (row byte offset, SGPR straight out of v_readlane): no vector ALU work per link for
// addressing.  Split into (dkey, pay): pay = hops << mbits | mask.
template <typename ST> struct StIO;
template <> struct StIO<uint64_t> {
  static constexpr uint32_t ROW_SHIFT = 9;
  struct Raw { u32x2 x; };
  static __device__ __forceinline__ Raw ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return Raw{__builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)};
  }
  static __device__ __forceinline__ uint32_t dkey(const Raw &q, const FusedParams &) { return q.x.y; }
  static __device__ __forceinline__ uint32_t pay(const Raw &q, const FusedParams &) { return q.x.x; }
  static __device__ __forceinline__ uint64_t join(uint32_t dk, uint32_t hops, uint32_t mask, const FusedParams &P) {
    return ((uint64_t)dk << 32) | ((uint64_t)hops << P.mbits) | mask;   // low word = [hops | mask], mbits = 16..24
  }
  static __device__ __forceinline__ uint64_t bits(const Raw &q) { return ((uint64_t)q.x.y << 32) | q.x.x; }
};
template <> struct StIO<uint32_t> {
  static constexpr uint32_t ROW_SHIFT = 8;
  struct Raw { uint32_t x; };
  static __device__ __forceinline__ Raw ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return Raw{__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)};
  }
  static __device__ __forceinline__ uint32_t dkey(const Raw &q, const FusedParams &P) { return q.x & ~((1u << P.sh) - 1u); }
  static __device__ __forceinline__ uint32_t pay(const Raw &q, const FusedParams &P) { return q.x & ((1u << P.sh) - 1u); }
  static __device__ __forceinline__ uint32_t join(uint32_t dk, uint32_t hops, uint32_t mask, const FusedParams &P) {
    return dk | (hops << P.mbits) | mask;
  }
  static __device__ __forceinline__ uint32_t bits(const Raw &q) { return q.x; }
};

struct RowAcc { uint32_t bd, bm, bpd, bh; bool sat; };
template <typename ST> struct RowOut { ST nw; bool sat, dyn, ovf; };

template <typename ST, bool MAXINF>
__device__ __forceinline__ void acc_link(RowAcc &a, const typename StIO<ST>::Raw &q, uint32_t wkey, const FusedParams &P) {
  const uint32_t d = StIO<ST>::dkey(q, P), pay = StIO<ST>::pay(q, P);
  const uint32_t c = add_sat(d, wkey);                            // leaves the field -> all ones
  if (MAXINF && sizeof(ST) == 8 && c == INF && d != INF && wkey != INF) a.sat = true;
  const bool lt = c < a.bd, eq = c == a.bd;                       // c == bd == "not reached": harmless
  const uint32_t hh = pay >> P.mbits, contrib = pay & ((1u << P.mbits) - 1u);
  const uint32_t m_or = a.bm | contrib;
  a.bm = lt ? contrib : (eq ? m_or : a.bm);
  const bool newp = lt || (eq && d < a.bpd);                      // first discoverer: smallest parent dist
  a.bpd = newp ? d : a.bpd;
  a.bh = newp ? hh : a.bh;
  a.bd = min(a.bd, c);
}

template <typename ST>
__device__ __forceinline__ RowOut<ST> finish_row(const RowAcc &a, uint32_t v, uint32_t my_root, uint32_t v_router,
                                                 uint32_t bd_all, const FusedParams &P) {
  RowOut<ST> o;
  o.sat = a.sat;
  o.ovf = false;
  if (v == my_root) o.nw = (ST)0;                                 // dist 0, hops 0, no next hops
  else if (a.bd >= P.inf_t || a.bd > P.maxkey) o.nw = sizeof(ST) == 8 ? (ST)~(ST)0 : (ST)P.infw;
  else {
    uint32_t hops = a.bh + v_router;
    // 16 hop bits: the reference's u16 saturating_add; fewer (4-byte state, or the 8-byte one with more than 16 mask
    // bits): the run is redone with the next wider representation
    if (hops > P.hmax) { hops = P.hmax; o.ovf = P.hmax < 0xFFFFu; }
    o.ovf = o.ovf || a.bd >= P.ovf_t;
    o.nw = StIO<ST>::join(a.bd, hops, a.bm & ((1u << P.mbits) - 1u), P);
  }
  // a shorter (or the only) way in through a zero-cost link from a higher-numbered source: the reference's pop order is
  // dynamic there.  The DISTANCE is order-independent and is stored (so that everything downstream converges to its final
  // distance too); hops and mask of this vertex and of whatever hangs below it are placeholders — the root is flagged
  // LF_DYN and k_repair (spf_repair.hip.h) recomputes them in the true pop order on the emitted tables.
  o.dyn = v != my_root && bd_all < P.inf_t && bd_all <= P.maxkey && bd_all < a.bd;
  if (o.dyn) {
    o.ovf = o.ovf || bd_all >= P.ovf_t;
    o.nw = StIO<ST>::join(bd_all, 0u, 0u, P);
  }
  return o;
}

// A zero-cost link from a higher- or equal-numbered source (u >= v, cost 0) is never a parent in the STATIC order — unless it
// is the row's only way in: then v enters the candidate list when the first such source is popped and, its index being the
// lower one, is popped right after it: ONE parent, the lowest-numbered one (rows list equal costs by ascending source), no
// union — the rule hop-count graphs have always used (fused_row_any).  Round 6 applies it to every graph, for the links that
// are SAFE (GraphDev::zcyc: not inside a possible cycle of zero-cost links, so the dependencies of a sweep stay acyclic and the
// fixed point unique); an unsafe one still only feeds the distance (bd_all).  Either way the root is flagged LF_DYN: the rule
// is exact only while every vertex involved is released in index order, and k_repair checks that in the true order.
__device__ __forceinline__ bool zlink_unsafe(const uint8_t *zcyc, uint32_t u, uint32_t v) { return !zcyc || (zcyc[u] && zcyc[v]); }
template <typename ST>
__device__ __forceinline__ RowOut<ST> finish_row_z(RowAcc &a, uint32_t v, uint32_t my_root, uint32_t v_router, uint32_t bd_all,
                                                   uint32_t zb, uint32_t zm, uint32_t zh, bool hc, const FusedParams &P) {
  const bool late = zb < a.bd;                                      // strictly better through the zero-cost links
  a.bm = late ? zm : a.bm; a.bh = late ? zh : a.bh; a.bd = late ? zb : a.bd;
  RowOut<ST> o = finish_row<ST>(a, v, my_root, v_router, bd_all, P);
  o.dyn = o.dyn || (!hc && late && v != my_root && a.bd < P.inf_t && a.bd <= P.maxkey);
  return o;
}

// Fast row routine: at most 16 links and none of the rare conditions.  The kernel is bound by
// vector-ALU issue, not by bytes (a wave64 integer op occupies its SIMD for 4 cycles), so the
// routine is written for vector instruction count:
//   * the link records (source, cost) of a row are one coalesced vector load each (lane j = link
//     j), broadcast by v_readlane; the neighbour rows come through raw buffer loads whose row
//     offset is that SGPR: no vector instruction is spent on addressing.  (Fetching the records
//     through the scalar cache instead saves the two v_readlane per link but costs 32 SGPRs per row
//     and a dependent scalar round trip per row: measured slower, 52 vs 45 us per sweep.);
//   * pass 1   c_j = dkey_j (+) wkey_j ;  bd = min(c_j)           (v_add_u32 clamp, v_min3_u32)
//     pass 2   t_j = (c_j == bd) ;  macc |= t_j ? pay_j : 0 ;  bpay = t_j ? pay_j : bpay  (LAST link first)
//   * exactly deg(v) links are processed: the routine is a compile-time recursion over the link
//     index with one uniform branch per link (no padding work), and every neighbour row is
//     requested before the first one is consumed.  Values are only ever defined in a block that
//     dominates their uses, so no loaded value needs a phi (a phi makes the compiler wait).
// The in-links of a row are stored by (cost descending, source ascending) — hspf_graph_upload —
// so among the tight links (equal c_j) the FIRST one in row order has the smallest parent distance
// and, on ties, the smallest parent index: the reference's first discoverer.  Walking pass 2
// backwards makes a plain overwrite end on it.
typedef uint32_t u32x16a __attribute__((ext_vector_type(16), aligned(4)));
typedef uint32_t u32x8a __attribute__((ext_vector_type(8), aligned(4)));

template <typename ST> struct PayBits;
template <> struct PayBits<uint64_t> { static __device__ __forceinline__ uint32_t of(const StIO<uint64_t>::Raw &q) { return q.x.x; } };
template <> struct PayBits<uint32_t> { static __device__ __forceinline__ uint32_t of(const StIO<uint32_t>::Raw &q) { return q.x; } };

template <typename ST> struct Row16 {
  __amdgpu_buffer_rsrc_t rs;
  uint32_t lvo, cnt;
  uint32_t so, wk;                 // lane j = link j: row byte offset of the source / shifted cost
  FusedParams P;
  typename StIO<ST>::Raw q[16];
  uint32_t c[16];
  uint32_t bd, macc, bpay;
  bool sat;
};

template <typename ST, bool MAXINF, int L>
__device__ __forceinline__ void row16_compute(Row16<ST> &r) {
  if constexpr (sizeof(ST) == 4) {
    // 4-byte state: add the shifted cost to the WHOLE word — the pay bits ride along in the low sh bits, the sum
    // leaves 32 bits exactly when the distance leaves its field — so the field is never masked out per link (one
    // v_and less of 8 vector instructions), the min runs over whole words (its distance field is the min distance), a
    // link is tight iff its word is <= (min distance | all pay bits), and the pay bits come out of the sum itself.
    const uint32_t paym = (1u << r.P.sh) - 1u;
    uint32_t m = INF;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      r.c[j] = add_sat(PayBits<ST>::of(r.q[j]), rdlane(r.wk, j));
      m = min(m, r.c[j]);
    }
    const uint32_t thr = m | paym;
    uint32_t macc = 0u, bpay = 0u;
#pragma unroll
    for (int j = L - 1; j >= 0; --j) {
      const bool t = r.c[j] <= thr;
      macc |= t ? r.c[j] : 0u;
      bpay = t ? r.c[j] : bpay;
    }
    r.bd = m & ~paym; r.macc = macc; r.bpay = bpay;
    return;
  }
  uint32_t bd = INF;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    const uint32_t d = StIO<ST>::dkey(r.q[j], r.P);
    r.c[j] = add_sat(d, rdlane(r.wk, j));                         // leaves the field -> all ones
    if (MAXINF && sizeof(ST) == 8 && r.c[j] == INF && d != INF) r.sat = true;
    bd = min(bd, r.c[j]);
  }
  uint32_t macc = 0u, bpay = 0u;
#pragma unroll
  for (int j = L - 1; j >= 0; --j) {
    const bool t = r.c[j] == bd;
    const uint32_t pb = PayBits<ST>::of(r.q[j]);
    macc |= t ? pb : 0u;
    bpay = t ? pb : bpay;
  }
  r.bd = bd; r.macc = macc; r.bpay = bpay;
}

template <typename ST, bool MAXINF, int L>
__device__ __forceinline__ void row16_case(Row16<ST> &r) {       // exactly L links: L requests, then compute
#pragma unroll
  for (int j = 0; j < L; ++j) r.q[j] = StIO<ST>::ld(r.rs, r.lvo, rdlane(r.so, j));
  row16_compute<ST, MAXINF, L>(r);
}

template <typename ST, bool MAXINF>
__device__ __forceinline__ RowOut<ST> fused_row16(__amdgpu_buffer_rsrc_t rs, uint32_t v, uint32_t cnt, uint32_t sv,
                                                  uint32_t wv, uint32_t lvo, uint32_t my_root, uint32_t v_router,
                                                  const FusedParams &P) {
  Row16<ST> r;
  r.rs = rs; r.lvo = lvo; r.cnt = cnt; r.P = P; r.sat = false;
  r.bd = INF; r.macc = 0u; r.bpay = 0u;
  r.so = (sv & SRC_MASK) << StIO<ST>::ROW_SHIFT;
  r.wk = wv << P.sh;                                              // host guarantees wv << sh fits
  switch (cnt) {                                                  // uniform; one straight-line block per degree
    case 0: break;
    case 1: row16_case<ST, MAXINF, 1>(r); break;   case 2: row16_case<ST, MAXINF, 2>(r); break;
    case 3: row16_case<ST, MAXINF, 3>(r); break;   case 4: row16_case<ST, MAXINF, 4>(r); break;
    case 5: row16_case<ST, MAXINF, 5>(r); break;   case 6: row16_case<ST, MAXINF, 6>(r); break;
    case 7: row16_case<ST, MAXINF, 7>(r); break;   case 8: row16_case<ST, MAXINF, 8>(r); break;
    case 9: row16_case<ST, MAXINF, 9>(r); break;   case 10: row16_case<ST, MAXINF, 10>(r); break;
    case 11: row16_case<ST, MAXINF, 11>(r); break; case 12: row16_case<ST, MAXINF, 12>(r); break;
    case 13: row16_case<ST, MAXINF, 13>(r); break; case 14: row16_case<ST, MAXINF, 14>(r); break;
    case 15: row16_case<ST, MAXINF, 15>(r); break; default: row16_case<ST, MAXINF, 16>(r); break;
  }
  // pay bits of the winner / of the union: wide = the low word, narrow = the whole word (distance
  // bits are dropped by the masks below)
  RowAcc a;
  a.bd = r.bd; a.sat = r.sat; a.bpd = 0u;
  a.bm = r.macc & ((1u << P.mbits) - 1u);
  a.bh = sizeof(ST) == 8 ? (r.bpay >> P.mbits) : ((r.bpay >> P.mbits) & P.hmax);
  return finish_row<ST>(a, v, my_root, v_router, INF, P);
}

// General row routine (rare: rows next to a vertex that can have hops == 0, overloaded sources,
// zero-cost links from higher-numbered sources, more than 16 links): rolled loops, four neighbour rows in flight.
// Inlined all the same: a real call would need a stack, i.e. scratch memory.
// Accumulator of the general routine: RowAcc + the zero-cost-link side state.
struct AnyAcc { RowAcc a; uint32_t bd_all, zb, zm, zh; };
__device__ __forceinline__ AnyAcc any_acc_init() { return AnyAcc{RowAcc{INF, 0u, INF, 0u, false}, INF, INF, 0u, 0u}; }

// Up to 64 links (lane j of sv / wv = link eb + j: source | NT bit, cost; lanes >= cnt: the row itself / INF) into x.
template <typename ST, bool MAXINF, bool HC, int PFN>
__device__ __forceinline__ void row_any_chunk(AnyAcc &x, const GraphDev &g, __amdgpu_buffer_rsrc_t rs, uint32_t v,
                                              uint32_t v_router, uint32_t eb, uint32_t cnt, uint32_t sv, uint32_t wv,
                                              uint32_t lane, uint32_t lvo, uint32_t my_root, uint32_t root_slot,
                                              const SlotTabs &tabs, uint32_t net_nexthops, uint32_t ignore_ovl,
                                              const FusedParams &P) {
  RowAcc &a = x.a;
  const bool has_nt = !ignore_ovl && __ballot(lane < cnt && (sv & SRC_NO_TRANSIT) != 0) != 0ull;
  const uint32_t zv = (lane < cnt && wv == 0u && (sv & SRC_MASK) >= v) ? 1u : 0u;
  const bool has_z = __ballot(zv != 0u) != 0ull;
  const uint32_t so = (lane < cnt ? (sv & SRC_MASK) : v) << StIO<ST>::ROW_SHIFT;
  const uint32_t wk = lane < cnt ? (wv << P.sh) : INF;
  // PF neighbour rows are requested before the first of them is consumed: a row with 100 links (a fat-tree switch, a
  // big LAN) otherwise pays 100 dependent round trips.  Lanes >= cnt of `so` point at the own row: harmless loads.
  // 4 in the sweep kernel (the registers of more cost the headline 9 %, r02m; 16 in the work-unit instantiation alone cost
  // isis-100k + a 100-router LAN 0.84 -> 0.96 ms, r02t), 16 in k_giant_part.
  constexpr uint32_t PF = (uint32_t)PFN;
#pragma unroll 1
  for (uint32_t j0 = 0; j0 < cnt; j0 += PF) {
    typename StIO<ST>::Raw qq[PF];
#pragma unroll
    for (uint32_t k = 0; k < PF; ++k) qq[k] = StIO<ST>::ld(rs, lvo, rdlane(so, min(j0 + k, 63u)));
#pragma unroll
    for (uint32_t k = 0; k < PF; ++k) {
      const uint32_t j = j0 + k;
      if (j >= cnt) break;
      typename StIO<ST>::Raw q = qq[k];
      const uint32_t w = rdlane(wk, j);
      const uint32_t sw = rdlane(sv, j);
      uint32_t d = StIO<ST>::dkey(q, P);
      uint32_t pay = StIO<ST>::pay(q, P);
      if (has_nt && (sw & SRC_NO_TRANSIT) && (sw & SRC_MASK) != my_root) d = INF;   // overloaded source
      const uint32_t c = add_sat(d, w);
      if (MAXINF && sizeof(ST) == 8 && c == INF && d != INF && w != INF) a.sat = true;
      const bool zlink = has_z && rdlane(zv, j) != 0u;            // uniform
      if (zlink && !HC && zlink_unsafe(g.zcyc, sw & SRC_MASK, v)) { x.bd_all = min(x.bd_all, c); continue; }
      const bool lt = zlink ? (c < x.zb) : (c < a.bd), eq = !zlink && c == a.bd;
      const uint32_t hh = pay >> P.mbits;
      uint32_t contrib = pay & ((1u << P.mbits) - 1u);
      const bool direct = (lt || eq) && hh == 0u && c < P.inf_t;  // parent: root or hops-0 network
      if (__ballot(direct) != 0ull) {
        const uint32_t u = sw & SRC_MASK;
        const uint32_t fpos = g.in_fpos[eb + j];
        if (direct) {
          const uint32_t base_s = (u == my_root) ? 0u : slot_base_of(tabs, root_slot, u);
          const uint32_t sidx = base_s + fpos;
          contrib = ((v_router || net_nexthops) && sidx < P.mbits) ? (1u << sidx) : 0u;
        }
      }
      if (zlink) {
        // (every graph since round 6, for SAFE links: finish_row_z)  Hop-count-like graph (holo-isis MetricMode::HopCount, spf.rs:1138-1145): a network whose
        // only way in is a zero-cost link from routers of the same distance is put on the candidate
        // list by the FIRST of them to be popped — the lowest-numbered one, rows list equal-cost links
        // by ascending source — and, its own index being lower than any router's, is popped next:
        // the later routers find it on the SPT already (spf.rs:630-632).  One parent, no union.
        if (lt) { x.zb = c; x.zm = contrib; x.zh = hh; }
        continue;
      }
      const uint32_t m_or = a.bm | contrib;
      a.bm = lt ? contrib : (eq ? m_or : a.bm);
      const bool newp = lt || (eq && d < a.bpd);
      a.bpd = newp ? d : a.bpd;
      a.bh = newp ? hh : a.bh;
      a.bd = min(a.bd, c);
    }
  }
}

template <typename ST, bool HC>
__device__ __forceinline__ RowOut<ST> any_finish(AnyAcc &x, uint32_t v, uint32_t my_root, uint32_t v_router,
                                                 const FusedParams &P) {
  return finish_row_z<ST>(x.a, v, my_root, v_router, x.bd_all, x.zb, x.zm, x.zh, HC, P);
}

template <typename ST, bool MAXINF, bool HC, int PFN>
__device__ __forceinline__ RowOut<ST> fused_row_any(const GraphDev &g, __amdgpu_buffer_rsrc_t rs, uint32_t v,
                                                    uint32_t e0, uint32_t e1, uint32_t sv0, uint32_t wv0,
                                                    uint32_t lane, uint32_t lvo, uint32_t my_root,
                                                    uint32_t root_slot, const SlotTabs &tabs,
                                                    uint32_t net_nexthops, uint32_t ignore_ovl,
                                                    const FusedParams &P) {
  const uint32_t v_router = (g.vflags[v] & 1u) ? 0u : 1u;
  AnyAcc x = any_acc_init();
  for (uint32_t eb = e0; eb < e1; eb += 64) {
    const uint32_t cnt = min(64u, e1 - eb);
    uint32_t sv = sv0, wv = wv0;
    if (eb != e0) {
      sv = lane < cnt ? g.in_src[eb + lane] : v;
      wv = lane < cnt ? g.in_w[eb + lane] : INF;
    }
    row_any_chunk<ST, MAXINF, HC, PFN>(x, g, rs, v, v_router, eb, cnt, sv, wv, lane, lvo, my_root, root_slot, tabs,
                                       net_nexthops, ignore_ovl, P);
  }
  return any_finish<ST, HC>(x, v, my_root, v_router, P);
}

// ---- giant rows ----------------------------------------------------------------------------------------
// Merge of two accumulators, `y` covering links that FOLLOW those of `x` in row order: what walking y's links after x's
// would have left (strictly smaller candidate replaces, equal one ORs the mask and takes over the parent only with a
// strictly smaller parent distance; the zero-cost side state: strictly smaller only).
__device__ __forceinline__ void any_merge(AnyAcc &x, const AnyAcc &y) {
  const bool lt = y.a.bd < x.a.bd, eq = y.a.bd == x.a.bd;
  const bool newp = lt || (eq && y.a.bpd < x.a.bpd);
  x.a.bm = lt ? y.a.bm : (eq ? (x.a.bm | y.a.bm) : x.a.bm);
  x.a.bpd = newp ? y.a.bpd : x.a.bpd;
  x.a.bh = newp ? y.a.bh : x.a.bh;
  x.a.bd = min(x.a.bd, y.a.bd);
  x.a.sat = x.a.sat || y.a.sat;
  x.bd_all = min(x.bd_all, y.bd_all);
  const bool zl = y.zb < x.zb;
  x.zm = zl ? y.zm : x.zm; x.zh = zl ? y.zh : x.zh; x.zb = min(x.zb, y.zb);
}

__device__ __forceinline__ uint32_t giant_index(const GraphDev &g, uint32_t v) {      // v is a giant row
  uint32_t lo = 0, hi = g.n_giant;
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (g.giant_vtx[mid] <= v) lo = mid; else hi = mid; }
  return lo;
}
__device__ __forceinline__ uint32_t giant_tag_words(uint32_t batches, uint32_t n_giant) { return (batches * n_giant + 63u) & ~63u; }

// One workgroup per (slice of a giant row, batch): wave w evaluates links [64 w, 64 w + 64) of the slice with the general
// routine, the four accumulators are merged in row order through LDS and stored (GIANT_WORDS x 64 lanes).  Launched
// before every sweep of k_fused on graphs with giant rows; a row that is not due costs its slices one load each.
template <typename ST, bool MAXINF, bool HC>
__global__ __launch_bounds__(256) void k_giant_part(const FusedGraph *__restrict__ gp, const int *changed, int sweep,
                                                    const uint32_t *__restrict__ act, uint32_t n, const ST *__restrict__ st,
                                                    const uint32_t *__restrict__ roots, uint32_t net_nexthops,
                                                    uint32_t ignore_ovl, FusedParams P) {
  __shared__ uint32_t sh[3][GIANT_WORDS][64];
  if (sweep > 0 && changed[sweep - 1] == 0) return;
  const GraphDev &g = gp->g;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slice = blockIdx.x, batch = blockIdx.y;
  uint32_t gi = 0;                                     // the giant row that owns the slice
  { uint32_t lo = 0, hi = g.n_giant;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (g.giant_slice0[mid] <= slice) lo = mid; else hi = mid; }
    gi = lo; }
  const uint32_t v = g.giant_vtx[gi];
  if (act[(size_t)batch * n + v] < (uint32_t)sweep + 2u) return;          // not due in this sweep (uniform)
  const uint32_t s_in_row = slice - g.giant_slice0[gi];
  const uint32_t e0 = g.in_ptr[v], e1 = g.in_ptr[v + 1];
  const uint32_t eb = e0 + s_in_row * GIANT_SLICE + wave * 64u;
  const uint32_t cnt = eb < e1 ? min(64u, e1 - eb) : 0u;
  const uint32_t root_slot = batch * 64 + lane;
  const uint32_t my_root = roots[root_slot];
  const ST *S = st + (size_t)batch * n * 64;
  const __amdgpu_buffer_rsrc_t rs = st_rsrc(S, n << StIO<ST>::ROW_SHIFT);
  const uint32_t lvo = lane * (uint32_t)sizeof(ST);
  const uint32_t v_router = (g.vflags[v] & 1u) ? 0u : 1u;
  AnyAcc x = any_acc_init();
  if (cnt) {
    const uint32_t sv = lane < cnt ? g.in_src[eb + lane] : v;
    const uint32_t wv = lane < cnt ? g.in_w[eb + lane] : INF;
    row_any_chunk<ST, MAXINF, HC, 16>(x, g, rs, v, v_router, eb, cnt, sv, wv, lane, lvo, my_root, root_slot, gp->tabs,
                                      net_nexthops, ignore_ovl, P);
  }
  if (wave) {
    uint32_t (*o)[64] = sh[wave - 1u];
    o[0][lane] = x.a.bd; o[1][lane] = x.a.bm; o[2][lane] = x.a.bpd; o[3][lane] = x.a.bh | (x.a.sat ? 0x80000000u : 0u);
    o[4][lane] = x.bd_all; o[5][lane] = x.zb; o[6][lane] = x.zm; o[7][lane] = x.zh;
  }
  __syncthreads();
  if (wave) return;
  for (uint32_t k = 0; k < 3u; ++k) {
    const uint32_t (*o)[64] = sh[k];
    AnyAcc y{RowAcc{o[0][lane], o[1][lane], o[2][lane], o[3][lane] & 0x7FFFFFFFu, (o[3][lane] >> 31) != 0u},
             o[4][lane], o[5][lane], o[6][lane], o[7][lane]};
    any_merge(x, y);
  }
  const uint32_t B = gridDim.y, n_slices = g.giant_slice0[g.n_giant];
  uint32_t *part = gp->giant_part + giant_tag_words(B, g.n_giant) + ((size_t)batch * n_slices + slice) * (GIANT_WORDS * 64u);
  part[0 * 64 + lane] = x.a.bd; part[1 * 64 + lane] = x.a.bm; part[2 * 64 + lane] = x.a.bpd;
  part[3 * 64 + lane] = x.a.bh | (x.a.sat ? 0x80000000u : 0u);
  part[4 * 64 + lane] = x.bd_all; part[5 * 64 + lane] = x.zb; part[6 * 64 + lane] = x.zm; part[7 * 64 + lane] = x.zh;
  if (s_in_row == 0u && lane == 0u) gp->giant_part[(size_t)batch * g.n_giant + gi] = (uint32_t)sweep + 1u;
}

// The sweep's visit of a giant row: the slices' accumulators of THIS sweep merged in row order.  false: k_giant_part did
// not evaluate the row for this sweep (it was stamped after that launch had looked) — the row stays stamped and is due
// in the next sweep, which exists because whoever stamped it has set this sweep's flag.
template <typename ST, bool HC>
__device__ __forceinline__ bool fused_row_giant(const FusedGraph *gp, uint32_t v, uint32_t batch, uint32_t batches,
                                                int sweep, uint32_t lane, uint32_t my_root, uint32_t v_router,
                                                const FusedParams &P, RowOut<ST> &out) {
  const GraphDev &g = gp->g;
  const uint32_t gi = giant_index(g, v);
  if (gp->giant_part[(size_t)batch * g.n_giant + gi] != (uint32_t)sweep + 1u) return false;
  const uint32_t s0 = g.giant_slice0[gi], s1 = g.giant_slice0[gi + 1], n_slices = g.giant_slice0[g.n_giant];
  const uint32_t *part = gp->giant_part + giant_tag_words(batches, g.n_giant) + ((size_t)batch * n_slices + s0) * (GIANT_WORDS * 64u);
  AnyAcc x = any_acc_init();
#pragma unroll 2
  for (uint32_t sidx = s0; sidx < s1; ++sidx, part += GIANT_WORDS * 64u) {
    const uint32_t w3 = part[3 * 64 + lane];
    AnyAcc y{RowAcc{part[0 * 64 + lane], part[1 * 64 + lane], part[2 * 64 + lane], w3 & 0x7FFFFFFFu, (w3 >> 31) != 0u},
             part[4 * 64 + lane], part[5 * 64 + lane], part[6 * 64 + lane], part[7 * 64 + lane]};
    any_merge(x, y);
  }
  out = any_finish<ST, HC>(x, v, my_root, v_router, P);
  return true;
}

// Wave schedule (what the sweep time is made of: SIMD utilisation = resident waves x compute /
// (compute + dependent memory round trips), measured 42 % for the first versions whose chain was 12
// round trips per wave): ONE round trip for everything that only depends on the wave's vertex
// range (stamps, row bounds, flags), ONE for the link vectors, old states and out-link vectors of
// all its active rows, then one per row for the neighbour rows; state stores and wake-up stamps
// are fire-and-forget (the out-link vector was fetched up front, speculatively).
// A wave owns FQ quads of VPW consecutive vertices (a block = 4 waves x FQ x VPW vertices): the
// wave-level set-up (ids, descriptors, activation stamps of all its vertices in ONE load) is paid
// once per FQ x VPW rows and a sweep launches FQ x fewer waves — the floor of a sweep with little
// to do is the cost of launching and retiring its waves (profiles/r01_notes.md).
constexpr int FQ = 1;          // measured: FQ = 4 makes sparse sweeps cheaper (tail 6 vs 10 us) but dense ones slower (50 vs 39.5 us: 4x fewer waves to hide latency)
constexpr int FVPW = FQ * VPW;                    // vertices per wave (<= 63: one lane per vertex + 1)
constexpr int FVPB = WAVES_PER_BLOCK * FVPW;      // vertices per block of the fused kernel

template <typename ST, bool MAXINF, bool COUNT, bool UNITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void k_fused(
    const FusedGraph *__restrict__ gp, int *changed, int sweep, uint32_t *__restrict__ act,
    const uint8_t *__restrict__ hnb, uint32_t n_arg, const uint32_t *__restrict__ a_in_ptr, const uint32_t *__restrict__ a_out_ptr,
    const uint8_t *__restrict__ a_vflags, ST *__restrict__ st, const uint32_t *__restrict__ roots, uint32_t *lane_flags,
    uint32_t net_nexthops, uint32_t ignore_ovl, FusedParams P, const uint32_t *__restrict__ a_in_src,
    const uint32_t *__restrict__ a_in_w, const uint32_t *__restrict__ a_out_dst, uint32_t a_e_in) {
  // Argument order: the first 16 dwords are preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count=16,
  // holo_amd/build.py): the flag array of the sweeps, the vertex count and the arrays of the first round trip arrive
  // without a load — before, every wave fetched its arguments, then the graph descriptor, then issued round trip 1
  // (measured: 75.6 k -> 78.5 k runs/s; computing equal XCD ranges from n alone instead of reading xcd_start added nothing).
  // The rest of the graph / slot-table descriptors stays in device memory and is fetched (scalar loads) where it is
  // used: as by-value arguments they pinned ~40 SGPRs, and above 96 SGPRs a CU admits 6 of these workgroups instead of 8.
  if (sweep > 0 && changed[sweep - 1] == 0) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t batch = blockIdx.y;
  // UNITS: the graph has heavy chunks and the blocks go through the work-unit table (GraphDev::unit_first); its own
  // instantiation, so that the common case keeps its compile-time row count (measured: a run-time one costs 3 %)
  const uint32_t n = n_arg;
  uint32_t wbeg, nrows_rt = (uint32_t)FVPW;
  if (UNITS) {
    if (!wave_rows(gp->g, blockIdx.x, wave, wbeg, nrows_rt)) return;
  } else {
    const uint32_t chunk = xcd_chunk(gp->g.xcd_start, blockIdx.x);
    if (chunk == 0xFFFFFFFFu) return;
    wbeg = chunk * FVPB + wave * FVPW;
    if (wbeg >= n) return;
  }
  const uint32_t nrows = UNITS ? nrows_rt : (uint32_t)FVPW;
  uint32_t *A = act + (size_t)batch * n;
  const uint32_t cur = (uint32_t)sweep + 2u;
  // ---- round trip 1: stamps, row bounds, flags of ALL the wave's vertices (lanes 0..FVPW)
  const uint32_t vl = min(wbeg + min(lane, (uint32_t)FVPW), n);   // in_ptr / out_ptr have n+1 (+16) entries
  const uint32_t vlc = min(vl, n - 1);
  const uint32_t av = A[vlc];
  const uint32_t pv = a_in_ptr[vl];
  const uint32_t po = a_out_ptr[vl];
  const uint32_t hb = hnb[(size_t)batch * n + vlc] & (ignore_ovl ? ~RF_NT : ~0u);
  const uint32_t vf = a_vflags[vlc];
  const uint64_t due = __ballot(lane < nrows && wbeg + lane < n && av >= cur);
  if (due == 0ull) return;
  const uint32_t root_slot = batch * 64 + lane;
  const uint32_t my_root = roots[root_slot];
  ST *S = st + (size_t)batch * n * 64;
  const __amdgpu_buffer_rsrc_t rs = st_rsrc(S, n << StIO<ST>::ROW_SHIFT);
  const __amdgpu_buffer_rsrc_t ra = st_rsrc(A, n * 4u);
  const uint32_t ebytes = (a_e_in + 16u) * 4u;                   // link arrays: arguments too (fetched with the others at
  const __amdgpu_buffer_rsrc_t rsrc_src = st_rsrc(a_in_src, ebytes);   // wave start, not through the descriptor between
  const __amdgpu_buffer_rsrc_t rsrc_w = st_rsrc(a_in_w, ebytes);       // round trips 1 and 2)
  const __amdgpu_buffer_rsrc_t rsrc_od = st_rsrc(a_out_dst, ebytes);
  const uint32_t lvo = lane * (uint32_t)sizeof(ST);
  const uint32_t lane4 = lane * 4u;
  bool any = false, sat = false, dyn = false, ovf = false;
  uint32_t n_done = 0;                                            // COUNT only: rows this wave evaluated
#pragma unroll 1
  for (uint32_t q = 0; q < (uint32_t)FQ; ++q) {
    if (((due >> (q * VPW)) & ((1ull << VPW) - 1ull)) == 0ull) continue;   // nothing due in this quad
    const uint32_t vbeg = wbeg + q * VPW;
    const uint32_t lb = q * VPW;                                  // lane of the quad's first vertex
    // ---- round trip 2: link vectors (first 64 links), old state, out-link vector of the quad's
    // rows: unconditional (rows that are not due simply are not used): 16 requests in flight, no
    // control flow; reads past a row's end stay inside the padded arrays and are masked afterwards
    uint32_t svv[VPW], wvv[VPW], odv[VPW];
    typename StIO<ST>::Raw oldq[VPW];
    {
      uint32_t ta[VPW], tb[VPW], tc[VPW];
#pragma unroll
      for (int i = 0; i < VPW; ++i) {
        const uint32_t e0 = rdlane(pv, lb + i), o0 = rdlane(po, lb + i);
        ta[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_src, lane4, e0 * 4u, 0);
        tb[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_w, lane4, e0 * 4u, 0);
        tc[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_od, lane4, o0 * 4u, 0);
        oldq[i] = StIO<ST>::ld(rs, lvo, min(vbeg + i, n - 1) << StIO<ST>::ROW_SHIFT);
      }
#pragma unroll
      for (int i = 0; i < VPW; ++i) {
        const uint32_t cnt = min(64u, rdlane(pv, lb + i + 1) - rdlane(pv, lb + i));
        const uint32_t ocnt = min(64u, rdlane(po, lb + i + 1) - rdlane(po, lb + i));
        svv[i] = lane < cnt ? ta[i] : min(vbeg + i, n - 1);
        wvv[i] = lane < cnt ? tb[i] : INF;
        odv[i] = lane < ocnt ? tc[i] * 4u : 0xFFFFFFFFu;          // byte offset into A; out of range = dropped
      }
    }
    auto row = [&](auto I) {                                      // explicit 4x instantiation
      constexpr int i = decltype(I)::value;
      const uint32_t v = vbeg + i;
      if (v >= n || (uint32_t)i >= nrows) return;
      if (rdlane(av, lb + i) < cur) return;                       // nothing changed around this row
      const uint32_t e0 = rdlane(pv, lb + i), e1 = rdlane(pv, lb + i + 1);
      const uint32_t v_router = (rdlane(vf, lb + i) & 1u) ? 0u : 1u;
      RowOut<ST> r;
      if (rdlane(hb, lb + i) == 0u)
        r = fused_row16<ST, MAXINF>(rs, v, e1 - e0, svv[i], wvv[i], lvo, my_root, v_router, P);
      else if (UNITS && (rdlane(hb, lb + i) & RF_GIANT) && gp->giant_part != nullptr) {   // slices evaluated by k_giant_part before this launch
        const bool got = P.hc ? fused_row_giant<ST, true>(gp, v, batch, gridDim.y, sweep, lane, my_root, v_router, P, r)
                              : fused_row_giant<ST, false>(gp, v, batch, gridDim.y, sweep, lane, my_root, v_router, P, r);
        if (!got) return;
      } else if (P.hc)       // hop-count-like graphs only (MANET): its own instantiation keeps the usual one lean
        r = fused_row_any<ST, MAXINF, true, 4>(gp->g, rs, v, e0, e1, svv[i], wvv[i], lane, lvo, my_root, root_slot, gp->tabs, net_nexthops, ignore_ovl, P);
      else
        r = fused_row_any<ST, MAXINF, false, 4>(gp->g, rs, v, e0, e1, svv[i], wvv[i], lane, lvo, my_root, root_slot, gp->tabs, net_nexthops, ignore_ovl, P);
      sat = sat || r.sat;
      dyn = dyn || r.dyn;
      ovf = ovf || r.ovf;
      const bool ch = r.nw != StIO<ST>::bits(oldq[i]);
      if (ch) { S[(size_t)v * 64 + lane] = r.nw; any = true; }
      if (COUNT) ++n_done;
      if (__ballot(ch) != 0ull) {                                 // wake the out-neighbours up
        __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, odv[i], 0, 0);
        const uint32_t o0 = rdlane(po, lb + i), o1 = rdlane(po, lb + i + 1);
        for (uint32_t ob = o0 + 64u + lane; ob < o1; ob += 64)    // rows with more than 64 out-links
          __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, gp->g.out_dst[ob] * 4u, 0, 0);
      }
    };
    static_assert(VPW == 4, "row() is instantiated four times");
    row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
  }
  if (__ballot(any) != 0ull && lane == 0) changed[sweep] = 1;
  if (COUNT && lane == 0) atomicAdd(&gp->rows_done[(blockIdx.x + wave) & 255u], n_done);
  uint32_t lf = 0;
  if (MAXINF && sat) lf |= LF_NEED_EXACT;
  if (ovf) lf |= LF_OVERFLOW;
  if (lf) atomicOr(&lane_flags[root_slot], lf);
  if (dyn) raise_dyn(gp, lane_flags, root_slot, blockIdx.x * 4u + wave);
}

// ---------------------------------------------------------------------------------------------
// k_fused_lean — the sweep of the 4-byte fused state, rewritten for instruction count AND memory-level parallelism.
//
// Round 3 measured what a dense sweep of k_fused is made of (profiles/r03_notes.md): 237 instructions per row evaluation
// for ~10 links — ~9 per link (two v_readlane, the load, a wait per link, add, min, compare, two selects, or) and ~150
// of per-row overhead —, a wave's life = six dependent round trips with at most ~10 row loads out, and an L1 that stalls
// on pending misses half of the time because a CU never has enough of them in flight.  This kernel:
//   * link records in a fixed-stride (ELL) copy built with the graph (kb_ell): 16 source byte offsets + 16 costs + 16
//     wake-up offsets per vertex at addresses that depend on the vertex alone.  The source offsets come through the
//     SCALAR cache (s_load) straight into the SGPRs the buffer loads take as row offset: no v_readlane, no hazard s_nop;
//     the costs are one vector load per row with the 16 costs replicated in every 16-lane DPP row, so that
//     `v_add_u32_dpp ... row_newbcast:j` adds cost j to a neighbour's word in ONE instruction;
//   * a DPP add cannot saturate, so "not reached" is stored as inf_t = (dmax - wmax) << sh instead of all ones: a sum
//     with any cost stays inside 32 bits, and whatever reaches inf_t is "not reached" again;
//   * ONE s_waitcnt per row (explicit: the compiler's own is one per link); rows padded to an even number of links with a
//     link from a never-reached pad row (state row n of every batch slab): 9 straight-line cases instead of 16;
//   * a 3-bit TAG field between distance and hops (zero in stored words): the cost vector adds j & 7 there, so the min
//     over whole candidate words is (smallest distance, first link in row order, ITS hops) at once and the second pass is
//     min + or per link instead of compare + two selects (lean_reduce);
//   * field checks out of the loop: the hop field saturates (one v_min) and k_emit_fused tests every FINAL word once
//     (a clipped transient is a weaker upper bound and harmless); no root test (root rows carry RF_HNB and take the
//     general routine); a changed row is stored whole (unchanged lanes rewrite their own value: single writer);
//   * per-wave masks (due, fast) as SGPR bit sets instead of per-row lane read-backs; the wake-up offsets of a row are
//     one ELL load in the set-up round trip (no out-row bounds, no dependent out-link load).
// Rows with a per-batch or static flag take k_fused's general routine on the CSR arrays, unchanged — except the rows
// next to a root of the batch and the roots' own rows (RF_HNB / RF_ROOT and nothing else), which have a path of their own
// below: 1 % of the rows, and through the general routine 14 % of the run, because a sweep ends with its LAST wave.
// What moved the run after the per-row cost had stopped mattering (profiles/r03_notes.md r03k, r03o, r03q): that path;
// launches WITHOUT activation stamps for the sweeps in which nearly every row is due (MODE, learned schedule); and a
// whole stretch of such sweeps as ONE launch of several passes (pass_blocks), so that a pass starts while the one
// before it drains instead of behind a kernel boundary.
template <int J> __device__ __forceinline__ uint32_t dpp_bcast16(uint32_t v) {     // lane 16 r + J of every DPP row r
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + J, 0xF, 0xF, false);
}

struct LeanOut { uint32_t m, macc, bpay; };

template <int J> struct LeanLinks {
  // sov: lane 16 r + j = row byte offset of link j's source (low byte of link 0: the row's info byte, masked by the caller)
  static __device__ __forceinline__ void load(uint32_t (&q)[16], __amdgpu_buffer_rsrc_t rs, uint32_t lane4, uint32_t sov) {
    LeanLinks<J - 1>::load(q, rs, lane4, sov);
    q[J - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4, rdlane(sov, J - 1), 0);
  }
  static __device__ __forceinline__ void add(uint32_t (&q)[16], uint32_t wk) {
    LeanLinks<J - 1>::add(q, wk);
    q[J - 1] = dpp_bcast16<J - 1>(wk) + q[J - 1];
  }
};
template <> struct LeanLinks<0> {
  static __device__ __forceinline__ void load(uint32_t (&)[16], __amdgpu_buffer_rsrc_t, uint32_t, uint32_t) {}
  static __device__ __forceinline__ void add(uint32_t (&)[16], uint32_t) {}
};

// Candidates c[0 .. L) of one row -> (word of the FIRST tight link in row order, OR of the tight links' words).
// The state word is [dist | tag (3 zero bits) | hops | mask]; the cost vector carries (cost << P.sh) + (j & 7) << tag
// position, so a candidate is [dist | j & 7 | hops | mask] and ONE min over whole words yields the smallest distance and,
// among the links that attain it, the first in row order together with its hop count — k_fused's second pass (compare +
// two selects per link) shrinks to min(c, threshold) + or.  Rows of more than 8 links: two halves of 8, the second one
// wins only with a strictly smaller distance.
template <int L>
__device__ __forceinline__ LeanOut lean_minor(uint32_t (&c)[16], uint32_t paym);
template <int L>
__device__ __forceinline__ LeanOut lean_reduce(uint32_t (&c)[16], uint32_t wk, uint32_t paym) {
  LeanLinks<L>::add(c, wk);
  return lean_minor<L>(c, paym);
}
template <int L>
__device__ __forceinline__ LeanOut lean_minor(uint32_t (&c)[16], uint32_t paym) {
  uint32_t m = INF;
#pragma unroll
  for (int j = 0; j < (L < 8 ? L : 8); ++j) m = min(m, c[j]);
  if (L > 8) {
    uint32_t mb = INF;
#pragma unroll
    for (int j = 8; j < L; ++j) mb = min(mb, c[j]);
    m = mb < (m & ~paym) ? mb : m;
  }
  const uint32_t thr1 = (m | paym) + 1u;                          // (all ones + 1 = 0: nothing reached, nothing tight)
  uint32_t macc = 0u;
#pragma unroll
  for (int j = 0; j < L; ++j) macc |= min(c[j], thr1);            // a tight link: its word; the others: low bits zero
  return LeanOut{m, macc, m};
}

template <int L>
__device__ __forceinline__ LeanOut lean_case(__amdgpu_buffer_rsrc_t rs, uint32_t lane4, uint32_t sov, uint32_t wk, uint32_t paym) {
  uint32_t c[16];
  LeanLinks<L>::load(c, rs, lane4, sov);
  __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): one wait for the row
  return lean_reduce<L>(c, wk, paym);
}

// The rows of one wave's group (up to four consecutive rows of one batch) evaluated and committed: the fast rows through
// the straight-line cases, the rows next to a root through their own path, flagged rows through k_fused's general routine.
// (A function of its own since round 4's experiment with a persistent dense kernel — every wave on its own rows for the
// whole stretch, link records staged once in LDS: 25.6 us per pass instead of 20.0 and a stretch that converges like plain
// Jacobi, because it loses what the in-order grid gives for free: a row reads the lower-numbered neighbours of the SAME
// pass.  profiles/r04_notes.md r04t; removed.)  Everything is passed in registers.
template <bool COUNT, int MODE>
__device__ __forceinline__ void lean_group(const FusedGraph *__restrict__ gp, const __amdgpu_buffer_rsrc_t rs, const __amdgpu_buffer_rsrc_t ra,
                                           const uint32_t lane, const uint32_t lane4, const uint32_t wbeg, const uint32_t cur, const FusedParams &P,
                                           const uint32_t *__restrict__ roots, const uint32_t root_slot, const uint32_t net_nexthops,
                                           const uint32_t ignore_ovl, const uint32_t due4, const uint32_t fast4, const uint32_t fasth4, const uint32_t fastz4,
                                           uint32_t (&sov)[VPW], uint32_t (&wk)[VPW], uint32_t (&od)[VPW], uint32_t (&oldq)[VPW], uint32_t (&info)[VPW],
                                           uint64_t &any, bool &dyn, uint32_t &n_done, uint32_t &n_chg) {
  typedef uint32_t ST;
  const uint32_t paym = (1u << P.sh) - 1u;
  const uint32_t hopm = P.hmax << P.mbits, maskm = (1u << P.mbits) - 1u;
  // result of fast row i -> state, wake-ups (info bit 5: more than 16 out-links: they are walked, k_fused's loop)
  auto commit = [&](auto I, uint32_t nw) {
    constexpr int i = decltype(I)::value;
    const uint32_t v = wbeg + i;
    if (COUNT) ++n_done;
    const uint64_t ch = __ballot(nw != oldq[i]);
    if (ch == 0ull) return;
    __builtin_amdgcn_raw_buffer_store_b32(nw, rs, lane4, v << 8, 0);          // the whole row: unchanged lanes rewrite their own value
    any |= ch;
    if (MODE == 1) { ++n_chg; return; }                                       // dense: nobody reads stamps
    if (!(info[i] & 0x20u)) { __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, od[i], 0, 0); return; }   // wake the out-neighbours up
    const GraphDev &g = gp->g;
    const uint32_t o0 = g.out_ptr[v], o1 = g.out_ptr[v + 1];
    for (uint32_t ob = o0 + lane; ob < o1; ob += 64) __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, g.out_dst[ob] * 4u, 0, 0);
  };
  auto finish = [&](const LeanOut &r, uint32_t inf) -> uint32_t {
    const uint32_t hinc = (inf & 0x80u) ? 0u : (1u << P.mbits);   // a network does not count as a hop
    const uint32_t hops = min((r.bpay & hopm) + hinc, hopm);      // saturating: P.hmax in a FINAL word = overflow (k_emit_fused)
    return min((r.m & ~paym) | hops | (r.macc & maskm), P.infw);  // a sum that reached inf_t: not reached
  };
  auto row = [&](auto I) {                                        // a due fast row: even number of links (pad link from the pad row)
    constexpr int i = decltype(I)::value;
    if (!((fast4 >> i) & 1u)) return;
    LeanOut r{INF, 0u, INF};
    switch (((info[i] & 0x1Fu) + 1u) >> 1) {
      case 0: break;
      case 1: r = lean_case<2>(rs, lane4, sov[i], wk[i], paym); break;
      case 2: r = lean_case<4>(rs, lane4, sov[i], wk[i], paym); break;
      case 3: r = lean_case<6>(rs, lane4, sov[i], wk[i], paym); break;
      case 4: r = lean_case<8>(rs, lane4, sov[i], wk[i], paym); break;
      case 5: r = lean_case<10>(rs, lane4, sov[i], wk[i], paym); break;
      case 6: r = lean_case<12>(rs, lane4, sov[i], wk[i], paym); break;
      case 7: r = lean_case<14>(rs, lane4, sov[i], wk[i], paym); break;
      default: r = lean_case<16>(rs, lane4, sov[i], wk[i], paym); break;
    }
    commit(I, finish(r, info[i]));
  };
  static_assert(VPW == 4, "row() is instantiated four times");
  // (Round 6, VERDICT r05 item 4: the gathers of ALL FOUR rows issued back to back, one wait per row in issue order, was built
  // and measured — HSPF_LEAN_ALLROWS, profiles/r06_notes.md r06n: 126.4 k against 144.3 k runs/s in flight on the same box, 92.4 k
  // against 114.0 k one at a time; 18 dense passes instead of 15 and 20.4 x N rows instead of 17.4-18.1, the time per row
  // unchanged.  The wave's four dependent row trips are what keeps a row behind the row before it in the SAME pass; removed.)
  {
  row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{});
  row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
  }
  // ---- rows next to a root (64 roots x ~14 neighbours = 1 % of the rows, and through the general routine 14 % of the
  // run: each kept its wave ~10 us behind the others).  A root's word is all zero in its own lane, so the candidate
  // through the link from the root is [cost | tag | hops 0 | mask 0] there, and what fused_row_any does for a hops-0
  // parent — the link's own first-hop slot instead of the parent's mask — is ONE or of `1 << position` into that lane's
  // candidate (the root's slot base is 0).  All 16 record slots are walked (the pad links never win): one code instance.
  if (fasth4 != 0u) {
    const uint32_t my_root = roots[root_slot];
#pragma unroll 1
    for (uint32_t hm = fasth4; hm != 0u; hm &= hm - 1u) {
      const uint32_t i = (uint32_t)__builtin_ctz(hm);
      const uint32_t v = wbeg + i;
      const uint32_t sov_i = i == 0 ? sov[0] : i == 1 ? sov[1] : i == 2 ? sov[2] : sov[3];
      const uint32_t wk_i = i == 0 ? wk[0] : i == 1 ? wk[1] : i == 2 ? wk[2] : wk[3];
      const uint32_t od_i = i == 0 ? od[0] : i == 1 ? od[1] : i == 2 ? od[2] : od[3];
      const uint32_t old_i = i == 0 ? oldq[0] : i == 1 ? oldq[1] : i == 2 ? oldq[2] : oldq[3];
      const uint32_t inf_i = i == 0 ? info[0] : i == 1 ? info[1] : i == 2 ? info[2] : info[3];
      const uint32_t deg = inf_i & 0x1Fu;                                 // <= 16 (a longer row carries RF_MANY)
      const uint32_t e0 = gp->g.in_ptr[v];
      const uint32_t fpv = gp->g.in_fpos[e0 + min(lane & 15u, deg - 1u)];  // lane j (< 16) = position of link j in its source row
      uint32_t c[16];
      LeanLinks<16>::load(c, rs, lane4, sov_i);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      LeanLinks<16>::add(c, wk_i);
      const bool slots_on = !(inf_i & 0x80u) || net_nexthops != 0u;       // next hops of a network vertex only on request
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint32_t u = rdlane(sov_i, j) >> 8, fp = rdlane(fpv, j);
        const uint32_t bit = (slots_on && (uint32_t)j < deg && fp < P.mbits) ? (1u << fp) : 0u;
        c[j] |= (my_root == u) ? bit : 0u;
      }
      const LeanOut r = lean_minor<16>(c, paym);
      const uint32_t nw = my_root == v ? 0u : finish(r, inf_i);          // a root's own lane: distance 0, hops 0, no next hops
      if (COUNT) ++n_done;
      const uint64_t ch = __ballot(nw != old_i);
      if (ch == 0ull) continue;
      __builtin_amdgcn_raw_buffer_store_b32(nw, rs, lane4, v << 8, 0);
      any |= ch;
      if (MODE == 1) { ++n_chg; continue; }
      if (!(inf_i & 0x20u)) { __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, od_i, 0, 0); continue; }
      const GraphDev &g = gp->g;
      const uint32_t o0 = g.out_ptr[v], o1 = g.out_ptr[v + 1];
      for (uint32_t ob = o0 + lane; ob < o1; ob += 64) __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, g.out_dst[ob] * 4u, 0, 0);
    }
  }
  // ---- rows whose only flag is RF_ZERO (a zero-cost link from a higher- or equal-numbered source; round 6): with 1 % of the
  // links at cost 0 that is 5 % of the rows, and through the general routine below a dense pass took 3.5 x as long (a wave
  // with one such row lives three times as long and a pass ends with its last wave).  The rule of finish_row_z on the
  // candidate words: the zero-cost links from sources >= v are taken OUT of the row's minimum (an all-ones candidate never
  // wins and is never tight); the first SAFE one with the smallest distance (whole-word minimum: distance, then row order) is
  // the row's one parent when it beats the rest — its hops and ITS mask, no union —, an UNSAFE one that beats everything
  // leaves the distance alone (hops 0, mask 0: the placeholder k_repair fills in).  Either way the lane's root is LF_DYN.
  if (fastz4 != 0u) {
    const uint8_t *__restrict__ zc = gp->g.zcyc;
#pragma unroll 1
    for (uint32_t hm = fastz4; hm != 0u; hm &= hm - 1u) {
      const uint32_t i = (uint32_t)__builtin_ctz(hm);
      const uint32_t v = wbeg + i;
      const uint32_t sov_i = i == 0 ? sov[0] : i == 1 ? sov[1] : i == 2 ? sov[2] : sov[3];
      const uint32_t wk_i = i == 0 ? wk[0] : i == 1 ? wk[1] : i == 2 ? wk[2] : wk[3];
      const uint32_t od_i = i == 0 ? od[0] : i == 1 ? od[1] : i == 2 ? od[2] : od[3];
      const uint32_t old_i = i == 0 ? oldq[0] : i == 1 ? oldq[1] : i == 2 ? oldq[2] : oldq[3];
      const uint32_t inf_i = i == 0 ? info[0] : i == 1 ? info[1] : i == 2 ? info[2] : info[3];
      const uint32_t deg = inf_i & 0x1Fu;                                 // <= 16 (a longer row carries RF_MANY)
      const uint32_t src_l = sov_i >> 8;                                  // lane 16 r + j: source of link j
      const bool zl = (lane & 15u) < deg && (wk_i >> P.sh) == 0u && src_l >= v;
      const bool ul = zl && (zc == nullptr || (zc[src_l] != 0 && zc[v] != 0));
      const uint32_t zmask = (uint32_t)__ballot(zl) & 0xFFFFu, umask = (uint32_t)__ballot(ul) & 0xFFFFu;
      uint32_t c[16];
      LeanLinks<16>::load(c, rs, lane4, sov_i);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      LeanLinks<16>::add(c, wk_i);
      uint32_t cz = INF, du = INF;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (!((zmask >> j) & 1u)) continue;                               // uniform
        if ((umask >> j) & 1u) du = min(du, c[j] & ~paym); else cz = min(cz, c[j]);
        c[j] = INF;
      }
      const LeanOut r = lean_minor<16>(c, paym);
      const uint32_t d_reg = r.m & ~paym, d_zs = cz & ~paym;
      const bool unsafe_wins = du < min(d_reg, d_zs), late = !unsafe_wins && d_zs < d_reg;
      const uint32_t nw = unsafe_wins ? min(du, P.infw) : finish(late ? LeanOut{cz, cz, cz} : r, inf_i);
      dyn = dyn || (unsafe_wins && du < P.inf_t) || (late && d_zs < P.inf_t);
      if (COUNT) ++n_done;
      const uint64_t ch = __ballot(nw != old_i);
      if (ch == 0ull) continue;
      __builtin_amdgcn_raw_buffer_store_b32(nw, rs, lane4, v << 8, 0);
      any |= ch;
      if (MODE == 1) { ++n_chg; continue; }
      if (!(inf_i & 0x20u)) { __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, od_i, 0, 0); continue; }
      const GraphDev &g = gp->g;
      const uint32_t o0 = g.out_ptr[v], o1 = g.out_ptr[v + 1];
      for (uint32_t ob = o0 + lane; ob < o1; ob += 64) __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, g.out_dst[ob] * 4u, 0, 0);
    }
  }
  // ---- the due rows that carry a flag: k_fused's general routine on the CSR arrays (rare: one code instance, run-time row)
#pragma unroll 1
  for (uint32_t gm = due4 & ~fast4 & ~fasth4 & ~fastz4; gm != 0u; gm &= gm - 1u) {
    const uint32_t v = wbeg + (uint32_t)__builtin_ctz(gm);
    const GraphDev &g = gp->g;
    const uint32_t e0 = g.in_ptr[v], e1 = g.in_ptr[v + 1];
    const uint32_t cnt = min(64u, e1 - e0);
    const uint32_t sv = lane < cnt ? g.in_src[e0 + lane] : v;
    const uint32_t wv = lane < cnt ? g.in_w[e0 + lane] : INF;
    const uint32_t my_root = roots[root_slot];
    const uint32_t old = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4, v << 8, 0);
    RowOut<ST> r;
    if (P.hc) r = fused_row_any<ST, false, true, 4>(g, rs, v, e0, e1, sv, wv, lane, lane4, my_root, root_slot, gp->tabs, net_nexthops, ignore_ovl, P);
    else r = fused_row_any<ST, false, false, 4>(g, rs, v, e0, e1, sv, wv, lane, lane4, my_root, root_slot, gp->tabs, net_nexthops, ignore_ovl, P);
    dyn = dyn || r.dyn;       // (r.ovf is a per-evaluation test of TRANSIENT values: the lean state's fields are tested on the final words, k_emit_fused)
    if (COUNT) ++n_done;
    const uint64_t ch = __ballot(r.nw != old);
    if (ch == 0ull) continue;
    __builtin_amdgcn_raw_buffer_store_b32(r.nw, rs, lane4, v << 8, 0);
    any |= ch;
    if (MODE == 1) { ++n_chg; continue; }
    const uint32_t o0 = g.out_ptr[v], o1 = g.out_ptr[v + 1];
    for (uint32_t ob = o0 + lane; ob < o1; ob += 64) __builtin_amdgcn_raw_buffer_store_b32(cur + 1u, ra, g.out_dst[ob] * 4u, 0, 0);
  }
}

// MODE (per launch; the host enqueues a PLAN — head sweeps, dense stretch, one all-due sweep, tail sweeps — and the
// launches decide ON THE DEVICE, from counters earlier launches / passes left in `ctl`, whether they still have a job):
//   0  stamped: a row is due when an in-neighbour changed in the previous sweep (activation stamps), the form above;
//      HEAD (the sweeps before the dense stretch): a sample of the waves adds its due rows to ctl[LEAN_CTL_DUE + sweep];
//      a head sweep whose predecessor counted at least `thr` (sampled units) — or was itself skipped: the sentinel —
//      does nothing but pass the sentinel on: the frontier covers the graph, the dense stretch is next;
//   1  dense:   every row is evaluated, no stamp is read or written — in the middle of a run (sweeps ~4-19 of 28 on
//               isis-100k) every row IS due, and the stamps cost a dependent round trip at the head of every wave
//               (stamps before records), a 64-byte offset row per vertex and a scattered store per changed row; a
//               sample of the waves adds its changed rows to ctl[LEAN_CTL_PCH + pass], and pass p does nothing when
//               pass p - 2 or p - 3 counted fewer than `thr`: the stretch ends where the corrections thin out, wherever
//               the host guessed its end (a first run on an unknown graph guesses long);
//   2  all due, stamped: the first sweep after a dense stretch (the stamps are stale: every row is evaluated, changed rows
//               stamp their out-neighbours again, and mode 0 can follow).
// Any plan and any outcome of the device-side decisions is correct: modes 1 and 2 evaluate a superset of the due rows, a
// skipped launch or pass sets its `changed` flag so that the chain of launches reaches the all-due sweep, and the run
// ends only behind a stamped (or all-rows) launch that changed nothing.  No rehearsal run, nothing learned per graph:
// the first run of a fresh context takes the same path as the thousandth (round 3 LEARNed a schedule on the second
// identical run and lost it with every structural patch).
// Every counter has a 128-byte line of its own (LEAN_CTL_STRIDE words): a pass reads the counters of the passes two and
// three before it with PLAIN loads — the line of pass p - 2 is touched for the first time when pass p starts, i.e. when
// pass p - 2 is over, so the first fetch of an XCD's L2 brings the final count and every later one may hit.  (Counters
// sharing a line would be served stale from L2 for the rest of the launch; agent-scope loads by every wave — 50 000 per
// pass at ONE address — made a dense pass 4x slower, profiles/r04_notes.md r04a.)
constexpr uint32_t LEAN_CTL_STRIDE = 32u;
constexpr uint32_t LEAN_CTL_DUE = 0u, LEAN_CTL_PCH = 64u * LEAN_CTL_STRIDE, LEAN_CTL_WORDS = 128u * LEAN_CTL_STRIDE;   // 64 head sweeps, 64 dense passes
constexpr uint32_t LEAN_SAMPLE = 32u;                                             // wave 0 of every 8th block of an XCD's range counts
constexpr uint32_t LEAN_SENTINEL = 0xFFFFFFFFu;
template <bool COUNT, int MODE, bool HEAD, bool BMAJ = false>
// (amdgpu_num_sgpr: 80 leaves 28 scalar spills in the prologue and ~10 reloads per row — v_readlane from a spill VGPR —, 96 keeps the same
// 8 waves per SIMD with 40 % fewer of them and 104 has none at 7 waves: 148.3 / 148.2 / 146.6 k runs/s in flight, 117.3 / 115.1 / 112.5 k
// one at a time; and TWENTY extra vector instructions per row cost 4 %, ten 1.5 %: a dense pass is not bound by VALU issue, although the
// counters show the vector ALUs ~85 % "busy" — profiles/r06_notes.md r06zf.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_fused_lean(
    const FusedGraph *__restrict__ gp, int *changed, int sweep, uint32_t *__restrict__ act,
    const uint8_t *__restrict__ hnb, uint32_t n_arg, const uint32_t *__restrict__ ell_so, const uint32_t *__restrict__ ell_w,
    uint32_t *__restrict__ st, const uint32_t *__restrict__ ell_od, const uint32_t *__restrict__ roots, uint32_t *lane_flags,
    uint32_t net_nexthops, uint32_t ignore_ovl, FusedParams P, uint32_t *__restrict__ ctl, uint32_t pass_blocks, uint32_t pass_batches,
    uint32_t pass_base, uint32_t thr) {
  typedef uint32_t ST;
  if (sweep > 0 && changed[sweep - 1] == 0) return;
  if (HEAD && sweep > 0 && ctl[LEAN_CTL_DUE + ((uint32_t)sweep - 1u) * LEAN_CTL_STRIDE] >= thr) {   // the dense stretch is due: pass the word on, keep the chain alive
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { ctl[LEAN_CTL_DUE + (uint32_t)sweep * LEAN_CTL_STRIDE] = LEAN_SENTINEL; changed[sweep] = 1; }
    return;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Dense launches may carry SEVERAL passes over the rows: a 1-D grid of passes x batches x blocks, pass p = blocks
  // [p G B, (p + 1) G B), batch-major inside a pass (so that a pass spans ALL batches of the call: 16 batches of a 10 000-
  // vertex graph make a pass of 10 000 workgroups, long enough to keep its successor behind it).
  // Blocks are dispatched in order, so a pass starts while the one before it drains — no kernel boundary between two dense
  // sweeps (a dense sweep keeps ~76 % of the wave slots busy: ramp and drain) — and reads what that pass has written
  // except in the rows still in flight; any interleaving is a valid chaotic iteration of the same monotone fixed point,
  // and the run's end is decided by stamped sweeps behind the stretch.  pass_blocks = G (0: one pass, 2-D grid), pass_batches = B.
  // BATCH-MAJOR placement (template BMAJ; dense launches of calls with at least 8 batches): consecutive
  // workgroups go round-robin to the 8 XCDs, so block f of a pass belongs to XCD f % 8 — which then owns the batches
  // x, x + 8, ... whole (all their rows, pass after pass) instead of an eighth of the rows of every batch: the state of a batch
  // of a 5 000-router area is 1.3 MB, two of them and the link records stay in the XCD's 4 MB L2 for the whole stretch,
  // and every neighbour row a wave reads was written on its own XCD (no stale line of another L2 in the way of the pass
  // order).  pass_blocks = row blocks of ONE batch in this form.  Speed only: any placement gives the same result.
  // (BMAJ: its own instantiation — the index arithmetic of both forms in one kernel cost the one-batch headline 3-4 %)
  const bool bmaj = BMAJ && MODE == 1;
  const uint32_t nbat = pass_batches;
  const uint32_t per_pass = bmaj ? 8u * ((nbat + 7u) >> 3) * pass_blocks : pass_blocks * nbat;      // blocks of one pass
  const uint32_t fpass = bmaj ? blockIdx.x % per_pass : 0u;
  const uint32_t batch = bmaj ? (fpass & 7u) + 8u * ((fpass >> 3) / pass_blocks)
                              : ((MODE == 1 && pass_blocks != 0u) ? (blockIdx.x / pass_blocks) % pass_batches : blockIdx.y);
  const uint32_t n = n_arg;
  const uint32_t bx = bmaj ? (fpass >> 3) % pass_blocks : ((MODE == 1 && pass_blocks != 0u) ? blockIdx.x % pass_blocks : blockIdx.x);
  // dense pass p of the stretch (counted across its launches): nothing to do when pass p - 2 or p - 3 saw the corrections
  // thin out (their counters: plain loads, one line each, see LEAN_CTL_STRIDE)
  const uint32_t pg = MODE == 1 ? pass_base + (pass_blocks != 0u ? blockIdx.x / per_pass : 0u) : 0u;
  uint32_t pc2 = LEAN_SENTINEL, pc3 = LEAN_SENTINEL;
  if (MODE == 1 && thr != 0u && pg >= 2u && pg < 64u) {                 // thr = 0 (HSPF_DENSE_STAY_PCT=0): every planned pass runs
    pc2 = ctl[LEAN_CTL_PCH + (pg - 2u) * LEAN_CTL_STRIDE];
    if (pg >= 3u) pc3 = ctl[LEAN_CTL_PCH + (pg - 3u) * LEAN_CTL_STRIDE];
  }
  const bool sampler = wave == 0u && (bmaj ? (bx & 7u) == 0u : ((bx >> 3) & 7u) == 0u);
  const uint32_t chunk = bmaj ? bx : xcd_chunk(gp->g.xcd_start, bx);
  if (MODE == 1 && min(pc2, pc3) < thr) {                       // a stopped pass must not read as "nothing left to do"
    if (bx == 0u && batch == 0u && threadIdx.x == 0) changed[sweep] = 1;
    return;
  }
  if (chunk == 0xFFFFFFFFu || (bmaj && batch >= nbat)) return;
  const uint32_t wbeg = chunk * (uint32_t)FVPB + wave * (uint32_t)FVPW;
  if (wbeg >= n) return;
  uint32_t *A = act + (size_t)batch * n;
  const uint32_t cur = (uint32_t)sweep + 2u;
  // ---- stamps and flags first: in the sparse head and tail of a run most waves end here, and what the sweep is short
  // of is vector memory instructions, not round trips (profiles/r03_notes.md)
  const uint32_t vl = min(wbeg + min(lane, (uint32_t)VPW - 1u), n - 1);
  const uint32_t av = MODE == 0 ? A[vl] : cur;
  const uint32_t hb = hnb[(size_t)batch * n + vl] & (ignore_ovl ? ~RF_NT : ~0u);
  const uint32_t due4 = (uint32_t)__ballot(lane < (uint32_t)VPW && wbeg + lane < n && av >= cur);
  if (MODE == 0 && due4 == 0u) return;
  if (HEAD && sampler && lane == 0 && (uint32_t)sweep < 64u) atomicAdd(&ctl[LEAN_CTL_DUE + (uint32_t)sweep * LEAN_CTL_STRIDE], (uint32_t)__builtin_popcount(due4));
  // ---- everything else whose address depends on the wave's vertex range alone (ELL row n exists: all pad)
  ST *S = st + (size_t)batch * (n + 1u) * 64;                      // slab of n + 1 rows: row n = the pad row (never reached)
  const __amdgpu_buffer_rsrc_t rs = st_rsrc(S, (n + 1u) << 8);
  const uint32_t lane4 = lane * 4u;
  uint32_t sov[VPW], wk[VPW], od[VPW], oldq[VPW];
#pragma unroll
  for (int i = 0; i < VPW; ++i) {
    const uint32_t e = (min(wbeg + i, n) * 16u + (lane & 15u)) * 4u;          // byte offset into the ELL arrays (n < 2^23)
    sov[i] = *(const uint32_t *)((const char *)ell_so + e);
    wk[i] = *(const uint32_t *)((const char *)ell_w + e);
    od[i] = MODE == 1 ? 0u : *(const uint32_t *)((const char *)ell_od + e);
    oldq[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4, min(wbeg + i, n - 1) << 8, 0);
  }
  const uint32_t fast4 = (uint32_t)__ballot(lane < (uint32_t)VPW && hb == 0u) & due4;
  const uint32_t fasth4 = (uint32_t)__ballot(lane < (uint32_t)VPW && (hb & ~RF_ROOT) == RF_HNB) & due4;   // next to a root of the batch (or the row of one), nothing else
  const uint32_t fastz4 = P.hc ? 0u : ((uint32_t)__ballot(lane < (uint32_t)VPW && hb == RF_ZERO) & due4);  // a zero-cost link from a higher-numbered source, nothing else
  const uint32_t root_slot = batch * 64 + lane;
  const __amdgpu_buffer_rsrc_t ra = st_rsrc(A, n * 4u);
  uint32_t info[VPW];                                             // low byte of ELL entry 0: in-degree | (> 16 out-links) << 5 | network << 7
#pragma unroll
  for (int i = 0; i < VPW; ++i) {
    info[i] = rdlane(sov[i], 0) & 0xFFu;
    sov[i] &= ~0xFFu;
    wk[i] = (wk[i] << P.sh) + ((lane & 7u) << (P.sh - 3u));       // cost in the distance field, j & 7 in the tag field below it
    od[i] = lane < 16u ? od[i] : 0xFFFFFFFFu;                     // byte offsets into A; pad / upper lanes: out of range = dropped
  }
  uint64_t any = 0ull;
  bool dyn = false;
  uint32_t n_done = 0, n_chg = 0;
  lean_group<COUNT, MODE>(gp, rs, ra, lane, lane4, wbeg, cur, P, roots, root_slot, net_nexthops, ignore_ovl, due4, fast4, fasth4, fastz4,
                          sov, wk, od, oldq, info, any, dyn, n_done, n_chg);
  // (Round 5, r05c: INNER iterations — the wave evaluating its rows again with the records it holds, no set-up and no record
  // round trip — were measured and rejected: an iteration that sees only its own rows' and some neighbouring waves' stores
  // converges worse than a pass behind a pass; 2 / 3 iterations per pass: 22.2 / 25.8 x N rows evaluated instead of 17.9,
  // 0.584 / 0.631 ms per run instead of 0.535.  profiles/r05_notes.md.)
  // A dense launch says "changed" ONCE per pass, whatever it did: the stretch is followed by the all-due sweep in any case (a dense
  // pass cannot end a run), and 25 000 waves per pass storing the same 1 to the same word were worth 2.4 % of the runs in flight
  // (three lanes' dense launches side by side: 143.4 / 144.0 -> 146.9 / 147.4 k runs/s on one box; profiles/r06_notes.md r06ze).
  if (MODE == 1) { if (bx == 0u && batch == 0u && threadIdx.x == 0) changed[sweep] = 1; }
  else if (any != 0ull && lane == 0) changed[sweep] = 1;
  if (COUNT && lane == 0) atomicAdd(&gp->rows_done[(blockIdx.x + wave) & 255u], n_done);
  if (MODE == 1 && sampler && lane == 0 && n_chg != 0u && pg < 64u) atomicAdd(&ctl[LEAN_CTL_PCH + pg * LEAN_CTL_STRIDE], n_chg);
  if (dyn) raise_dyn(gp, lane_flags, root_slot, blockIdx.x * 4u + wave);
}

// ---------------------------------------------------------------------------------------------
// k_single — small graphs: ONE workgroup per root, lane = vertex, the whole packed state of the run in LDS.
//
// run_area / compute_spt are called with one root per area / level (holo-ospf/src/spf.rs:540-542,
// holo-isis/src/spf.rs:527), mostly on LSDBs of tens to a few thousand routers — the reference's own benchmark is a
// 500-router area.  There the batched sweep engine is bound by kernel boundaries (51 launches x ~6 us on ospf-500,
// slower than one CPU core) and 63 of its 64 lanes idle.  This kernel runs the SAME fixed point — the row routine
// below is fused_row_any / finish_row with the row walked by one lane instead of one wave — inside one launch: thread
// t owns vertices t, t + 1024, ...; the state word of every vertex ([dist32 | hops | mask], the 8-byte form of the
// fused path, <= 24 first-hop slots) lives in LDS; a sweep recomputes every vertex from its in-links and ends at a
// workgroup barrier; the run ends after a sweep that changed nothing.  Reads of a neighbour's word race with its
// owner's write inside a sweep: one 8-byte LDS access per word, so a (distance, hops, mask) triple is never torn,
// and a stale triple is a triple the neighbour had earlier in the run — the same "memory is monotone" argument as
// k_fused.  Results go straight to the row-major output arrays (no transpose pass).  Many roots = many workgroups,
// each with its own LDS: a batch on a small graph uses this kernel too.
// Link records come from global memory every sweep: for the graphs this kernel is chosen for they stay in L1 / L2.
constexpr int SINGLE_THREADS = 1024;
constexpr uint32_t SINGLE_MAX_N = 8192;          // 64 KB of LDS state
constexpr uint32_t SINGLE_MAX_E = 65536;
constexpr size_t SINGLE_LDS_MAX = 160 * 1024 - 64;   // dynamic LDS a workgroup may ask for (one workgroup per CU then)

// The status word of a root in the one-workgroup kernels: the workgroup owns it, so it is reduced over the block (through
// a word of LDS the sweeps no longer need) and STORED once (SingleArgs::lane_flags may be pinned host memory: no memset
// before the launch, no copy after it).  No __syncthreads_or: its hidden LDS word would not fit next to 160 KB - 64.
__device__ __forceinline__ void single_store_flags(uint32_t *lane_flags, uint32_t root_slot, uint32_t lf, int *word) {
  __syncthreads();
  if (threadIdx.x == 0) *word = 0;
  __syncthreads();
  if (lf) atomicOr(word, (int)lf);
  __syncthreads();
  if (threadIdx.x == 0) lane_flags[root_slot] = (uint32_t)*word;
}

struct SingleArgs {
  const FusedGraph *gp;
  const uint32_t *roots;
  FusedParams P;               // the 8-byte state's parameters (sh = 0)
  uint32_t net_nexthops, ignore_ovl, n_roots, count_rows, lds_links;
  uint32_t *lane_flags;
  OutDev o;
  __device__ __forceinline__ const SlotTabs &slot_tabs() const { return gp->tabs; }
  __device__ __forceinline__ const uint8_t *zcyc() const { return gp->g.zcyc; }
};

// LDS layout (dynamic): state words [n] | when the links are staged, per link: source|NT, cost, position in the source's
// row (3 x u32 arrays).  The link records do not change during a run; staged once, a sweep touches global memory only
// for the slot base of a hops-0 NETWORK parent (the root's own base is 0).  Row bounds, vertex kind and "needs the
// general routine" of a thread's own vertices stay in registers for the whole run.
__host__ __device__ inline size_t single_lds_bytes(uint32_t n, uint32_t e, bool lds_links) {
  return (size_t)n * 8 + (lds_links ? (size_t)e * 12 : 0);
}

constexpr uint32_t SINGLE_VPT_MAX = SINGLE_MAX_N / SINGLE_THREADS;     // vertices per thread at most (8)
constexpr uint32_t SINGLE_RL = 8;                                      // in-links a thread keeps in registers
constexpr uint32_t SINGLE_FREE = 7;                                    // free-running sweeps between two checked ones

// One link into the row accumulator: the per-lane form of fused_row_any's loop body.  RARE = the row has an
// overloaded source, a zero-cost link from a higher-numbered source, or the graph is hop-count-like.
template <bool MAXINF, bool RARE, typename Args, typename FPos>
__device__ __forceinline__ void single_link(RowAcc &r, uint32_t &bd_all, uint32_t &zb, uint32_t &zm, uint32_t &zh,
                                            uint32_t sw, uint32_t w, FPos &&fpos_of, uint64_t q, uint32_t v, uint32_t v_router,
                                            uint32_t my_root, uint32_t root_slot, const Args &a, const FusedParams &P,
                                            uint32_t mmask) {
  const uint32_t u = sw & SRC_MASK;
  uint32_t d = (uint32_t)(q >> 32);
  const uint32_t pay = (uint32_t)q;
  if (RARE && !a.ignore_ovl && (sw & SRC_NO_TRANSIT) && u != my_root) d = INF;        // overloaded source
  const uint32_t c = add_sat(d, w);
  if (MAXINF && c == INF && d != INF) r.sat = true;
  const bool zlink = RARE && w == 0u && u >= v;
  if (RARE && zlink && !P.hc && zlink_unsafe(a.zcyc(), u, v)) { bd_all = min(bd_all, c); return; }
  const bool hz = RARE && zlink;
  const bool lt = hz ? (c < zb) : (c < r.bd), eq = !hz && c == r.bd;
  const uint32_t hh = pay >> P.mbits;
  uint32_t contrib = pay & mmask;
  if ((lt || eq) && hh == 0u && c < P.inf_t) {                               // parent: root or hops-0 network
    const uint32_t base_s = (u == my_root) ? 0u : slot_base_of(a.slot_tabs(), root_slot, u);
    const uint32_t sidx = base_s + fpos_of();
    contrib = ((v_router || a.net_nexthops) && sidx < P.mbits) ? (1u << sidx) : 0u;
  }
  if (hz) { if (lt) { zb = c; zm = contrib; zh = hh; } return; }             // see fused_row_any
  const uint32_t m_or = r.bm | contrib;
  r.bm = lt ? contrib : (eq ? m_or : r.bm);
  const bool newp = lt || (eq && d < r.bpd);
  r.bpd = newp ? d : r.bpd;
  r.bh = newp ? hh : r.bh;
  r.bd = min(r.bd, c);
}

// VPT = vertices per thread (1, 2, 4 or 8: the host picks the smallest that covers n with the block size it launches).
// LL = the link records are staged in LDS: a template parameter, not a run-time select, so that the loads are
// ds_read — a pointer that may be LDS or global compiles to FLAT loads, whose latency is that of a global load
// (measured: 3 700 cycles per sweep on ospf-500 with the select, profiles/r02_notes.md).
template <bool MAXINF, int VPT, bool LL>
__global__ __launch_bounds__(SINGLE_THREADS) void k_single(SingleArgs a) {
  constexpr uint32_t SINGLE_VPT = (uint32_t)VPT;
  extern __shared__ uint64_t s_st[];
  __shared__ int s_changed[4];
  const GraphDev &g = a.gp->g;
  const uint32_t n = g.n;
  const uint32_t root_slot = blockIdx.x;
  const uint32_t my_root = a.roots[root_slot];
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  const FusedParams P = a.P;
  const uint32_t mmask = (1u << P.mbits) - 1u;
  const size_t orow = a.o.row(root_slot) * (size_t)n;
  if (my_root == INF) {                            // padding root: empty SPT
    for (uint32_t v = tid; v < n; v += nthr) {
      a.o.dist[orow + v] = INF;
      if (a.o.hops) a.o.hops[orow + v] = 0;
      if (a.o.flags) a.o.flags[orow + v] = 0;
      if (a.o.mask) for (uint32_t k = 0; k < a.o.out_words; ++k) a.o.mask[(orow + v) * a.o.out_words + k] = 0;
    }
    if (tid == 0) a.lane_flags[root_slot] = 0u;
    return;
  }
  const uint64_t t_clk0 = clock64(), t_wall0 = wall_clock64();
  const uint32_t e_in = g.e_in;
  uint32_t *const s_src = (uint32_t *)(s_st + n);
  uint32_t *const s_w = s_src + e_in;
  uint32_t *const s_fp = s_w + e_in;
  const uint32_t *const g_src = g.in_src, *const g_w = g.in_w, *const g_fp = g.in_fpos;
  for (uint32_t v = tid; v < n; v += nthr) s_st[v] = (v == my_root) ? 0ull : ~0ull;
  if (LL) {
    for (uint32_t e = tid; e < e_in; e += nthr) { s_src[e] = g_src[e]; s_w[e] = g_w[e]; s_fp[e] = g_fp[e]; }
  }
  if (tid < 4) s_changed[tid] = 0;
  // this thread's vertices: row bounds, kind and "needs the general routine" in registers for the whole run
  uint32_t ce0[SINGLE_VPT], ce1[SINGLE_VPT];       // ce1 bit 31: network, bit 30: rare row
  uint64_t cur[SINGLE_VPT];
#pragma unroll
  for (uint32_t i = 0; i < SINGLE_VPT; ++i) {
    const uint32_t v = tid + i * nthr;
    ce0[i] = 0u; ce1[i] = 0u; cur[i] = ~0ull;
    if (v < n) {
      const uint32_t e0 = g.in_ptr[v], e1 = g.in_ptr[v + 1];
      bool rare = P.hc != 0u;
      for (uint32_t e = e0; e < e1; ++e) {
        const uint32_t sw = g.in_src[e];
        rare = rare || (!a.ignore_ovl && (sw & SRC_NO_TRANSIT)) || (g.in_w[e] == 0u && (sw & SRC_MASK) >= v);
      }
      ce0[i] = e0;
      ce1[i] = e1 | ((g.vflags[v] & 1u) ? 0x80000000u : 0u) | (rare ? 0x40000000u : 0u);
      cur[i] = (v == my_root) ? 0ull : ~0ull;
    }
  }
  // One vertex per thread (the reference's own 500-router case): a PLAIN vertex — at most SINGLE_RL in-links, none of
  // the rare conditions, no source that can have hops == 0 (the root, a network vertex) — keeps its link records in
  // REGISTERS for the whole run and takes the lean accumulation (acc_link): a sweep is then one LDS round trip (the
  // sources' words) plus ~15 vector instructions per link, instead of two round trips and the general per-link routine
  // (51 sweeps on ospf-500: 3 700 -> ~1 000 cycles each).
  uint32_t ls[SINGLE_RL], lw[SINGLE_RL], lb[SINGLE_RL];   // source, cost, slot bit of a link out of the ROOT (its word is 0:
  bool plain = false;                                      // OR-ing the bit into the pay bits is the whole "direct" case)
  if (SINGLE_VPT == 1) {
    const uint32_t v = tid;
#pragma unroll
    for (uint32_t k = 0; k < SINGLE_RL; ++k) { ls[k] = min(v, n - 1u); lw[k] = INF; lb[k] = 0u; }
    if (v < n && v != my_root) {
      const uint32_t e0 = ce0[0], e1 = ce1[0] & 0x3FFFFFFFu;
      const bool v_rt = !(ce1[0] >> 31);
      plain = !(ce1[0] & 0x40000000u) && e1 - e0 <= SINGLE_RL;
#pragma unroll
      for (uint32_t k = 0; k < SINGLE_RL; ++k)
        if (e0 + k < e1) {
          ls[k] = g.in_src[e0 + k] & SRC_MASK; lw[k] = g.in_w[e0 + k];
          plain = plain && !(g.vflags[ls[k]] & 1u);           // a network source may have hops == 0: general routine
          if (ls[k] == my_root) {
            const uint32_t sidx = g.in_fpos[e0 + k];          // the root's slot base is 0
            lb[k] = ((v_rt || a.net_nexthops) && sidx < P.mbits) ? (1u << sidx) : 0u;
          }
        }
    }
  }
  __syncthreads();
  bool sat = false, need_exact = false, dyn = false, ovf = false;
  const uint32_t max_sweeps = 4u * n + 64u;        // far beyond any run; a run that gets there is handed to k_exact
  uint32_t sweep = 0;
  const uint64_t t_loop0 = clock64();
  // Sweeps run FREE in groups of SINGLE_FREE: no barrier, no flag — reads of a neighbour's word race with its owner's
  // write anyway, a stale word is a word the neighbour held earlier, and values only decrease — then one CHECKED sweep
  // between two barriers: it starts after every write of the free sweeps is visible, and if no thread changes anything
  // in it, every thread has read the final state.  (A barrier + flag round trip per sweep was half of a sweep's time on
  // the 500-router case.)
  for (;; ++sweep) {
    const bool checked = sweep % (SINGLE_FREE + 1u) == SINGLE_FREE;
    const uint32_t round = sweep / (SINGLE_FREE + 1u);
    if (checked) __syncthreads();
    bool any = false;
    if (SINGLE_VPT == 1 && plain) {
      const uint32_t v = tid;
      uint64_t qs[SINGLE_RL];
#pragma unroll
      for (uint32_t k = 0; k < SINGLE_RL; ++k) qs[k] = s_st[ls[k]];
      // two passes as in row16_compute (independent per-link work, short dependency chains: two waves per SIMD cannot
      // hide a 50-deep chain): candidates and their minimum, then the tight links — the in-row order (cost descending,
      // source ascending) makes the FIRST tight link the first discoverer, so the walk goes backwards and overwrites
      uint32_t c[SINGLE_RL];
      bool sat1 = false;
#pragma unroll
      for (uint32_t k = 0; k < SINGLE_RL; ++k) {
        const uint32_t d = (uint32_t)(qs[k] >> 32);
        c[k] = add_sat(d, lw[k]);                                // padding links cost all ones: never reached
        if (MAXINF && c[k] == INF && d != INF && lw[k] != INF) sat1 = true;
      }
      const uint32_t bd = min(min(min(c[0], c[1]), min(c[2], c[3])), min(min(c[4], c[5]), min(c[6], c[7])));
      static_assert(SINGLE_RL == 8, "min tree");
      uint32_t macc = 0u, bpay = 0u;
#pragma unroll
      for (int k = (int)SINGLE_RL - 1; k >= 0; --k) {
        const bool t = c[k] == bd;
        const uint32_t pb = (uint32_t)qs[k] | lb[k];
        macc |= t ? pb : 0u;
        bpay = t ? pb : bpay;
      }
      RowAcc r{bd, macc & mmask, 0u, bpay >> P.mbits, sat1};
      const RowOut<uint64_t> o = finish_row<uint64_t>(r, v, my_root, (ce1[0] >> 31) ^ 1u, INF, P);
      sat = sat || o.sat; ovf = ovf || o.ovf;
      if (o.nw != cur[0]) { cur[0] = o.nw; s_st[v] = o.nw; any = true; }
    } else
#pragma unroll
    for (uint32_t i = 0; i < SINGLE_VPT; ++i) {
      const uint32_t v = tid + i * nthr;
      if (v >= n || v == my_root) continue;
      const uint32_t e0 = ce0[i], e1 = ce1[i] & 0x3FFFFFFFu;
      const uint32_t v_router = (ce1[i] >> 31) ^ 1u;
      const bool rare = (ce1[i] & 0x40000000u) != 0u;
      RowAcc r{INF, 0u, INF, 0u, false};
      uint32_t bd_all = INF, zb = INF, zm = 0u, zh = 0u;
      constexpr uint32_t PF = 4;                    // link records, then their sources' states, of 4 links in flight
      for (uint32_t eb = e0; eb < e1; eb += PF) {
        uint32_t rs[PF], rw[PF], rf[PF];
        uint64_t qs[PF];
#pragma unroll
        for (uint32_t k = 0; k < PF; ++k) {
          const uint32_t e = min(eb + k, e1 - 1u);
          if (LL) { rs[k] = s_src[e]; rw[k] = s_w[e]; rf[k] = s_fp[e]; }
          else    { rs[k] = g_src[e]; rw[k] = g_w[e]; rf[k] = g_fp[e]; }
        }
#pragma unroll
        for (uint32_t k = 0; k < PF; ++k) qs[k] = s_st[rs[k] & SRC_MASK];
#pragma unroll
        for (uint32_t k = 0; k < PF; ++k) {
          if (eb + k >= e1) break;
          const uint32_t fk = rf[k];
          if (rare) single_link<MAXINF, true>(r, bd_all, zb, zm, zh, rs[k], rw[k], [fk]() { return fk; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
          else      single_link<MAXINF, false>(r, bd_all, zb, zm, zh, rs[k], rw[k], [fk]() { return fk; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
        }
      }
      const RowOut<uint64_t> o = finish_row_z<uint64_t>(r, v, my_root, v_router, bd_all, rare ? zb : INF, zm, zh, P.hc != 0u, P);
      sat = sat || o.sat; dyn = dyn || o.dyn; ovf = ovf || o.ovf;
      if (o.nw != cur[i]) { cur[i] = o.nw; s_st[v] = o.nw; any = true; }
    }
    if (!checked) continue;
    // flag slot round & 3 is set during the checked sweep of `round`, read after its barrier, and cleared by thread 0
    // during round + 2 — after every thread has passed the barriers of round + 1 and therefore finished reading it, and
    // long before round + 4 sets it again.
    if (any) s_changed[round & 3u] = 1;
    if (tid == 0) s_changed[(round + 2u) & 3u] = 0;
    __syncthreads();
    if (s_changed[round & 3u] == 0) break;
    if (sweep >= max_sweeps) { need_exact = true; break; }
  }
  // results: one row of the row-major outputs per root, consecutive threads = consecutive vertices
  for (uint32_t v = tid; v < n; v += nthr) {
    const uint64_t x = s_st[v];
    const bool in = x != ~0ull;
    const uint32_t pay = (uint32_t)x;
    a.o.dist[orow + v] = in ? (uint32_t)(x >> 32) : INF;
    if (a.o.hops) a.o.hops[orow + v] = in ? (uint16_t)(pay >> P.mbits) : (uint16_t)0;
    if (a.o.flags) a.o.flags[orow + v] = in ? 1 : 0;
    if (a.o.mask) {
      a.o.mask[(orow + v) * a.o.out_words] = in ? (uint64_t)(pay & mmask) : 0ull;
      for (uint32_t k = 1; k < a.o.out_words; ++k) a.o.mask[(orow + v) * a.o.out_words + k] = 0;
    }
  }
  uint32_t lf = 0;
  if ((MAXINF && sat) || need_exact) lf |= LF_NEED_EXACT;
  if (dyn) lf |= LF_DYN;
  if (ovf) lf |= LF_OVERFLOW;
  single_store_flags(a.lane_flags, root_slot, lf, &s_changed[0]);
  if (a.count_rows && tid == 0) {                  // HSPF_RUN_COUNT_ROWS: rows evaluated; workgroup 0 also leaves its sweep
    atomicAdd(&a.gp->rows_done[root_slot & 127u], (sweep + 1u) * n);   // count, shader cycles and 100 MHz wall ticks
    if (root_slot == 0) {
      a.gp->rows_done[128] = sweep + 1u;
      a.gp->rows_done[129] = (uint32_t)(clock64() - t_clk0);
      a.gp->rows_done[130] = (uint32_t)(wall_clock64() - t_wall0);
      a.gp->rows_done[131] = (uint32_t)(t_loop0 - t_clk0);          // cycles before the first sweep (staging, set-up)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_lv — a few roots on a graph too large for k_single: lane = VERTEX.
//
// run_area / compute_spt are called with ONE root per area / level (holo-ospf/src/spf.rs:540-542,
// holo-isis/src/spf.rs:746-761).  On the lane = root engine such a run keeps one lane of every 64 busy: a sweep walks
// 256-byte rows for 4 useful bytes (isis-100k, one root: 25 launches x 21 us).  Here the state of a root is one row-major
// array of packed 8-byte words [dist32 | hops | mask] (the k_single / wide k_fused word, <= 24 first-hop slots) in
// HBM — 0.8 MB at 100 k vertices, i.e. L2 resident — thread t of block b owns vertex 256 b + t of root blockIdx.y, and
// a sweep is one launch over the vertices whose activation stamp says that an in-neighbour changed (the push-stamp
// protocol of k_fused).  Same fixed point, same row routine (single_link / finish_row), same exactness flags; a
// neighbour's word is ONE 8-byte access, so a (distance, hops, mask) triple is never torn and a stale one is a triple
// the neighbour held earlier in the run ("memory is monotone", as in k_fused).  Neighbour states are scattered 8-byte
// gathers, which is why this layout loses against lane = root beyond a handful of roots (run_impl picks by root count).
struct LvArgs {
  // the first 16 dwords are preloaded into SGPRs at wave launch (holo_amd/build.py): what the early exit and the first
  // round trip of a thread need
  int *changed;
  int sweep;
  uint32_t n;
  uint32_t *act;               // [n_roots][n] activation stamps
  uint64_t *st;                // [n_roots][n]
  const uint32_t *roots;
  const uint32_t *in_ptr;
  const uint8_t *rowflags;
  const uint8_t *vflags;
  // the rest comes with the ordinary argument load
  GraphDev g;
  SlotTabs tabs;
  uint32_t *rows_done;
  FusedParams P;               // the 8-byte state's parameters (sh = 0)
  uint32_t net_nexthops, ignore_ovl, n_roots, count_rows;
  uint32_t *lane_flags;
  const uint32_t *ell_so, *ell_w, *ell_od;   // the fixed-stride copy of the rows with at most 16 in-links / out-links (kb_ell)
  uint32_t ell_mode;                  // 2: records and wake-up offsets from it, 1: records only, 0: not used (HSPF_VARIANT bits 25 / 26: A/B)
  __device__ __forceinline__ const SlotTabs &slot_tabs() const { return tabs; }
  __device__ __forceinline__ const uint8_t *zcyc() const { return g.zcyc; }
};

// A launch is a chain of dependent round trips, not a stream of bytes (one root on isis-100k: ~43 k due vertices per
// sweep): everything that depends only on the thread's vertex comes in ONE trip (stamp, row bounds, flags, old word),
// then the records of up to LV_PF links in one, then their sources' words in one.  The position of a link in its
// source row is only fetched for links out of a hops-0 parent (the root's neighbours).
constexpr uint32_t LV_PF = 16;

// Round 5: the link records of the rows with at most 16 in-links come from the fixed-stride (ELL) copy the lean sweep
// uses, fetched by the WAVE — 64 rows x 64 bytes per array as four 16-byte-per-lane loads, handed to their threads through
// LDS — instead of 32 loads per thread that each touch 64 different cache lines (a thread's records are consecutive, a
// wave's are 64 rows apart): a mid-run sweep on isis-100k made ~2 M cache-line requests, 1.4 M of them for records, and
// took 14 us for it; the gathers of the sources' words stay as they are.  The row's out-neighbours (their stamp offsets,
// `ell_od`) come the same way: a changed vertex wakes them up without the two dependent round trips through out_ptr / out_dst.  Rows that need the rare-case routine (an
// overloaded source, a zero-cost link from a higher-numbered source, hop-count graphs) or have more than 16 in-links keep
// the walk over in_src / in_w.
constexpr uint32_t LV_ROW_WORDS = 20;             // LDS stride of a staged 16-word record row: 80 bytes (16-byte aligned, off the bank period)

template <bool MAXINF>
__global__ __launch_bounds__(256) void k_lv(LvArgs a) {
  if (a.sweep > 0 && a.changed[a.sweep - 1] == 0) return;
  __shared__ uint32_t s_rec[4][3][64 * LV_ROW_WORDS];
  const GraphDev &g = a.g;
  const uint32_t n = a.n;
  const uint32_t root_slot = blockIdx.y;
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t cur = (uint32_t)a.sweep + 2u;
  uint32_t *A = a.act + (size_t)root_slot * n;
  uint64_t *S = a.st + (size_t)root_slot * n;
  const uint32_t my_root = a.roots[root_slot];
  const uint32_t vc = min(v, n - 1u);              // (lanes past the last vertex ride along for the wave's loads)
  const uint32_t av = A[vc];
  const uint32_t e0 = a.in_ptr[vc], e1 = a.in_ptr[vc + 1];
  const uint32_t rf = a.rowflags[vc], vf = a.vflags[vc];
  const uint64_t old = S[vc];
  const bool due = v < n && my_root != INF && av >= cur && v != my_root;   // else: nothing changed around this vertex
  if (!__any(due ? 1 : 0)) return;
  const FusedParams P = a.P;
  const uint32_t mmask = (1u << P.mbits) - 1u;
  const bool rare = P.hc != 0u || (!a.ignore_ovl && (rf & RF_NT)) || (rf & RF_ZERO);
  const bool ell = due && !rare && !(rf & RF_MANY) && a.ell_mode != 0u;
  if (__any(ell ? 1 : 0)) {                        // the wave's 64 record rows, 1 KB per load instruction
    const uint32_t v_w = blockIdx.x * 256u + wave * 64u;
    const uint4 *eso = (const uint4 *)a.ell_so, *ew = (const uint4 *)a.ell_w, *eod = (const uint4 *)a.ell_od;
    uint4 qa[4], qb[4], qc[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint32_t row = min(v_w + (lane >> 2) + 16u * i, n);               // row n of the copy is all pad
      qa[i] = eso[(size_t)row * 4u + (lane & 3u)];
      qb[i] = ew[(size_t)row * 4u + (lane & 3u)];
      qc[i] = eod[(size_t)row * 4u + (lane & 3u)];                             // the out-neighbours' stamp offsets: the wake-up below
    }
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint32_t o = ((lane >> 2) + 16u * i) * LV_ROW_WORDS + (lane & 3u) * 4u;
      *(uint4 *)&s_rec[wave][0][o] = qa[i];
      *(uint4 *)&s_rec[wave][1][o] = qb[i];
      *(uint4 *)&s_rec[wave][2][o] = qc[i];
    }
  }
  if (!due) return;
  const uint32_t v_router = (vf & 1u) ? 0u : 1u;
  RowAcc r{INF, 0u, INF, 0u, false};
  uint32_t bd_all = INF, zb = INF, zm = 0u, zh = 0u;
  bool od_row = false;                             // the row's out-links are all in its staged record (at most 16 of them)
  if (ell) {
    uint32_t rs[LV_PF], rw[LV_PF];
    uint64_t qs[LV_PF];
    static_assert(LV_PF == 16, "one ELL row");
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint4 x = *(const uint4 *)&s_rec[wave][0][lane * LV_ROW_WORDS + 4u * k];
      const uint4 y = *(const uint4 *)&s_rec[wave][1][lane * LV_ROW_WORDS + 4u * k];
      rs[4 * k] = x.x; rs[4 * k + 1] = x.y; rs[4 * k + 2] = x.z; rs[4 * k + 3] = x.w;
      rw[4 * k] = y.x; rw[4 * k + 1] = y.y; rw[4 * k + 2] = y.z; rw[4 * k + 3] = y.w;
    }
    const uint32_t deg = e1 - e0;                                             // (<= 16: not RF_MANY)
    od_row = !(rs[0] & 0x20u) && a.ell_mode == 2u;                                                // entry 0's low byte: in-degree | (more than 16 out-links) << 5 | network << 7
#pragma unroll
    for (uint32_t k = 0; k < LV_PF; ++k) { rs[k] = k < deg ? rs[k] >> 8 : v; qs[k] = S[rs[k]]; }   // entry = source << 8 (| row info in entry 0)
#pragma unroll
    for (uint32_t k = 0; k < LV_PF; ++k) {
      if (k < deg) {
        const uint32_t *fp = g.in_fpos + (e0 + k);
        single_link<MAXINF, false>(r, bd_all, zb, zm, zh, rs[k], rw[k], [fp]() { return *fp; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
      }
    }
  } else
  for (uint32_t eb = e0; eb < e1; eb += LV_PF) {
    uint32_t rs[LV_PF], rw[LV_PF];
    uint64_t qs[LV_PF];
#pragma unroll
    for (uint32_t k = 0; k < LV_PF; ++k) {
      const uint32_t e = min(eb + k, e1 - 1u);
      rs[k] = g.in_src[e]; rw[k] = g.in_w[e];
    }
#pragma unroll
    for (uint32_t k = 0; k < LV_PF; ++k) qs[k] = S[rs[k] & SRC_MASK];
#pragma unroll
    for (uint32_t k = 0; k < LV_PF; ++k) {
      if (eb + k >= e1) break;
      const uint32_t *fp = g.in_fpos + (eb + k);
      if (rare) single_link<MAXINF, true>(r, bd_all, zb, zm, zh, rs[k], rw[k], [fp]() { return *fp; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
      else      single_link<MAXINF, false>(r, bd_all, zb, zm, zh, rs[k], rw[k], [fp]() { return *fp; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
    }
  }
  const RowOut<uint64_t> o = finish_row_z<uint64_t>(r, v, my_root, v_router, bd_all, rare ? zb : INF, zm, zh, P.hc != 0u, P);
  if (o.nw != old) {
    S[v] = o.nw;
    a.changed[a.sweep] = 1;
    if (od_row) {                                  // wake the out-neighbours up: their stamp offsets came with the record row
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint4 x = *(const uint4 *)&s_rec[wave][2][lane * LV_ROW_WORDS + 4u * k];
        if (x.x != 0xFFFFFFFFu) A[x.x >> 2] = cur + 1u;
        if (x.y != 0xFFFFFFFFu) A[x.y >> 2] = cur + 1u;
        if (x.z != 0xFFFFFFFFu) A[x.z >> 2] = cur + 1u;
        if (x.w != 0xFFFFFFFFu) A[x.w >> 2] = cur + 1u;
      }
    } else
    for (uint32_t k = g.out_ptr[v], k1 = g.out_ptr[v + 1]; k < k1; ++k) A[g.out_dst[k]] = cur + 1u;
  }
  if (a.count_rows) atomicAdd(&a.rows_done[(blockIdx.x + threadIdx.x) & 127u], 1u);
  uint32_t lf = 0;
  if (MAXINF && o.sat) lf |= LF_NEED_EXACT;
  if (o.dyn) lf |= LF_DYN;
  if (o.ovf) lf |= LF_OVERFLOW;
  if (lf) atomicOr(&a.lane_flags[root_slot], lf);
}

// init for k_lv (state = all ones, stamps = 0 by memset): the root's word = (0, 0, 0), its out-neighbours due in sweep 0.
__global__ void k_init_lv(GraphDev g, uint64_t *st, uint32_t *act, const uint32_t *roots, uint32_t n_roots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_roots) return;
  const uint32_t r = roots[i];
  if (r == INF) return;
  st[(size_t)i * g.n + r] = 0ull;
  for (uint32_t k = g.out_ptr[r]; k < g.out_ptr[r + 1]; ++k) act[(size_t)i * g.n + g.out_dst[k]] = 2u;
}

// results of k_lv: the state is row-major already, one thread per (root, vertex)
__global__ __launch_bounds__(256) void k_emit_lv(uint32_t n, const uint64_t *__restrict__ st, FusedParams P, OutDev o) {
  const uint32_t v = blockIdx.x * 256u + threadIdx.x, r = blockIdx.y;
  if (v >= n) return;
  const uint64_t x = st[(size_t)r * n + v];
  const size_t idx = o.row(r) * n + v;
  const bool in = x != ~0ull;
  const uint32_t pay = (uint32_t)x;
  o.dist[idx] = in ? (uint32_t)(x >> 32) : INF;
  if (o.hops) o.hops[idx] = in ? (uint16_t)(pay >> P.mbits) : (uint16_t)0;
  if (o.flags) o.flags[idx] = in ? 1 : 0;
  if (o.mask) {
    o.mask[idx * o.out_words] = in ? (uint64_t)(pay & ((1u << P.mbits) - 1u)) : 0ull;
    for (uint32_t k = 1; k < o.out_words; ++k) o.mask[idx * o.out_words + k] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// k_xcd — one root (up to eight) on a MID-SIZE graph: one XCD per root, the whole state replicated in every CU's LDS.
//
// The reference's real call pattern is one root per area / level (holo-ospf/src/spf.rs:540-542, holo-isis/src/spf.rs:
// 746-761).  On a graph too large for the one-workgroup kernel (k_single: 1-2 k vertices) such a run was a chain of ~40
// launches of ~5 us each (ospf-10k: 0.20 ms), whatever kernel ran inside them.  Here the chain stays inside ONE launch:
//   * consecutive workgroups of a grid go to consecutive XCDs, so the workgroups b with equal b % 8 share one (observed
//     placement, MI355X_MICROARCH.md — which XCD that is depends on where the previous dispatch stopped; never relied on
//     for correctness: see "bounded" below; every workgroup reports its HW_REG_XCC_ID and the host notes a root whose
//     workgroups did not share one); root r of the run takes the class (xcd0 + r) % 8, one workgroup per CU, each
//     owning a contiguous range of `vw` vertices, one vertex per thread (and n / 16 threads at least: the re-read below
//     is one batch of loads per thread then);
//   * every workgroup keeps ALL n words of its root ([dist32 | hops | mask], the k_single / k_lv word) in LDS: a
//     neighbour's word is an LDS read, never a round trip to the L2.  A vertex of at most XCD_RL links keeps its link
//     records in registers for the whole run — a plain one (k_single's definition) for the two-pass routine, the others
//     raw for the general row routine (single_link / finish_row: same fixed point, same exactness flags as every other
//     kernel); longer rows walk their records in global memory;
//   * a sweep = evaluate the own vertices from the replica, store what changed to the replica AND to the root's array in
//     global memory (plain stores: the lines stay in this XCD's L2), then a barrier among the workgroups of the XCD made
//     of plain flag stores and sc1 polling loads (L1 bypassed, L2 served: 0.55 us for up to 64 workgroups,
//     tools/ubench/xcd_barrier.hip — a chip-wide barrier is 4-10 us because the eight L2s are not coherent).  The flag
//     carries "something in my range changed"; after the barrier a workgroup re-reads (16-byte sc1 loads) exactly the
//     ranges that changed.  No range changed = fixed point.  Jacobi across workgroups, in place inside one: 37 sweeps on
//     ospf-10k where the sweep engine's dispatch-order Gauss-Seidel needs ~38 launches;
//   * flags are numbered (run epoch, sweep) and double-buffered by sweep parity — a workgroup can be at most one barrier
//     ahead of another — so nothing is cleared between runs;
//   * BOUNDED: every wait gives up after `timeout_ticks` (a workgroup that is not resident, or not on the expected XCD,
//     whose stores the pollers then may never see): the workgroup reports XCD_ST_ABORT in its status word (pinned host
//     memory, one plain store per workgroup at the end) and the host redoes the run on the launch-per-sweep path.
// Results go straight to the row-major output arrays (no emit launch), as in k_single.
constexpr uint32_t XCD_MAX_WG = 32;            // workgroups per root: the CUs of one XCD
constexpr uint32_t XCD_MAX_N = 20000;          // 160 000 bytes of LDS per workgroup
constexpr uint32_t XCD_MAX_ROOTS = 8;
constexpr uint32_t XCD_RL = 12;                // link records a plain vertex keeps in registers
constexpr uint32_t XCD_CTL_WORDS = 2 * XCD_MAX_WG;
constexpr uint32_t XCD_MAX_SWEEPS = 4000;      // (12 bits of a flag's counter)
constexpr uint32_t XCD_ST_DONE = 0x80000000u, XCD_ST_ABORT = 0x40000000u;   // | HW_REG_XCC_ID << 24 | sweeps << 8 | LF_* bits

struct XcdArgs {
  const FusedGraph *gp;
  const uint32_t *roots;
  FusedParams P;                 // the 8-byte state's parameters (sh = 0)
  uint32_t net_nexthops, ignore_ovl, n_roots, epoch;
  uint32_t n_wg, vw, xcd0, n_pad;
  uint32_t vw_inv;               // ceil(2^32 / vw): i / vw = __umulhi(i, vw_inv) for every vertex index
  uint64_t *st;                  // [n_roots][n_pad]: every vertex' word; the owner's plain stores, everybody's sc1 loads
  uint32_t *ctl;                 // [n_roots][XCD_CTL_WORDS] barrier flags: (epoch << 12 | sweep + 1) << 1 | changed
  uint32_t *status;              // [n_roots][XCD_MAX_WG], pinned host memory
  uint64_t timeout_ticks;        // 100 MHz ticks
  uint32_t skew;                 // tests (HSPF_XCD_SKEW): a root's workgroups are CONSECUTIVE blocks, i.e. sit on different XCDs
  OutDev o;
  __device__ __forceinline__ const SlotTabs &slot_tabs() const { return gp->tabs; }
  __device__ __forceinline__ const uint8_t *zcyc() const { return gp->g.zcyc; }
};

template <bool MAXINF, bool PROF>
__global__ __launch_bounds__(640) void k_xcd(XcdArgs a) {     // (XCD_MAX_N / XCD_MAX_WG vertices, one per thread)
  extern __shared__ uint64_t s_st[];
  __shared__ uint32_t s_wany[16], s_mask, s_go, s_lf;
  const uint32_t p = a.skew ? blockIdx.x % a.n_wg : blockIdx.x >> 3;
  const uint32_t root_slot = a.skew ? blockIdx.x / a.n_wg : ((blockIdx.x & 7u) + 8u - a.xcd0) & 7u;
  if (root_slot >= a.n_roots) return;
  const GraphDev &g = a.gp->g;
  const uint32_t n = g.n, tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t my_root = a.roots[root_slot];
  const FusedParams P = a.P;
  const uint32_t mmask = (1u << P.mbits) - 1u;
  const size_t orow = a.o.row(root_slot) * (size_t)n;
  const uint32_t v0 = p * a.vw, v1 = min(n, v0 + a.vw);
  const uint32_t v = v0 + tid;
  const bool mine = v < v1;
  uint32_t *const status = a.status + root_slot * XCD_MAX_WG + p;
  if (my_root == INF) {                            // padding root: empty SPT
    if (mine) {
      a.o.dist[orow + v] = INF;
      if (a.o.hops) a.o.hops[orow + v] = 0;
      if (a.o.flags) a.o.flags[orow + v] = 0;
      if (a.o.mask) for (uint32_t k = 0; k < a.o.out_words; ++k) a.o.mask[(orow + v) * a.o.out_words + k] = 0;
    }
    if (tid == 0) *status = XCD_ST_DONE;
    return;
  }
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;        // HW_REG_XCC_ID
  uint64_t *const S = a.st + (size_t)root_slot * a.n_pad;
  const __amdgpu_buffer_rsrc_t rS = st_rsrc(S, a.n_pad * 8u);
  uint32_t *const F = a.ctl + root_slot * XCD_CTL_WORDS;
  for (uint32_t i = tid; i < n; i += nthr) s_st[i] = (i == my_root) ? 0ull : ~0ull;
  uint64_t cur = (v == my_root) ? 0ull : ~0ull;
  if (mine) S[v] = cur;
  if (tid == 0) s_lf = 0u;
  // this thread's vertex: row bounds, kind, "needs the general routine"; a plain vertex' link records in registers
  uint32_t e0 = 0u, e1 = 0u;
  bool rare = P.hc != 0u, plain = false, regs = false;
  uint32_t v_router = 1u;
  uint32_t ls[XCD_RL], lw[XCD_RL], lb[XCD_RL];
#pragma unroll
  for (uint32_t k = 0; k < XCD_RL; ++k) { ls[k] = min(v, n - 1u); lw[k] = INF; lb[k] = 0u; }
  if (mine && v != my_root) {
    e0 = g.in_ptr[v]; e1 = g.in_ptr[v + 1];
    v_router = (g.vflags[v] & 1u) ? 0u : 1u;
    for (uint32_t e = e0; e < e1; ++e) {
      const uint32_t sw = g.in_src[e];
      rare = rare || (!a.ignore_ovl && (sw & SRC_NO_TRANSIT)) || (g.in_w[e] == 0u && (sw & SRC_MASK) >= v);
    }
    plain = !rare && e1 - e0 <= XCD_RL;
#pragma unroll
    for (uint32_t k = 0; k < XCD_RL; ++k)
      if (e0 + k < e1) {
        ls[k] = g.in_src[e0 + k] & SRC_MASK; lw[k] = g.in_w[e0 + k];
        plain = plain && !(g.vflags[ls[k]] & 1u);               // a network source may have hops == 0: general routine
        if (ls[k] == my_root) {
          const uint32_t sidx = g.in_fpos[e0 + k];               // the root's slot base is 0
          lb[k] = ((v_router || a.net_nexthops) && sidx < P.mbits) ? (1u << sidx) : 0u;
        }
      }
    // a vertex that needs the general routine but has at most XCD_RL links keeps its records in the same registers, raw:
    // source | SRC_NO_TRANSIT, cost, position of the link in its source row (the sweeps then never leave the CU either)
    regs = !plain && e1 - e0 <= XCD_RL;
    if (regs) {
#pragma unroll
      for (uint32_t k = 0; k < XCD_RL; ++k)
        if (e0 + k < e1) { ls[k] = g.in_src[e0 + k]; lw[k] = g.in_w[e0 + k]; lb[k] = g.in_fpos[e0 + k]; }
        else { ls[k] = min(v, n - 1u); lw[k] = INF; lb[k] = 0u; }
    }
  }
  bool sat = false, need_exact = false, dyn = false, ovf = false, aborted = false;
  const uint32_t max_sweeps = min(4u * n + 64u, XCD_MAX_SWEEPS);
  uint32_t sweep = 0;
  // PROF (HSPF_XCD_PROF, tuning only): 100 MHz ticks workgroup 0 of the first root spends evaluating / publishing / waiting
  // at the barrier / re-reading ranges, left in the last four status words
  uint64_t t_ph[4] = {0, 0, 0, 0}, t_mark = PROF ? wall_clock64() : 0;
  __syncthreads();
  for (;; ++sweep) {
    bool any = false;
    if (mine && v != my_root) {
      RowOut<uint64_t> o;
      if (plain) {
        uint64_t qs[XCD_RL];
#pragma unroll
        for (uint32_t k = 0; k < XCD_RL; ++k) qs[k] = s_st[ls[k]];
        uint32_t c[XCD_RL];
        bool sat1 = false;
        uint32_t bd = INF;
#pragma unroll
        for (uint32_t k = 0; k < XCD_RL; ++k) {
          const uint32_t d = (uint32_t)(qs[k] >> 32);
          c[k] = add_sat(d, lw[k]);                              // padding links cost all ones: never reached
          if (MAXINF && c[k] == INF && d != INF && lw[k] != INF) sat1 = true;
          bd = min(bd, c[k]);
        }
        uint32_t macc = 0u, bpay = 0u;                           // tight links walked backwards: the first one is the first discoverer
#pragma unroll
        for (int k = (int)XCD_RL - 1; k >= 0; --k) {
          const bool t = c[k] == bd;
          const uint32_t pb = (uint32_t)qs[k] | lb[k];
          macc |= t ? pb : 0u;
          bpay = t ? pb : bpay;
        }
        RowAcc r{bd, macc & mmask, 0u, bpay >> P.mbits, sat1};
        o = finish_row<uint64_t>(r, v, my_root, v_router, INF, P);
      } else if (regs) {
        uint64_t qs[XCD_RL];
#pragma unroll
        for (uint32_t k = 0; k < XCD_RL; ++k) qs[k] = s_st[ls[k] & SRC_MASK];
        RowAcc r{INF, 0u, INF, 0u, false};
        uint32_t bd_all = INF, zb = INF, zm = 0u, zh = 0u;
        const uint32_t cnt = e1 - e0;
#pragma unroll
        for (uint32_t k = 0; k < XCD_RL; ++k) {
          if (k < cnt) {                                           // (no break: the loop has to unroll for the records to stay in registers)
            const uint32_t fk = lb[k];
            if (rare) single_link<MAXINF, true>(r, bd_all, zb, zm, zh, ls[k], lw[k], [fk]() { return fk; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
            else      single_link<MAXINF, false>(r, bd_all, zb, zm, zh, ls[k], lw[k], [fk]() { return fk; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
          }
        }
        o = finish_row_z<uint64_t>(r, v, my_root, v_router, bd_all, rare ? zb : INF, zm, zh, P.hc != 0u, P);
      } else {
        RowAcc r{INF, 0u, INF, 0u, false};
        uint32_t bd_all = INF, zb = INF, zm = 0u, zh = 0u;
        constexpr uint32_t PF = 4;
        for (uint32_t eb = e0; eb < e1; eb += PF) {
          uint32_t rs[PF], rw[PF], rf[PF];
          uint64_t qs[PF];
#pragma unroll
          for (uint32_t k = 0; k < PF; ++k) {
            const uint32_t e = min(eb + k, e1 - 1u);
            rs[k] = g.in_src[e]; rw[k] = g.in_w[e]; rf[k] = g.in_fpos[e];
          }
#pragma unroll
          for (uint32_t k = 0; k < PF; ++k) qs[k] = s_st[rs[k] & SRC_MASK];
#pragma unroll
          for (uint32_t k = 0; k < PF; ++k) {
            if (eb + k >= e1) break;
            const uint32_t fk = rf[k];
            if (rare) single_link<MAXINF, true>(r, bd_all, zb, zm, zh, rs[k], rw[k], [fk]() { return fk; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
            else      single_link<MAXINF, false>(r, bd_all, zb, zm, zh, rs[k], rw[k], [fk]() { return fk; }, qs[k], v, v_router, my_root, root_slot, a, P, mmask);
          }
        }
        o = finish_row_z<uint64_t>(r, v, my_root, v_router, bd_all, rare ? zb : INF, zm, zh, P.hc != 0u, P);
      }
      sat = sat || o.sat; dyn = dyn || o.dyn; ovf = ovf || o.ovf;
      if (o.nw != cur) { cur = o.nw; s_st[v] = o.nw; S[v] = o.nw; any = true; }
    }
    // ---- barrier among the workgroups of this root's XCD, carrying "my range changed"
    if (PROF && tid == 0) { const uint64_t t = wall_clock64(); t_ph[0] += t - t_mark; t_mark = t; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's stores have reached the L2
    const bool wave_any = __any(any ? 1 : 0) != 0;
    if ((tid & 63u) == 0u) s_wany[tid >> 6] = wave_any ? 1u : 0u;
    __syncthreads();
    if (tid < 64u) {
      const uint32_t nw = (nthr + 63u) >> 6;
      const bool wg_any = __any((tid < nw && s_wany[tid]) ? 1 : 0) != 0;
      const uint32_t want = (a.epoch << 12) | (sweep + 1u);
      uint32_t *const Fg = F + (sweep & 1u) * XCD_MAX_WG;
      if (tid == 0) {
        Fg[p] = (want << 1) | (wg_any ? 1u : 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const uint64_t t0 = wall_clock64();
      if (PROF && tid == 0) { t_ph[1] += t0 - t_mark; t_mark = t0; }
      uint32_t f = want << 1;
      bool ok = true, gave_up = a.timeout_ticks == 0ull;           // (0: the test of the fallback — every workgroup gives up at its first barrier)
      while (!gave_up) {
        if (tid < a.n_wg) f = __hip_atomic_load(Fg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (f >> 1) >= want;
        if (__all(ok ? 1 : 0)) break;
        if (wall_clock64() - t0 > a.timeout_ticks) { gave_up = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      const uint64_t chg = __ballot((ok && tid < a.n_wg && (f & 1u)) ? 1 : 0);
      if (tid == 0) { s_mask = (uint32_t)chg; s_go = gave_up ? 2u : 1u; }
    }
    __syncthreads();
    if (PROF && tid == 0) { const uint64_t t = wall_clock64(); t_ph[2] += t - t_mark; t_mark = t; }
    if (s_go == 2u) { aborted = true; break; }
    const uint32_t cm = s_mask;
    if (cm == 0u) break;                                          // nobody changed anything: every replica holds the fixed point
    if (sweep + 1u >= max_sweeps) { need_exact = true; break; }   // far beyond any run: handed to k_exact
    // ---- the ranges that changed, from the L2 into the replica (the own range is up to date): ONE pass over the whole
    // array, eight 16-byte loads per thread in flight.  (Range by range the loop was a chain of dependent L2 round trips:
    // 7.5 us per sweep on ospf-10k instead of 2.2.  Twelve loads in flight, and only the 64-vertex chunks this workgroup's
    // vertices have in-links from — a third of them on ospf-10k's grid — measured SLOWER, 2.6 us: the loop is bound by its
    // own instructions, not by the 80 KB it moves; profiles/r05_notes.md, r05h-r05k.)
    const uint32_t want_ranges = cm & ~(1u << p);
    constexpr uint32_t XU = 8;
    for (uint32_t base = 2u * tid; base < n; base += 2u * nthr * XU) {
      uint4 q[XU];
      bool take[XU];
#pragma unroll
      for (uint32_t u = 0; u < XU; ++u) {
        const uint32_t i = base + 2u * nthr * u;                   // vw is even and the array 16-byte aligned: a pair never straddles ranges
        take[u] = i < n && ((want_ranges >> __umulhi(i, a.vw_inv)) & 1u);
        if (take[u]) q[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rS, i * 8u, 0, 16));   // aux 16 = sc1
      }
#pragma unroll
      for (uint32_t u = 0; u < XU; ++u) {
        const uint32_t i = base + 2u * nthr * u;
        if (take[u]) {
          s_st[i] = ((uint64_t)q[u].y << 32) | q[u].x;
          if (i + 1u < n) s_st[i + 1u] = ((uint64_t)q[u].w << 32) | q[u].z;
        }
      }
    }
    if (PROF && tid == 0) { const uint64_t t = wall_clock64(); t_ph[3] += t - t_mark; t_mark = t; }
    __syncthreads();
  }
  // ---- results of the own range, row-major
  if (mine) {
    const bool in = cur != ~0ull;
    const uint32_t pay = (uint32_t)cur;
    a.o.dist[orow + v] = in ? (uint32_t)(cur >> 32) : INF;
    if (a.o.hops) a.o.hops[orow + v] = in ? (uint16_t)(pay >> P.mbits) : (uint16_t)0;
    if (a.o.flags) a.o.flags[orow + v] = in ? 1 : 0;
    if (a.o.mask) {
      a.o.mask[(orow + v) * a.o.out_words] = in ? (uint64_t)(pay & mmask) : 0ull;
      for (uint32_t k = 1; k < a.o.out_words; ++k) a.o.mask[(orow + v) * a.o.out_words + k] = 0;
    }
  }
  uint32_t lf = 0;
  if ((MAXINF && sat) || need_exact) lf |= LF_NEED_EXACT;
  if (dyn) lf |= LF_DYN;
  if (ovf) lf |= LF_OVERFLOW;
  if (lf) atomicOr(&s_lf, lf);
  __syncthreads();
  if (tid == 0)
    *status = XCD_ST_DONE | (aborted ? XCD_ST_ABORT : 0u) | (xcc << 24) | (min(sweep + 1u, 0xFFFFu) << 8) | s_lf;
  if (PROF && tid == 0 && p == 0 && root_slot == 0 && a.n_roots < XCD_MAX_ROOTS)
    for (int k = 0; k < 4; ++k) a.status[(XCD_MAX_ROOTS - 1) * XCD_MAX_WG + 28 + k] = (uint32_t)t_ph[k];
}

// k_single_lean — the reference's own case in its plainest form (BASELINE configs[0]: 500 routers, p2p links): a graph
// without network vertices, without any static row flag (no overloaded source, no zero-cost link from a higher-numbered
// source, in-degrees <= SINGLE_RL) and not hop-count-like (hspf_graph::lean).  One workgroup per root, ONE vertex per
// thread, the vertex' link records in registers for the whole run, the 8-byte words of all vertices in LDS; a sweep is
// one LDS round trip (the sources' words), the two-pass row routine on registers (candidates + minimum, then the tight
// links walked backwards: in-row order makes the first tight link the first discoverer) and one LDS store.  A link out
// of the ROOT contributes its first-hop slot bit instead of the root's (empty) mask: the root's word is 0, so OR-ing the
// precomputed bit into the loaded pay bits is the whole `hops == 0` case.  Sweeps run free in groups of SINGLE_FREE
// between checked ones (see k_single).  k_single carries the general routine next to this path and spills its scalar
// state around every sweep; alone, the loop is ~70 vector instructions.
// RL = link records per thread: 4 when no row has more (the reference's own p2p grid: 1 400 -> ~900 cycles per sweep), else 8.
template <bool MAXINF, int RL>
__global__ __launch_bounds__(SINGLE_THREADS) void k_single_lean(SingleArgs a) {
  extern __shared__ uint64_t s_st[];
  __shared__ int s_changed[4];
  const GraphDev &g = a.gp->g;
  const uint32_t n = g.n;
  const uint32_t root_slot = blockIdx.x;
  const uint32_t my_root = a.roots[root_slot];
  const uint32_t v = threadIdx.x;
  const FusedParams P = a.P;
  const uint32_t mmask = (1u << P.mbits) - 1u;
  const size_t orow = a.o.row(root_slot) * (size_t)n;
  const bool mine = v < n;
  if (my_root == INF) {                            // padding root: empty SPT
    if (mine) {
      a.o.dist[orow + v] = INF;
      if (a.o.hops) a.o.hops[orow + v] = 0;
      if (a.o.flags) a.o.flags[orow + v] = 0;
      if (a.o.mask) for (uint32_t k = 0; k < a.o.out_words; ++k) a.o.mask[(orow + v) * a.o.out_words + k] = 0;
    }
    if (v == 0) a.lane_flags[root_slot] = 0u;
    return;
  }
  const uint64_t t_clk0 = clock64(), t_wall0 = wall_clock64();
  uint32_t ls[RL], lw[RL], lb[RL];
  const uint32_t vc = min(v, n - 1u);
  const uint32_t e0 = g.in_ptr[vc], e1 = mine ? g.in_ptr[vc + 1] : e0;
  const uint32_t v_router = 1u;                    // no network vertices in a lean graph
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)RL; ++k) {
    const bool in = e0 + k < e1;
    const uint32_t e = in ? e0 + k : e0;           // (arrays are padded: e0 is readable even for an empty row)
    const uint32_t sv = g.in_src[e] & SRC_MASK, wv = g.in_w[e], fp = g.in_fpos[e];
    ls[k] = in ? sv : vc;
    lw[k] = in ? wv : INF;                         // padding: a candidate of all ones, never reached
    lb[k] = (in && sv == my_root && fp < P.mbits) ? (1u << fp) : 0u;   // the root's slot base is 0
  }
  uint64_t cur = (mine && v == my_root) ? 0ull : ~0ull;
  if (mine) s_st[v] = cur;
  if (v < 4) s_changed[v] = 0;
  __syncthreads();
  const bool live = mine && v != my_root;
  bool sat = false, ovf = false, need_exact = false;
  const uint32_t max_sweeps = 4u * n + 64u;
  uint32_t sweep = 0;
  const uint64_t t_loop0 = clock64();
  for (;; ++sweep) {
    const bool checked = sweep % (SINGLE_FREE + 1u) == SINGLE_FREE;
    const uint32_t round = sweep / (SINGLE_FREE + 1u);
    if (checked) __syncthreads();
    bool any = false;
    if (live) {
      uint64_t qs[RL];
#pragma unroll
      for (uint32_t k = 0; k < (uint32_t)RL; ++k) qs[k] = s_st[ls[k]];
      uint32_t c[RL];
      bool sat1 = false;
#pragma unroll
      for (uint32_t k = 0; k < (uint32_t)RL; ++k) {
        const uint32_t d = (uint32_t)(qs[k] >> 32);
        c[k] = add_sat(d, lw[k]);
        if (MAXINF && c[k] == INF && d != INF && lw[k] != INF) sat1 = true;
      }
      uint32_t bd = min(min(c[0], c[1]), min(c[2], c[3]));
      if constexpr (RL == 8) bd = min(bd, min(min(c[4], c[5]), min(c[6], c[7])));
      static_assert(RL == 4 || RL == 8, "min tree");
      uint32_t macc = 0u, bpay = 0u;
#pragma unroll
      for (int k = RL - 1; k >= 0; --k) {
        const bool t = c[k] == bd;
        const uint32_t pb = (uint32_t)qs[k] | lb[k];
        macc |= t ? pb : 0u;
        bpay = t ? pb : bpay;
      }
      RowAcc r{bd, macc & mmask, 0u, bpay >> P.mbits, sat1};
      const RowOut<uint64_t> o = finish_row<uint64_t>(r, v, my_root, v_router, INF, P);
      sat = sat || o.sat; ovf = ovf || o.ovf;
      if (o.nw != cur) { cur = o.nw; s_st[v] = o.nw; any = true; }
    }
    if (!checked) continue;
    if (any) s_changed[round & 3u] = 1;
    if (v == 0) s_changed[(round + 2u) & 3u] = 0;
    __syncthreads();
    if (s_changed[round & 3u] == 0) break;
    if (sweep >= max_sweeps) { need_exact = true; break; }
  }
  if (mine) {
    const bool in = cur != ~0ull;
    const uint32_t pay = (uint32_t)cur;
    a.o.dist[orow + v] = in ? (uint32_t)(cur >> 32) : INF;
    if (a.o.hops) a.o.hops[orow + v] = in ? (uint16_t)(pay >> P.mbits) : (uint16_t)0;
    if (a.o.flags) a.o.flags[orow + v] = in ? 1 : 0;
    if (a.o.mask) {
      a.o.mask[(orow + v) * a.o.out_words] = in ? (uint64_t)(pay & mmask) : 0ull;
      for (uint32_t k = 1; k < a.o.out_words; ++k) a.o.mask[(orow + v) * a.o.out_words + k] = 0;
    }
  }
  uint32_t lf = 0;
  if ((MAXINF && sat) || need_exact) lf |= LF_NEED_EXACT;
  if (ovf) lf |= LF_OVERFLOW;
  single_store_flags(a.lane_flags, root_slot, lf, &s_changed[0]);
  if (a.count_rows && v == 0) {
    atomicAdd(&a.gp->rows_done[root_slot & 127u], (sweep + 1u) * n);
    if (root_slot == 0) {
      a.gp->rows_done[128] = sweep + 1u;
      a.gp->rows_done[129] = (uint32_t)(clock64() - t_clk0);
      a.gp->rows_done[130] = (uint32_t)(wall_clock64() - t_wall0);
      a.gp->rows_done[131] = (uint32_t)(t_loop0 - t_clk0);
    }
  }
}

// Everything a fused run starts from, in ONE launch instead of five fills: packed state = all ones, activation stamps =
// 0, per-batch row flags = the graph's static ones (k_init_fused then adds RF_HNB), sweep flags = 0, per-root status
// bits = 0, row counter = 0.  (A run of 64 roots on isis-100k spent ~45 us in seven tiny launches before its first sweep.)
__global__ __launch_bounds__(256) void k_init_fill(uint4 *__restrict__ st16, size_t n_st16, uint32_t fillw, uint32_t *__restrict__ act, size_t n_act,
                                                   const uint8_t *__restrict__ rowflags, uint8_t *__restrict__ hnb, uint32_t n,
                                                   int *__restrict__ changed, uint32_t n_changed,
                                                   uint32_t *__restrict__ lane_flags, uint32_t n_lf, uint32_t *__restrict__ kcnt,
                                                   int guard_sweep, uint32_t *__restrict__ ctl = nullptr, uint32_t n_ctl = 0u) {
  // guard_sweep >= 0: enqueued SPECULATIVELY behind the emit and the flag read-back of a chunk of sweeps, before the host
  // knows whether the chunk reached the fixed point: a chunk whose last sweep still changed something is not over, its
  // state must stay.  (Every thread reads the flag before any thread can have zeroed it to a different value: a
  // converged chunk's flag IS zero.)
  if (guard_sweep >= 0 && changed[guard_sweep] != 0) return;
  const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x, T = (size_t)gridDim.x * 256u;
  const uint4 ones = make_uint4(fillw, fillw, fillw, fillw);            // "not reached": all ones, or FusedParams::infw
  for (size_t i = t; i < n_st16; i += T) st16[i] = ones;
  for (size_t i = t; i < n_act; i += T) { act[i] = 0u; hnb[i] = rowflags[i % n]; }
  for (size_t i = t; i < n_changed; i += T) changed[i] = 0;
  for (size_t i = t; i < n_lf; i += T) lane_flags[i] = 0u;
  if (kcnt && t < 256) kcnt[t] = 0u;
  if (ctl && t < n_ctl) ctl[t] = 0u;                                    // the lean sweep's due / changed counters (LEAN_CTL_WORDS)
}

// init for the fused path: roots' own lanes = (0, 0, 0); their out-neighbours are due in the
// first sweep (id 2); out-neighbours of every vertex that can have hops == 0 for some root of the
// batch (the root and its slot-table networks) are marked for the general row routine.  One wave per root: the lanes
// walk the out-links (a thread per root walked them one dependent load at a time: 13 us).
template <typename ST>
__global__ __launch_bounds__(256) void k_init_fused(GraphDev g, ST *st, uint32_t *act, uint8_t *hnb, const uint32_t *roots,
                                                    SlotTabs tabs, uint32_t n_lanes, uint32_t ns) {
  const uint32_t i = (blockIdx.x * 256u + threadIdx.x) >> 6, ln = threadIdx.x & 63u;
  if (i >= n_lanes) return;
  const uint32_t r = roots[i];
  if (r == INF) return;
  const uint32_t n = g.n;
  const uint32_t batch = i >> 6, lane = i & 63;
  // ns: rows per batch slab (n, or n + 1 for the lean sweep's never-reached pad row).  The root's own row takes the
  // general routine or the lean sweep's root-neighbour path (they keep a root's lane at zero): the plain fast routine never sees a root row.
  // The marks are OR-ed into the byte k_init_fill left there (the graph's static flags), through the aligned word that
  // holds it: a row can be next to one lane's root AND behind another lane's hops-0 network, and a plain byte store of
  // either mark could drop the other.  RF_HNB alone = "some in-neighbour is a root of the batch, nothing else": the
  // lean sweep handles that inline (the root's link contributes its slot bit); RF_HNBN = the general routine.
  auto mark = [&](uint32_t row, uint32_t bits) {
    const size_t idx = (size_t)batch * n + row;
    atomicOr((uint32_t *)(hnb + (idx & ~(size_t)3)), bits << (8u * (uint32_t)(idx & 3u)));
  };
  if (ln == 0) {
    st[((size_t)batch * ns + r) * 64 + lane] = (ST)0;
    mark(r, RF_HNB | RF_ROOT);
  }
  for (uint32_t k = g.out_ptr[r] + ln; k < g.out_ptr[r + 1]; k += 64) {
    const uint32_t d = g.out_dst[k];
    act[(size_t)batch * n + d] = 2u;
    mark(d, RF_HNB);
  }
  for (uint32_t j = tabs.ptr[i]; j < tabs.ptr[i + 1]; ++j) {
    const uint32_t h = tabs.vtx[j];
    for (uint32_t k = g.out_ptr[h] + ln; k < g.out_ptr[h + 1]; k += 64) mark(g.out_dst[k], RF_HNB | RF_HNBN);
  }
}

// Rows of LEAF ROOTS (run_classes, spf_capi.hip): a root h whose only kept link leads to a router p that is a root of the
// same call has p's tree behind that one link — every path from h starts h -> p, and h is on no path of p's tree but its
// own: distance + cost of the link (beyond the max-path metric: not reached, as the reference prunes the relaxation that
// would exceed it: holo-isis/src/spf.rs:640-646), hops + 1 (p is a router; u16 saturating), ONE first-hop slot — the link's
// position in h's row — for every vertex reached, and (0, 0, no next hop) for h itself.  Reads p's finished row of the
// caller's tables, writes h's: no fixed point.  grid = (ceil(n / 256), jobs).
struct LeafRootJob { uint32_t h, p_row, h_row, link, fpos; };
__global__ __launch_bounds__(256) void k_leaf_root_rows(uint32_t n, const LeafRootJob *__restrict__ jobs, const uint32_t *__restrict__ row_ptr_raw,
                                                        const uint32_t *__restrict__ metric_raw,
                                                        uint32_t maxpath, uint32_t W, uint32_t *dist, uint16_t *hops, uint16_t *flags,
                                                        uint64_t *mask) {
  const LeafRootJob j = jobs[blockIdx.y];
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  if (v >= n) return;
  const uint32_t w = metric_raw[row_ptr_raw[j.h] + j.fpos];      // (j.link: the link in the HOST's row pool, not an index into the device's raw CSR)
  const size_t pi = (size_t)j.p_row * n + v, hi = (size_t)j.h_row * n + v;
  const uint32_t pd = dist[pi];
  const uint64_t sum = (uint64_t)pd + w;
  const bool self = v == j.h;
  const bool reach = self || (pd != INF && sum <= (uint64_t)maxpath);
  dist[hi] = self ? 0u : (reach ? (uint32_t)sum : INF);
  if (hops) hops[hi] = (self || !reach) ? (uint16_t)0 : (uint16_t)min((uint32_t)hops[pi] + 1u, 0xFFFFu);
  if (flags) flags[hi] = self ? (uint16_t)1 : (reach ? (uint16_t)(flags[pi] | 1u) : (uint16_t)0);
  if (mask)
    for (uint32_t q = 0; q < W; ++q)
      mask[hi * W + q] = (!self && reach && (j.fpos >> 6) == q) ? (1ull << (j.fpos & 63u)) : 0ull;
}

__global__ void k_clear_lane_flag(uint32_t *lane_flags, uint32_t n_lf, uint32_t bit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_lf) lane_flags[i] &= ~bit;
}

// Emit for the fused path: packed lane-major state -> row-major results.
// reset_guard >= 0: the emit is also the first half of the NEXT run's scratch fill — a tile that has been read is
// overwritten with "not reached" (fillw), provided the chunk of sweeps in front of this emit reached the fixed point
// (changed[reset_guard] == 0: the flag k_init_fill's speculative launch tests, which then skips the 25.6 MB state slab:
// 14 us -> ~3 us per run).  An emit behind a chunk that had not converged leaves the state alone.
template <typename ST>
__global__ __launch_bounds__(256) void k_emit_fused(uint32_t n, uint32_t n_roots, ST *__restrict__ st,
                                                    FusedParams P, OutDev o, uint32_t ns, uint32_t *lane_flags,
                                                    const int *__restrict__ changed, int reset_guard, uint32_t fillw) {
  __shared__ ST tt[64][65];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t batch = blockIdx.y, v0 = blockIdx.x * 64;
  const uint32_t nv = min(64u, n - v0);
  const uint32_t r0 = batch * 64;
  const uint32_t nr = min(64u, n_roots - r0);
  ST *S = st + ((size_t)batch * ns + v0) * 64;
  const bool reset = reset_guard >= 0 && changed[reset_guard] == 0;
  const ST fillv = sizeof(ST) == 8 ? (ST)(((uint64_t)fillw << 32) | fillw) : (ST)fillw;
  // lane_flags != null (lean sweep, 4-byte state): the fields are checked HERE, once per final word, instead of in
  // every row evaluation — a reached lane within one link cost of "not reached" (P.ovf_t), or a hop count that
  // saturated (P.hmax), means some value on the way may have been clipped: LF_OVERFLOW, the run is redone wide.  Clipped
  // TRANSIENT values are harmless (a weaker upper bound; every evaluation is a pure function of the in-neighbours).
  bool ovf = false;
  for (uint32_t j = wave; j < nv; j += 4) {
    const ST x = S[(size_t)j * 64 + lane];
    tt[j][lane] = x;
    if (reset) S[(size_t)j * 64 + lane] = fillv;
    if (sizeof(ST) == 4 && lane_flags)
      ovf = ovf || ((uint32_t)x < P.inf_t && ((uint32_t)x >= P.ovf_t || (((uint32_t)x >> P.mbits) & P.hmax) == P.hmax));
  }
  if (ovf) atomicOr(&lane_flags[r0 + lane], LF_OVERFLOW);
  __syncthreads();
  for (uint32_t r = wave; r < nr; r += 4)
    if (lane < nv) {
      const ST x = tt[lane][r];
      const size_t idx = o.row(r0 + r) * n + v0 + lane;
      if (o.packed) { ((ST *)o.packed)[idx] = x; continue; }       // the word is the result (include/holo_spf_hip.h "packed results")
      uint32_t d, pay;
      if (sizeof(ST) == 8) { d = (uint32_t)((uint64_t)x >> 32); pay = (uint32_t)x; }
      else { d = (uint32_t)x >> P.sh; pay = (uint32_t)x & ((1u << P.sh) - 1u); }
      const bool in = sizeof(ST) == 8 ? x != (ST)~(ST)0 : (uint32_t)x < P.inf_t;
      o.dist[idx] = in ? d : INF;
      if (o.hops) o.hops[idx] = in ? (uint16_t)(pay >> P.mbits) : (uint16_t)0;
      if (o.flags) o.flags[idx] = in ? 1 : 0;
      if (o.mask) {
        o.mask[idx * o.out_words] = in ? (uint64_t)(pay & ((1u << P.mbits) - 1u)) : 0ull;
        for (uint32_t k = 1; k < o.out_words; ++k) o.mask[idx * o.out_words + k] = 0;
      }
    }
}

// hspf_run_packed for the rows that do not come out of the fused emit — the one-workgroup and lane = vertex kernels write
// row-major tables, so does the sequential kernel for the roots it takes over —: row-major (dist, hops, flags, mask word 0)
// -> packed words of the run's layout (FusedParams P; WB = 4: [dist | .. | hops | mask] with dist at P.sh, WB = 8:
// [dist32 | hops | mask]).  rows == null: all n_rows rows, else the listed ones.  A value that does not fit its field
// raises *misfit (the call then returns HSPF_E_NO_PACKED: nothing a caller could mistake for a result).
template <int WB>
__global__ __launch_bounds__(256) void k_pack_full(uint32_t n, uint32_t n_rows, const uint32_t *__restrict__ rows, const uint32_t *__restrict__ dist,
                                                   const uint16_t *__restrict__ hops, const uint16_t *__restrict__ flags,
                                                   const uint64_t *__restrict__ mask, uint32_t mask_words, FusedParams P, void *out, uint32_t *misfit) {
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  if (v >= n || blockIdx.y >= n_rows) return;
  const size_t idx = (size_t)(rows ? rows[blockIdx.y] : blockIdx.y) * n + v;
  const bool in = (flags[idx] & 1u) != 0;
  const uint32_t d = dist[idx], h = hops[idx];
  const uint64_t m = mask[idx * mask_words];
  bool bad = false;
  if (WB == 8) {
    uint64_t w = ~0ull;
    if (in) {
      bad = h > P.hmax || (m >> P.mbits) != 0ull;
      w = ((uint64_t)d << 32) | ((uint64_t)(h & P.hmax) << P.mbits) | (m & ((1ull << P.mbits) - 1ull));
      bad = bad || w == ~0ull;
    }
    ((uint64_t *)out)[idx] = w;
  } else {
    uint32_t w = P.infw;
    if (in) {
      const uint32_t dmax = (P.ovf_t >> P.sh);                     // reached distances stay below the overflow threshold
      bad = d >= dmax || h >= P.hmax || (m >> P.mbits) != 0ull;
      w = (d << P.sh) | ((h & P.hmax) << P.mbits) | (uint32_t)(m & ((1ull << P.mbits) - 1ull));
      bad = bad || w >= P.inf_t;
    }
    ((uint32_t *)out)[idx] = w;
  }
  if (bad) atomicOr(misfit, 1u);
}

// Epoch rebase (only when a DAG phase needs more than 65534 launches): every final row -> epoch 1.
__global__ void k_rebase(uint32_t *hv, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t x = hv[i];
  if (x >> HV_EPOCH_SHIFT) hv[i] = (x & 0xFFFFu) | (1u << HV_EPOCH_SHIFT);
}

// ---------------------------------------------------------------------------------------------
// Emit: lane-major state -> row-major results.  One block = 64 vertices x 64 roots of one batch.
// grid = (ceil(n/64), n_batches), block = 256.

// k_fw<.., LEAF>: "not reached" in the rows that take part (the leaves' rows are never read or written: on a fat-tree
// with single-homed hosts that is 95 % of the 800 MB the three fills of the other path write).
template <int W>
__global__ __launch_bounds__(256) void k_init_fw_state(uint32_t n, uint32_t n_batches, const uint8_t *__restrict__ leaf,
                                                       uint32_t *__restrict__ dist, uint32_t *__restrict__ hv, uint64_t *__restrict__ mask) {
  const uint32_t lane = threadIdx.x & 63u;
  const size_t rows = (size_t)n * n_batches;
  for (size_t r = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * 4u) {
    if (leaf[r % n]) continue;
    dist[r * 64 + lane] = INF;
    hv[r * 64 + lane] = 0u;
#pragma unroll
    for (int q = 0; q < W; ++q) mask[(r * W + q) * 64 + lane] = 0ull;
  }
}

// The row of a leaf v (one in-link, from u): what k_fw's visit of the row would store — fw_chunk over that one link,
// then the finish — from u's FINAL triple (d_in, h_in, mu = word q of its mask; not read when u is a leaf itself).  The
// zero-cost rule of fused_row_any (a zero-cost link from a higher-numbered source is left to the sequential kernel)
// guards the order in which several tight parents were found; one link has no order to guard, so it is taken as it is.
template <int W, bool WANTMASK>
__device__ __forceinline__ void fw_leaf_apply(const FusedGraph *__restrict__ gp, uint32_t v, uint32_t sraw, uint32_t w, uint32_t fpos,
                                              uint32_t v_router, uint32_t d_in, uint32_t h_in, uint64_t mu, uint32_t q,
                                              uint32_t my_root, uint32_t root_slot, uint32_t maxpath, uint32_t net_nexthops,
                                              uint32_t ignore_ovl, uint32_t &nd, uint32_t &nh, uint64_t &nm) {
  const uint32_t u = sraw & SRC_MASK;
  uint32_t d = d_in, hh = h_in & 0xFFFFu;
  if (sraw & SRC_LEAF) { d = (u == my_root) ? 0u : INF; hh = 0u; mu = 0ull; }   // the neighbour is a leaf itself (a two-vertex component)
  if (!ignore_ovl && (sraw & SRC_NO_TRANSIT) && u != my_root) d = INF;           // overloaded source
  const uint32_t c = add_sat(d, w);
  nd = (v == my_root) ? 0u : ((c == INF || c > maxpath) ? INF : c);
  const bool live = v != my_root && nd != INF;
  nh = live ? min(hh + v_router, 0xFFFFu) : 0u;
  nm = 0ull;
  if (WANTMASK) {
    nm = (live && hh != 0u) ? mu : 0ull;
    if (__ballot(live && hh == 0u) != 0ull) {                                     // parent: root or hops-0 network -> the link's own slot
      if (live && hh == 0u) {
        const uint32_t base_s = (u == my_root) ? 0u : slot_base_of(gp->tabs, root_slot, u);
        const uint32_t sidx = base_s + fpos;
        const bool on = (v_router || net_nexthops) && sidx < (uint32_t)W * 64u;
        nm = (on && (sidx >> 6) == q) ? (1ull << (sidx & 63u)) : 0ull;
      }
    }
  }
}

// LEAF: the rows of leaves (GraphDev::leaf) were left out of the fixed point; their triples are derived here, per lane,
// from the neighbour's final one (fw_leaf_row) on the way into the transposition tile — once per pass (distance, hops,
// each mask word): the neighbour's row is shared by all the leaves behind it and stays in L1 / L2, so deriving again
// is cheaper than holding 2 + 2 W values per vertex across the passes.
struct EmitLeaf {
  const FusedGraph *gp;
  const uint32_t *roots;
  uint32_t maxpath, net_nexthops, ignore_ovl;
};

// grid = (ceil(n / 64), batches, 1 + W), independent blocks instead of passes of one block (a block that walked distance,
// hops and every mask word one after the other, two barriers each, wrote 321 MB in 325 us):
//   z == 0      distances, in-SPT flags and hops of 64 vertices x 64 roots
//   z == 1 + s  ALL W mask words of the 64 / W vertices v0 + s * 64 / W ..: tile row l = (vertex l / W, word l % W), so
//               that a root's 64 values are one contiguous 512-byte run of its output row (a block per mask word wrote
//               every other 8 bytes of it)
template <int W, bool LEAF>
__global__ __launch_bounds__(256) void k_emit(uint32_t n, uint32_t n_roots,
                                              const uint32_t *__restrict__ dist,
                                              const uint32_t *__restrict__ hv,
                                              const uint64_t *__restrict__ mask, OutDev o, EmitLeaf el) {
  __shared__ uint64_t t64[64][65];
  uint32_t (*td)[65] = (uint32_t (*)[65])&t64[0][0];          // z == 0: two 32-bit tiles in the same space
  uint32_t (*th)[65] = (uint32_t (*)[65])&t64[32][0];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t batch = blockIdx.y, v0 = blockIdx.x * 64;
  const uint32_t nv = min(64u, n - v0);
  const uint32_t r0 = batch * 64;
  const uint32_t nr = min(64u, n_roots - r0);
  const uint32_t *Db = dist + (size_t)batch * n * 64, *Hb = hv + (size_t)batch * n * 64;
  const uint64_t *Mb = mask + (size_t)batch * n * 64 * W;
  const uint32_t root_slot = r0 + lane;
  const uint32_t my_root = LEAF ? el.roots[root_slot] : 0u;
  // LEAF: lane j holds what the derivation of vertex v0 + j needs (its one in-link), read once, side by side
  uint32_t m_src = 0u, m_w = 0u, m_fpos = 0u, m_fl = 0u;       // m_fl: bit0 leaf, bit1 router
  if (LEAF && lane < nv) {
    const GraphDev &g = el.gp->g;
    const uint32_t v = v0 + lane;
    m_fl = (g.leaf[v] ? 1u : 0u) | ((g.vflags[v] & 1u) ? 0u : 2u);
    if (m_fl & 1u) { const uint32_t e0 = g.in_ptr[v]; m_src = g.in_src[e0]; m_w = g.in_w[e0]; m_fpos = g.in_fpos[e0]; }
  }
  // the state row a tile row is read from: its own, or (leaf) its neighbour's
  auto src_row = [&](uint32_t j) -> uint32_t {
    if (!LEAF) return v0 + j;
    const uint32_t fl = rdlane(m_fl, j), sraw = rdlane(m_src, j);
    return ((fl & 1u) && !(sraw & SRC_LEAF)) ? (sraw & SRC_MASK) : v0 + j;
  };
  if (blockIdx.z == 0) {
    // dist + flags (in SPT <=> dist != INF on this path; saturated roots go through k_exact) + hops
#pragma unroll 1
    for (uint32_t jb = wave; jb < nv; jb += 32u) {             // 8 tile rows per step: their loads are in flight together
      uint32_t dd[8], hh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t j = min(jb + 4u * i, nv - 1u), row = src_row(j);
        dd[i] = Db[(size_t)row * 64 + lane];
        hh[i] = (o.hops || LEAF) ? Hb[(size_t)row * 64 + lane] : 0u;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t j = jb + 4u * i;
        if (j >= nv) break;
        uint32_t nd = dd[i], nh = hh[i];
        if (LEAF && (rdlane(m_fl, j) & 1u)) {
          uint64_t nm;
          fw_leaf_apply<W, false>(el.gp, v0 + j, rdlane(m_src, j), rdlane(m_w, j), rdlane(m_fpos, j), (rdlane(m_fl, j) >> 1) & 1u, dd[i], hh[i],
                                  0ull, 0u, my_root, root_slot, el.maxpath, el.net_nexthops, el.ignore_ovl, nd, nh, nm);
        }
        td[j][lane] = nd; th[j][lane] = nh;
      }
    }
    __syncthreads();
    for (uint32_t r = wave; r < nr; r += 4)
      if (lane < nv) {
        const uint32_t x = td[lane][r];
        const size_t idx = o.row(r0 + r) * n + v0 + lane;
        o.dist[idx] = x;
        if (o.flags) o.flags[idx] = (x != INF) ? 1 : 0;
        if (o.hops) o.hops[idx] = (uint16_t)(th[lane][r] & 0xFFFFu);
      }
    return;
  }
  if (!o.mask) return;
  // W is the run's word count rounded up to 1/2/4/8/16; the caller's rows have out_words >= the words really needed
  // (maybe fewer than W, e.g. 3): the surplus internal words are zero and must not be written past a row's end
  constexpr uint32_t VB = 64u / (uint32_t)W;                   // vertices of this block
  const uint32_t vb0 = (blockIdx.z - 1u) * VB;                 // first of them inside the tile
  if (vb0 >= nv) return;
  const uint32_t nvb = min(VB, nv - vb0), nl = nvb * W;
#pragma unroll 1
  for (uint32_t lb = wave; lb < nl; lb += 32u) {
    uint32_t dd[8], hh[8];
    uint64_t mm[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t l = min(lb + 4u * i, nl - 1u), j = vb0 + l / W, k = l % W, row = src_row(j);
      mm[i] = Mb[((size_t)row * W + k) * 64 + lane];
      dd[i] = LEAF ? Db[(size_t)row * 64 + lane] : 0u;
      hh[i] = LEAF ? Hb[(size_t)row * 64 + lane] : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t l = lb + 4u * i;
      if (l >= nl) break;
      const uint32_t j = vb0 + l / W, k = l % W;
      uint64_t nm = mm[i];
      if (LEAF && (rdlane(m_fl, j) & 1u)) {
        uint32_t nd, nh;
        fw_leaf_apply<W, true>(el.gp, v0 + j, rdlane(m_src, j), rdlane(m_w, j), rdlane(m_fpos, j), (rdlane(m_fl, j) >> 1) & 1u, dd[i], hh[i],
                               mm[i], k, my_root, root_slot, el.maxpath, el.net_nexthops, el.ignore_ovl, nd, nh, nm);
      }
      t64[l][lane] = nm;
    }
  }
  __syncthreads();
  {
    const uint32_t j = lane / W, k = lane % W;
    for (uint32_t r = wave; r < nr; r += 4) {
      const size_t base = (o.row(r0 + r) * n + v0 + vb0) * o.out_words;
      if (j < nvb && k < o.out_words) o.mask[base + (size_t)j * o.out_words + k] = t64[lane][r];
      // words beyond W (caller capacity larger than needed) are zero
      if (o.out_words > (uint32_t)W && lane < nvb)
        for (uint32_t kk = W; kk < o.out_words; ++kk) o.mask[base + (size_t)lane * o.out_words + kk] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_fw<W> — the fused fixed point for runs whose roots have MORE first-hop slots than a packed state word holds
// (> 24: routers on a big LAN, fat-tree switches): distances, hops and W mask words in ONE label-correcting sweep
// instead of a distance phase followed by the epoch-driven DAG phase (k_relax + k_dag: every row is walked in both
// phases, and a DAG sweep only accepts parents finalised by an EARLIER launch).  The reference merges next hops
// without any width limit (holo-isis/src/spf.rs:702-704); this is the same merge with the set as a W x 64 bit mask.
//
// State = the three lane-major arrays of the two-phase path (dist u32, hv u32 = hops, mask u64 x W), so k_init_roots
// and k_emit<W> serve both.  A (distance, hops, mask) triple is therefore NOT one memory access and a reader can see a
// neighbour's new distance with its old mask.  That is harmless for the result: every recomputation of a row is a pure
// function of what it read; a row that changes stamps its out-neighbours for the NEXT launch, so every row is
// recomputed at least once after the last change of any in-neighbour, in a later launch, where it reads that
// neighbour's final triple whole; and the run ends only after a launch that changed nothing.  The tight links that
// feed masks always come from a vertex that precedes the row in (distance, index) order (zero-cost links from
// higher-numbered sources are excluded, exactly as in fused_row_any), so the final masks are determined by induction
// over that order, whatever transient values were seen on the way.
//
// Activation: act[v] >= sweep + 1 <=> row v is due in this sweep; k_init_fw stamps the roots' out-neighbours with 1,
// a row that changes in sweep s stamps s + 2 on its out-neighbours (due next sweep — and in this one if not yet
// visited).  Row routine: the per-lane accumulator of fused_row_any with W-word masks; hops and masks of a neighbour
// are only requested when its candidate is <= the running minimum for some lane.
// Row accumulator of k_fw: RowAcc with W-word masks + the zero-cost side state.
template <int W, bool HC> struct FwAcc {
  uint32_t bd, bpd, bh, bd_all, zb, zh;
  bool sat;
  uint64_t am[W];
  uint64_t zm[HC ? W : 1];
};
template <int W, bool HC> __device__ __forceinline__ FwAcc<W, HC> fw_acc_init() {
  FwAcc<W, HC> x;
  x.bd = INF; x.bpd = INF; x.bh = 0u; x.bd_all = INF; x.zb = INF; x.zh = 0u; x.sat = false;
#pragma unroll
  for (int q = 0; q < W; ++q) x.am[q] = 0ull;
#pragma unroll
  for (int q = 0; q < (HC ? W : 1); ++q) x.zm[q] = 0ull;
  return x;
}

// Up to 64 links (lane j of sv / wv = link eb + j) into x: the per-lane accumulator of fused_row_any with W-word masks;
// the whole triple (distance, hops, W mask words) of DG links is requested in ONE round trip.  The sweep is bound by
// dependent round trips, not bytes, on most rows: asking for the hops and masks only after the distances have shown
// which links are tight — what k_dag does — doubles the chain.
#ifndef FW_DG2
#define FW_DG2 8
#endif
template <int W, bool MAXINF, bool HC, bool LEAF = false>
__device__ __forceinline__ void fw_chunk_1p(FwAcc<W, HC> &x, const FusedGraph *__restrict__ gp, const uint32_t *D, const uint32_t *H,
                                         const uint64_t *M, uint32_t v, uint32_t v_router, uint32_t eb, uint32_t cnt,
                                         uint32_t sv, uint32_t wv, uint32_t lane, uint32_t my_root, uint32_t root_slot,
                                         uint32_t net_nexthops, uint32_t ignore_ovl) {
  constexpr int DG = W <= 2 ? FW_DG2 : (W <= 4 ? 4 : (W <= 8 ? 2 : 1));
  const GraphDev &g = gp->g;
  const uint32_t lane4 = lane * 4u, lane8 = lane * 8u;
  const bool has_nt = !ignore_ovl && __ballot(lane < cnt && (sv & SRC_NO_TRANSIT) != 0) != 0ull;
  const uint32_t zv = (lane < cnt && wv == 0u && (sv & SRC_MASK) >= v) ? 1u : 0u;
  const bool has_z = __ballot(zv != 0u) != 0ull;
#pragma unroll 1
  for (uint32_t j0 = 0; j0 < cnt; j0 += DG) {
    uint32_t du[DG], hu[DG];
    uint64_t mu[DG][W];
#pragma unroll
    for (int k = 0; k < DG; ++k) {
      const uint32_t sraw = rdlane(sv, min(j0 + (uint32_t)k, 63u)), u = sraw & SRC_MASK;
      const bool in = (j0 + k) < cnt && !(LEAF && (sraw & SRC_LEAF));   // uniform: no requests for a short row's padding, nor for a leaf's row (not kept)
      du[k] = in ? ld_row(D, u * 256u + lane4) : INF;
      hu[k] = in ? ld_row(H, u * 256u + lane4) : 0u;
#pragma unroll
      for (int q = 0; q < W; ++q) mu[k][q] = in ? ld_row64(M, ((size_t)u * W + q) * 512u + lane8) : 0ull;
    }
    uint32_t cc[DG];
#pragma unroll
    for (int k = 0; k < DG; ++k) {
      const uint32_t j = min(j0 + (uint32_t)k, 63u);
      const uint32_t w = rdlane(wv, j);
      uint32_t d = du[k];
      if (LEAF) {
        const uint32_t sw = rdlane(sv, j);
        if (sw & SRC_LEAF) d = ((sw & SRC_MASK) == my_root) ? 0u : INF;      // a leaf is on nobody's path but its own (hops 0, mask 0 were not requested)
      }
      if (has_nt) {
        const uint32_t sw = rdlane(sv, j);
        if ((sw & SRC_NO_TRANSIT) && (sw & SRC_MASK) != my_root) d = INF;     // overloaded source
      }
      du[k] = d;
      cc[k] = (j0 + k) < cnt ? add_sat(d, w) : INF;
      if (MAXINF && cc[k] == INF && d != INF && w != INF) x.sat = true;
    }
#pragma unroll
    for (int k = 0; k < DG; ++k) {
      const uint32_t j = j0 + k;
      if (j >= cnt) break;
      const uint32_t c = cc[k], d = du[k];
      if (LEAF && (rdlane(sv, j) & SRC_LEAF) && __ballot(c != INF) == 0ull) continue;   // a leaf that is no root of this batch
      const bool zlink = has_z && rdlane(zv, j) != 0u;              // uniform
      if (zlink && !HC) { x.bd_all = min(x.bd_all, c); continue; }
      const bool hz = HC && zlink;
      const bool lt = hz ? (c < x.zb) : (c < x.bd), eq = !hz && c == x.bd && c != INF;
      const uint32_t hh = hu[k] & 0xFFFFu;
      uint64_t contrib[W];
#pragma unroll
      for (int q = 0; q < W; ++q) contrib[q] = mu[k][q];
      const bool direct = (lt || eq) && hh == 0u && c != INF;      // parent: root or hops-0 network
      if (__ballot(direct) != 0ull) {
        const uint32_t u = rdlane(sv, j) & SRC_MASK;
        const uint32_t fpos = g.in_fpos[eb + j];
        if (direct) {
          const uint32_t base_s = (u == my_root) ? 0u : slot_base_of(gp->tabs, root_slot, u);
          const uint32_t sidx = base_s + fpos;
          const bool on = (v_router || net_nexthops) && sidx < (uint32_t)W * 64u;
#pragma unroll
          for (int q = 0; q < W; ++q) contrib[q] = (on && (sidx >> 6) == (uint32_t)q) ? (1ull << (sidx & 63u)) : 0ull;
        }
      }
      if (hz) {
        if (lt) {
          x.zb = c; x.zh = hh;
#pragma unroll
          for (int q = 0; q < (HC ? W : 1); ++q) x.zm[q] = contrib[HC ? q : 0];
        }
        continue;
      }
#pragma unroll
      for (int q = 0; q < W; ++q) x.am[q] = lt ? contrib[q] : (eq ? (x.am[q] | contrib[q]) : x.am[q]);
      const bool newp = lt || (eq && d < x.bpd);
      x.bpd = newp ? d : x.bpd;
      x.bh = newp ? hh : x.bh;
      x.bd = min(x.bd, c);
    }
  }
}

template <int W, bool MAXINF, bool HC, bool LEAF = false>
__device__ __forceinline__ void fw_chunk(FwAcc<W, HC> &x, const FusedGraph *__restrict__ gp, const uint32_t *D, const uint32_t *H,
                                         const uint64_t *M, uint32_t v, uint32_t v_router, uint32_t eb, uint32_t cnt,
                                         uint32_t sv, uint32_t wv, uint32_t lane, uint32_t my_root, uint32_t root_slot,
                                         uint32_t net_nexthops, uint32_t ignore_ovl) {
  fw_chunk_1p<W, MAXINF, HC, LEAF>(x, gp, D, H, M, v, v_router, eb, cnt, sv, wv, lane, my_root, root_slot, net_nexthops, ignore_ovl);
}

// y covers links that FOLLOW x's in row order (any_merge with W-word masks).  Candidates equal to INF never merge masks
// (fw_chunk: eq requires c != INF).
template <int W, bool HC>
__device__ __forceinline__ void fw_merge(FwAcc<W, HC> &x, const FwAcc<W, HC> &y) {
  const bool lt = y.bd < x.bd, eq = y.bd == x.bd && y.bd != INF;
  const bool newp = lt || (eq && y.bpd < x.bpd);
#pragma unroll
  for (int q = 0; q < W; ++q) x.am[q] = lt ? y.am[q] : (eq ? (x.am[q] | y.am[q]) : x.am[q]);
  x.bpd = newp ? y.bpd : x.bpd;
  x.bh = newp ? y.bh : x.bh;
  x.bd = min(x.bd, y.bd);
  x.sat = x.sat || y.sat;
  x.bd_all = min(x.bd_all, y.bd_all);
  const bool zl = y.zb < x.zb;
#pragma unroll
  for (int q = 0; q < (HC ? W : 1); ++q) x.zm[q] = zl ? y.zm[q] : x.zm[q];
  x.zh = zl ? y.zh : x.zh; x.zb = min(x.zb, y.zb);
}

// Slices of a giant row on the wide-mask path (see k_giant_part): one WAVE per 64 links, its accumulator stored as
// [6 words x 64 lanes | am: W x 64 u64 | zm: W x 64 u64]; the tags are those of FusedGraph::giant_part.
template <int W> __host__ __device__ constexpr size_t fw_part_bytes() { return 6u * 64u * 4u + 2u * (size_t)W * 64u * 8u; }

template <int W, bool HC>
__device__ __forceinline__ void fw_part_store(char *p, const FwAcc<W, HC> &x, uint32_t lane) {
  uint32_t *w = (uint32_t *)p;
  w[0 * 64 + lane] = x.bd; w[1 * 64 + lane] = x.bpd; w[2 * 64 + lane] = x.bh | (x.sat ? 0x80000000u : 0u);
  w[3 * 64 + lane] = x.bd_all; w[4 * 64 + lane] = x.zb; w[5 * 64 + lane] = x.zh;
  uint64_t *m = (uint64_t *)(p + 6 * 64 * 4);
#pragma unroll
  for (int q = 0; q < W; ++q) { m[q * 64 + lane] = x.am[q]; m[(W + q) * 64 + lane] = HC ? x.zm[HC ? q : 0] : 0ull; }
}
template <int W, bool HC>
__device__ __forceinline__ FwAcc<W, HC> fw_part_load(const char *p, uint32_t lane) {
  const uint32_t *w = (const uint32_t *)p;
  FwAcc<W, HC> x;
  const uint32_t w2 = w[2 * 64 + lane];
  x.bd = w[0 * 64 + lane]; x.bpd = w[1 * 64 + lane]; x.bh = w2 & 0x7FFFFFFFu; x.sat = (w2 >> 31) != 0u;
  x.bd_all = w[3 * 64 + lane]; x.zb = w[4 * 64 + lane]; x.zh = w[5 * 64 + lane];
  const uint64_t *m = (const uint64_t *)(p + 6 * 64 * 4);
#pragma unroll
  for (int q = 0; q < W; ++q) x.am[q] = m[q * 64 + lane];
#pragma unroll
  for (int q = 0; q < (HC ? W : 1); ++q) x.zm[q] = HC ? m[(W + q) * 64 + lane] : 0ull;
  return x;
}

template <int W, bool MAXINF, bool HC, bool LEAF = false>
__global__ __launch_bounds__(256) void k_fw_giant_part(const FusedGraph *__restrict__ gp, const uint32_t *__restrict__ dist,
                                                       const uint32_t *__restrict__ hv, const uint64_t *__restrict__ mask,
                                                       const uint32_t *__restrict__ act, const uint32_t *__restrict__ roots,
                                                       uint32_t net_nexthops, uint32_t ignore_ovl, const int *changed, int sweep) {
  if (sweep > 0 && changed[sweep - 1] == 0) return;
  const GraphDev &g = gp->g;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slice = blockIdx.x, batch = blockIdx.y, n = g.n;
  uint32_t gi = 0;
  { uint32_t lo = 0, hi = g.n_giant;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (g.giant_slice0[mid] <= slice) lo = mid; else hi = mid; }
    gi = lo; }
  const uint32_t v = g.giant_vtx[gi];
  if (act[(size_t)batch * n + v] < (uint32_t)sweep + 1u) return;          // not due in this sweep (k_fw: due <=> stamp >= sweep + 1)
  const uint32_t s_in_row = slice - g.giant_slice0[gi];
  const uint32_t e0 = g.in_ptr[v], e1 = g.in_ptr[v + 1];
  const uint32_t eb = e0 + s_in_row * GIANT_SLICE + wave * 64u;
  const uint32_t cnt = eb < e1 ? min(64u, e1 - eb) : 0u;
  const uint32_t root_slot = batch * 64 + lane;
  const uint32_t my_root = roots[root_slot];
  const uint32_t *D = dist + (size_t)batch * n * 64;
  const uint32_t *H = hv + (size_t)batch * n * 64;
  const uint64_t *M = mask + (size_t)batch * n * 64 * W;
  const uint32_t v_router = (g.vflags[v] & 1u) ? 0u : 1u;
  FwAcc<W, HC> x = fw_acc_init<W, HC>();
  if (cnt) {
    const uint32_t sv = lane < cnt ? g.in_src[eb + lane] : v;
    const uint32_t wv = lane < cnt ? g.in_w[eb + lane] : INF;
    fw_chunk<W, MAXINF, HC, LEAF>(x, gp, D, H, M, v, v_router, eb, cnt, sv, wv, lane, my_root, root_slot, net_nexthops, ignore_ovl);
  }
  const uint32_t B = gridDim.y, n_ws = g.giant_slice0[g.n_giant] * 4u;
  char *base = (char *)(gp->giant_part + giant_tag_words(B, g.n_giant));
  fw_part_store<W, HC>(base + ((size_t)batch * n_ws + (size_t)slice * 4u + wave) * fw_part_bytes<W>(), x, lane);
  if (s_in_row == 0u && wave == 0u && lane == 0u) gp->giant_part[(size_t)batch * g.n_giant + gi] = (uint32_t)sweep + 1u;
}

template <int W, bool MAXINF, bool HC, bool LEAF = false>
__global__ __launch_bounds__(256) void k_fw(const FusedGraph *__restrict__ gp, uint32_t *__restrict__ dist,
                                            uint32_t *__restrict__ hv, uint64_t *__restrict__ mask,
                                            uint32_t *__restrict__ act, const uint32_t *__restrict__ roots,
                                            uint32_t maxpath, uint32_t net_nexthops, uint32_t ignore_ovl,
                                            int *changed, int sweep, uint32_t *lane_flags) {
  if (sweep > 0 && changed[sweep - 1] == 0) return;
  const GraphDev &g = gp->g;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t batch = blockIdx.y;
  const uint32_t n = g.n;
  uint32_t vbeg, nrows;
  if (!wave_rows(g, blockIdx.x, wave, vbeg, nrows)) return;
  uint32_t *A = act + (size_t)batch * n;
  const uint32_t av = A[min(vbeg + min(lane, (uint32_t)VPW - 1u), n - 1)];
  const uint32_t lfv = LEAF ? (uint32_t)g.leaf[min(vbeg + min(lane, (uint32_t)VPW - 1u), n - 1)] : 0u;   // leaves wait for the emit
  const uint64_t due = __ballot(lane < nrows && vbeg + lane < n && av >= (uint32_t)sweep + 1u && lfv == 0u);
  if (due == 0ull) return;
  const uint32_t *__restrict__ in_src = g.in_src;
  const uint32_t *__restrict__ in_w = g.in_w;
  const uint32_t root_slot = batch * 64 + lane;
  const uint32_t my_root = roots[root_slot];
  uint32_t *D = dist + (size_t)batch * n * 64;
  uint32_t *H = hv + (size_t)batch * n * 64;
  uint64_t *M = mask + (size_t)batch * n * 64 * W;
  const uint32_t lane4 = lane * 4u, lane8 = lane * 8u;
  const uint32_t pv = g.in_ptr[min(vbeg + min(lane, (uint32_t)VPW), n)];
  const uint32_t po = g.out_ptr[min(vbeg + min(lane, (uint32_t)VPW), n)];
  bool any = false, sat = false, dyn = false;
#pragma unroll 1
  for (int i = 0; i < VPW; ++i) {
    const uint32_t v = vbeg + i;
    if (v >= n) break;
    if (!((due >> i) & 1ull)) continue;
    const uint32_t e0 = rdlane(pv, i), e1 = rdlane(pv, i + 1);
    const uint32_t v_router = (g.vflags[v] & 1u) ? 0u : 1u;
    // the row's own triple (to see whether anything changes)
    const uint32_t od = ld_row(D, v * 256u + lane4), oh = ld_row(H, v * 256u + lane4);
    uint64_t om[W];
#pragma unroll
    for (int q = 0; q < W; ++q) om[q] = ld_row64(M, ((size_t)v * W + q) * 512u + lane8);
    FwAcc<W, HC> x = fw_acc_init<W, HC>();
    bool have = true;
    if (e1 - e0 > GIANT_DEG && g.n_giant != 0u && gp->giant_part != nullptr) {
      // giant row: the accumulators k_fw_giant_part stored for THIS sweep, merged in row order (see fused_row_giant)
      const uint32_t gi = giant_index(g, v);
      have = gp->giant_part[(size_t)batch * g.n_giant + gi] == (uint32_t)sweep + 1u;
      if (have) {
        const uint32_t n_ws = g.giant_slice0[g.n_giant] * 4u;
        const char *p = (const char *)(gp->giant_part + giant_tag_words(gridDim.y, g.n_giant)) +
                        ((size_t)batch * n_ws + (size_t)g.giant_slice0[gi] * 4u) * fw_part_bytes<W>();
        const uint32_t cntp = (e1 - e0 + 63u) / 64u;
        for (uint32_t k = 0; k < cntp; ++k, p += fw_part_bytes<W>()) fw_merge<W, HC>(x, fw_part_load<W, HC>(p, lane));
      }
    } else {
      for (uint32_t eb = e0; eb < e1; eb += 64) {
        const uint32_t cnt = min(64u, e1 - eb);
        const uint32_t sv = lane < cnt ? in_src[eb + lane] : v;
        const uint32_t wv = lane < cnt ? in_w[eb + lane] : INF;
        fw_chunk<W, MAXINF, HC, LEAF>(x, gp, D, H, M, v, v_router, eb, cnt, sv, wv, lane, my_root, root_slot, net_nexthops, ignore_ovl);
      }
    }
    if (!have) continue;                       // stamped after k_fw_giant_part had looked: due again in the next sweep
    uint32_t bd = x.bd, bh = x.bh;
    const uint32_t bd_all = x.bd_all, zb = x.zb, zh = x.zh;
    uint64_t am[W];
#pragma unroll
    for (int q = 0; q < W; ++q) am[q] = x.am[q];
    sat = sat || x.sat;
    if (HC) {
      const bool late = zb < bd;
#pragma unroll
      for (int q = 0; q < (HC ? W : 1); ++q) am[HC ? q : 0] = late ? x.zm[q] : am[HC ? q : 0];
      bh = late ? zh : bh;
      bd = late ? zb : bd;
    }
    // finish (finish_row of the packed path, on a triple)
    uint32_t nd, nh;
    if (v == my_root) { nd = 0u; nh = 0u; }
    else if (bd == INF || bd > maxpath) { nd = INF; nh = 0u; }
    else { nd = bd; nh = min(bh + v_router, 0xFFFFu); }                 // u16 saturating_add
    // the only (or a shorter) way in is a zero-cost link from a higher-numbered source: dynamic pop order (finish_row) —
    // the distance is stored, hops / mask are placeholders that k_repair recomputes
    const bool dynv = v != my_root && bd_all != INF && bd_all <= maxpath && bd_all < bd;
    if (dynv) { nd = bd_all; nh = 0u; }
    dyn = dyn || dynv;
    const bool live = v != my_root && nd != INF && !dynv;
    bool ch = nd != od || nh != (oh & 0xFFFFu);
#pragma unroll
    for (int q = 0; q < W; ++q) { am[q] = live ? am[q] : 0ull; ch = ch || am[q] != om[q]; }
    if (ch) {
#pragma unroll
      for (int q = 0; q < W; ++q) M[((size_t)v * W + q) * 64 + lane] = am[q];
      H[(size_t)v * 64 + lane] = nh;
      D[(size_t)v * 64 + lane] = nd;
      any = true;
    }
    if (__ballot(ch) != 0ull) {                        // wake the out-neighbours up
      const uint32_t o0 = rdlane(po, i), o1 = rdlane(po, i + 1);
      for (uint32_t ob = o0 + lane; ob < o1; ob += 64) A[g.out_dst[ob]] = (uint32_t)sweep + 2u;
    }
  }
  if (__ballot(any) != 0ull && lane == 0) changed[sweep] = 1;
  uint32_t lf = 0;
  if (MAXINF && sat) lf |= LF_NEED_EXACT;
  if (dyn) lf |= LF_DYN;
  if (lf) atomicOr(&lane_flags[root_slot], lf);
}

// init for k_fw: the roots' own distance (dist = 0; hv / mask are zero already) and the first activation stamps.
// One wave per root: its lanes walk the out-links (a thread per root walked a fat-tree switch's hundred links one
// dependent load at a time: 31 us).
__global__ __launch_bounds__(256) void k_init_fw(GraphDev g, uint32_t *dist, uint32_t *act, const uint32_t *roots, uint32_t n_lanes) {
  const uint32_t i = (blockIdx.x * 256u + threadIdx.x) >> 6, ln = threadIdx.x & 63u;
  if (i >= n_lanes) return;
  const uint32_t r = roots[i];
  if (r == INF) return;
  const uint32_t n = g.n, batch = i >> 6, lane = i & 63;
  if (ln == 0) dist[((size_t)batch * n + r) * 64 + lane] = 0;
  for (uint32_t k = g.out_ptr[r] + ln; k < g.out_ptr[r + 1]; k += 64u) act[(size_t)batch * n + g.out_dst[k]] = 1u;
}

// ---------------------------------------------------------------------------------------------
// Exact sequential kernel: one lane = one root.  Literal restatement of the reference loop
//  with a binary heap keyed (distance, vertex) with decrease-key, which
// pops in exactly the ordered-map order.  State lives in the row-major OUTPUT arrays of that root
// (dist / hops / flags / mask) plus two u32 scratch rows (heap, pos) per root.
struct ExactArgs {
  GraphDev g;
  const uint32_t *root_list;   // [n_exact] indices into roots[]
  const uint32_t *roots;       // [n_roots]
  uint32_t n_exact;
  uint32_t maxpath, net_nexthops, ignore_ovl;
  SlotTabs tabs;
  uint32_t *dist; uint16_t *hops; uint16_t *flags; uint64_t *mask; uint32_t words;
  uint32_t *pop_rank;          // may be null
  const uint32_t *row_map;     // output row of root index ri (null: ri itself)
  uint32_t *heap; uint32_t *pos;   // [n_exact][n]
};

__device__ __forceinline__ bool ex_less(const uint32_t *dist, uint32_t a, uint32_t b) {
  const uint32_t da = dist[a], db = dist[b];
  return da < db || (da == db && a < b);
}

__global__ void k_exact(ExactArgs a) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n_exact) return;
  const uint32_t ri = a.root_list[t];
  const uint32_t root = a.roots[ri];
  const uint32_t n = a.g.n;
  const size_t orow = a.row_map ? a.row_map[ri] : ri;
  uint32_t *dist = a.dist + orow * n;
  uint16_t *hops = a.hops + orow * n;
  uint16_t *flags = a.flags + orow * n;
  uint64_t *mask = a.mask + orow * n * a.words;
  uint32_t *rank = a.pop_rank ? a.pop_rank + orow * n : nullptr;
  uint32_t *heap = a.heap + (size_t)t * n;
  uint32_t *pos = a.pos + (size_t)t * n;
  const uint32_t W = a.words;
  for (uint32_t i = 0; i < n; ++i) {
    dist[i] = INF; hops[i] = 0; flags[i] = 0; pos[i] = INF;
    if (rank) rank[i] = INF;
    for (uint32_t k = 0; k < W; ++k) mask[(size_t)i * W + k] = 0;
  }
  if (root == INF) return;
  uint32_t hn = 0, popped = 0;
  dist[root] = 0; heap[0] = root; pos[root] = 0; hn = 1;
  while (hn) {
    const uint32_t v = heap[0];
    // pop
    --hn;
    if (hn) {
      uint32_t x = heap[hn], i = 0;
      for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= hn) break;
        if (c + 1 < hn && ex_less(dist, heap[c + 1], heap[c])) ++c;
        if (!ex_less(dist, heap[c], x)) break;
        heap[i] = heap[c]; pos[heap[i]] = i; i = c;
      }
      heap[i] = x; pos[x] = i;
    }
    pos[v] = INF;
    flags[v] = 1 | 2;                                  // HSPF_RF_IN_SPT | HSPF_RF_EXACT
    if (rank) rank[v] = popped;
    ++popped;
    const uint32_t vf = a.g.vflags[v];
    const uint32_t vhops = hops[v];
    // NO_EXPAND sources have no kept out-links at all (dropped at upload)
    if (vhops != 0 && !(vf & 1u) && !a.ignore_ovl && (vf & 2u)) continue;
    const uint32_t dvv = dist[v];
    uint32_t vbase = 0xFFFFFFFFu;
    for (uint32_t k = a.g.out_ptr[v]; k < a.g.out_ptr[v + 1]; ++k) {
      const uint32_t tv = a.g.out_dst[k];
      if (flags[tv] & 1) continue;                      // already on the SPT
      const uint64_t c64 = (uint64_t)dvv + a.g.out_w[k];
      const uint32_t d = c64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c64;
      if (d > a.maxpath) continue;
      const bool on_cand = pos[tv] != INF;
      if (on_cand && d > dist[tv]) continue;
      const bool t_router = !(a.g.vflags[tv] & 1u);
      if (!on_cand || d < dist[tv]) {
        dist[tv] = d;
        uint32_t h = vhops + (t_router ? 1u : 0u);
        hops[tv] = (uint16_t)(h > 0xFFFFu ? 0xFFFFu : h);
        for (uint32_t q = 0; q < W; ++q) mask[(size_t)tv * W + q] = 0;
        uint32_t i;
        if (!on_cand) { i = hn++; } else { i = pos[tv]; }
        // sift up
        while (i > 0) {
          uint32_t p = (i - 1) / 2;
          const uint32_t hp = heap[p];
          // compare (d,tv) with heap[p]
          const uint32_t dp = dist[hp];
          if (!(d < dp || (d == dp && tv < hp))) break;
          heap[i] = hp; pos[hp] = i; i = p;
        }
        heap[i] = tv; pos[tv] = i;
      }
      if (vhops == 0) {
        if (t_router || a.net_nexthops) {
          if (vbase == 0xFFFFFFFFu) vbase = (v == root) ? 0u : slot_base_of(a.tabs, ri, v);
          const uint32_t s = vbase + a.g.out_fpos[k];
          mask[(size_t)tv * W + (s >> 6)] |= 1ull << (s & 63u);
        }
      } else {
        for (uint32_t q = 0; q < W; ++q) mask[(size_t)tv * W + q] |= mask[(size_t)v * W + q];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Route derivation (prefix attachment): one thread per (root, prefix).  Consecutive threads take
// consecutive prefixes of one root: the table reads and the result writes are coalesced, the
// dist / mask gathers stay inside one root's row-major slab (L2 resident: 400 KB + 800 KB per root
// at 100 k vertices).  HBM bound on the results it writes: (8 + 8W) bytes per (root, prefix).
__global__ __launch_bounds__(256) void k_routes(uint32_t n, uint32_t n_roots, uint32_t W, uint32_t n_pfx,
                                                const uint32_t *__restrict__ pfx_ptr,
                                                const uint32_t *__restrict__ pfx_vertex,
                                                const uint32_t *__restrict__ pfx_metric,
                                                const uint32_t *__restrict__ dist,
                                                const uint16_t *__restrict__ flags,
                                                const uint64_t *__restrict__ mask,
                                                uint32_t *__restrict__ best_metric,
                                                uint32_t *__restrict__ best_entry,
                                                uint64_t *__restrict__ nh_mask, uint32_t mode) {
  const bool sat = mode & HSPF_PFX_SATURATING, lastmin = mode & HSPF_PFX_LAST_MIN;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = blockIdx.y;
  if (p >= n_pfx) return;
  const uint32_t *D = dist + (size_t)r * n;
  const uint16_t *F = flags + (size_t)r * n;
  const uint64_t *M = mask + (size_t)r * n * W;
  uint32_t bm = INF, be = INF;
  const uint32_t a = pfx_ptr[p], b = pfx_ptr[p + 1];
  for (uint32_t e = a; e < b; ++e) {                     // entries are in ascending vertex order
    const uint32_t v = pfx_vertex[e];
    if (!(F[v] & 1u)) continue;                          // vertex not in this root's SPT
    uint32_t m = D[v] + pfx_metric[e];                   // `vertex.distance + network.metric`, plain add
    if (sat && m < D[v]) m = INF;                        // holo-ospf: saturating_add (route.rs:362-366)
    if (be == INF || m < bm || (lastmin && m == bm)) { bm = m; be = e; }
  }
  const size_t o = (size_t)r * n_pfx + p;
  best_metric[o] = bm;
  best_entry[o] = be;
  for (uint32_t w = 0; w < W; ++w) {
    uint64_t acc = 0;
    if (be != INF && lastmin) acc = M[(size_t)pfx_vertex[be] * W + w];     // the later entry replaced the route
    else if (be != INF)
      for (uint32_t e = be; e < b; ++e) {                // entries before `be` have a larger metric
        const uint32_t v = pfx_vertex[e];
        if (!(F[v] & 1u)) continue;
        uint32_t m = D[v] + pfx_metric[e];
        if (sat && m < D[v]) m = INF;
        if (m == bm) acc |= M[(size_t)v * W + w];
      }
    nh_mask[o * W + w] = acc;
  }
}


// HSPF_PFX_ORDERED: the order-dependent fold of update_rib_intra_area (holo-ospf/src/route.rs:343-448) over entries in
// the reference's iteration order (OSPFv3: Intra-Area-Prefix-LSAs in LSDB order, ospfv3/spf.rs:421-478).  One thread per
// (root, prefix) walks its entries once per mask word: the decisions do not depend on the masks, so word w > 0 simply
// replays them.  State = the route so far: exists, metric, origin, owner entry (HSPF_PFX_KEPT_INIT: the route another
// area left), union of the merged masks.
__global__ __launch_bounds__(256) void k_routes_ordered(uint32_t n, uint32_t n_roots, uint32_t W, uint32_t n_pfx,
                                                        const uint32_t *__restrict__ pfx_ptr,
                                                        const uint32_t *__restrict__ pfx_vertex,
                                                        const uint32_t *__restrict__ pfx_metric,
                                                        const uint32_t *__restrict__ pfx_origin,
                                                        const uint8_t *__restrict__ init_exists,
                                                        const uint32_t *__restrict__ init_metric,
                                                        const uint32_t *__restrict__ init_origin,
                                                        const uint32_t *__restrict__ dist,
                                                        const uint16_t *__restrict__ flags,
                                                        const uint64_t *__restrict__ mask,
                                                        uint32_t *__restrict__ best_metric,
                                                        uint32_t *__restrict__ best_entry,
                                                        uint64_t *__restrict__ nh_mask) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = blockIdx.y;
  if (p >= n_pfx) return;
  const uint32_t *D = dist + (size_t)r * n;
  const uint16_t *F = flags + (size_t)r * n;
  const uint64_t *M = mask + (size_t)r * n * W;
  const uint32_t a = pfx_ptr[p], b = pfx_ptr[p + 1];
  const bool ex0 = init_exists && init_exists[p] != 0;
  const uint32_t bm0 = ex0 ? init_metric[p] : INF, bo0 = ex0 ? init_origin[p] : 0u;
  const size_t o = (size_t)r * n_pfx + p;
  for (uint32_t w = 0; w < W; ++w) {
    bool exists = ex0;
    uint32_t bm = bm0, bo = bo0, be = ex0 ? HSPF_PFX_KEPT_INIT : INF;
    uint64_t acc = 0;
    for (uint32_t e = a; e < b; ++e) {
      const uint32_t pv = pfx_vertex[e], v = pv & 0x7FFFFFFFu;
      if (!(F[v] & 1u)) continue;                          // `area.state.spt.get(&vid)`: vertex not in this root's SPT
      const uint32_t m = add_sat(D[v], pfx_metric[e]);     // stub.vertex.distance.saturating_add(stub.metric), :362-366
      if (exists && m > bm) continue;                      // :371-375
      if ((pv & HSPF_PFX_ENTRY_NETWORK) && exists) {       // :388-400
        if (m < bm || (m == bm && pfx_origin[e] > bo)) exists = false;
        else continue;
      }
      if (!exists || m < bm) { exists = true; bm = m; bo = pfx_origin[e]; be = e; acc = M[(size_t)v * W + w]; }
      else acc |= M[(size_t)v * W + w];                    // m == bm: merge (route_update, :918-965)
    }
    if (w == 0) { best_metric[o] = exists ? bm : INF; best_entry[o] = be; }
    nh_mask[o * W + w] = acc;
  }
}

// Several areas, ONE RIB (include/holo_spf_hip.h hspf_rib_fold_device): k_routes_ordered's literal fold of one area's table
// with the instance-wide state as initial state, written back in place.  One thread per AREA prefix p -> instance prefix
// map[p] (each at most once: no two threads touch one state row).  The selection is recomputed per mask word, like
// k_routes_ordered; the state's own words of OTHER areas are cleared when this area's entry takes the route over.
__global__ __launch_bounds__(256) void k_rib_fold(uint32_t n, uint32_t W, uint32_t n_pfx, const uint32_t *__restrict__ pfx_ptr,
                                                  const uint32_t *__restrict__ pfx_vertex, const uint32_t *__restrict__ pfx_metric,
                                                  const uint32_t *__restrict__ pfx_origin, const uint32_t *__restrict__ map,
                                                  const uint32_t *__restrict__ dist, const uint16_t *__restrict__ flags,
                                                  const uint64_t *__restrict__ mask, uint32_t area_tag, uint32_t word_offset, uint32_t rib_words,
                                                  uint32_t *__restrict__ rib_metric, uint32_t *__restrict__ rib_entry,
                                                  uint64_t *__restrict__ rib_mask, uint32_t *__restrict__ rib_origin) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pfx) return;
  const uint32_t ip = map[p];
  const uint32_t a = pfx_ptr[p], b = pfx_ptr[p + 1];
  const bool ex0 = rib_entry[ip] != INF;
  const uint32_t bm0 = ex0 ? rib_metric[ip] : INF, bo0 = ex0 ? rib_origin[ip] : 0u;
  bool exists = ex0;
  uint32_t bm = bm0, bo = bo0, be = ex0 ? HSPF_PFX_KEPT_INIT : INF;
  for (uint32_t e = a; e < b; ++e) {                       // who owns the route after this area (k_routes_ordered, word-independent part)
    const uint32_t pv = pfx_vertex[e], v = pv & 0x7FFFFFFFu;
    if (!(flags[v] & 1u)) continue;
    const uint32_t m = add_sat(dist[v], pfx_metric[e]);
    if (exists && m > bm) continue;
    if ((pv & HSPF_PFX_ENTRY_NETWORK) && exists) {
      if (m < bm || (m == bm && pfx_origin[e] > bo)) exists = false;
      else continue;
    }
    if (!exists || m < bm) { exists = true; bm = m; bo = pfx_origin[e]; be = e; }
  }
  if (be == INF) return;                                   // no entry of this area in the SPT, nothing held before
  uint64_t *R = rib_mask + (size_t)ip * rib_words;
  if (be != HSPF_PFX_KEPT_INIT)
    for (uint32_t w = 0; w < rib_words; ++w) R[w] = 0ull;  // the route changes hands: the earlier areas' next hops go
  for (uint32_t w = 0; w < W; ++w) {                       // this area's next hops: the entries merged since the last take-over
    bool ex = ex0;
    uint32_t m0 = bm0, o0 = bo0;
    uint64_t acc = 0;
    for (uint32_t e = a; e < b; ++e) {
      const uint32_t pv = pfx_vertex[e], v = pv & 0x7FFFFFFFu;
      if (!(flags[v] & 1u)) continue;
      const uint32_t m = add_sat(dist[v], pfx_metric[e]);
      if (ex && m > m0) continue;
      if ((pv & HSPF_PFX_ENTRY_NETWORK) && ex) {
        if (m < m0 || (m == m0 && pfx_origin[e] > o0)) ex = false;
        else continue;
      }
      if (!ex || m < m0) { ex = true; m0 = m; o0 = pfx_origin[e]; acc = mask[(size_t)v * W + w]; }
      else acc |= mask[(size_t)v * W + w];
    }
    if (word_offset + w < rib_words) R[word_offset + w] |= acc;
  }
  if (be != HSPF_PFX_KEPT_INIT) { rib_metric[ip] = bm; rib_origin[ip] = bo; rib_entry[ip] = area_tag | be; }
}

__global__ __launch_bounds__(256) void k_rib_clear(uint32_t n_pfx, uint32_t rib_words, uint32_t *__restrict__ rib_metric, uint32_t *__restrict__ rib_entry,
                                                   uint64_t *__restrict__ rib_mask, uint32_t *__restrict__ rib_origin) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pfx) return;
  rib_metric[p] = INF; rib_entry[p] = INF; rib_origin[p] = 0u;
  for (uint32_t w = 0; w < rib_words; ++w) rib_mask[(size_t)p * rib_words + w] = 0ull;
}

// RIB diff: update_global_rib's comparison (holo-isis/src/route.rs:254-312) on two result sets of
// k_routes.  One thread per (root, prefix); HBM bound: 2 x (8 + 8W) bytes read, 1 + 1 written per pair.
__global__ __launch_bounds__(256) void k_routes_diff(size_t count, uint32_t W,
                                                     const uint32_t *__restrict__ om, const uint32_t *__restrict__ oe,
                                                     const uint64_t *__restrict__ on,
                                                     const uint32_t *__restrict__ nm, const uint32_t *__restrict__ ne,
                                                     const uint64_t *__restrict__ nn,
                                                     uint8_t *__restrict__ action, uint8_t *__restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const bool had = oe[i] != INF, has = ne[i] != INF;
  bool same_nh = true, old_nh = false, new_nh = false;
  for (uint32_t w = 0; w < W; ++w) {
    const uint64_t a = on[i * W + w], b = nn[i * W + w];
    same_nh = same_nh && a == b;
    old_nh = old_nh || a != 0ull;
    new_nh = new_nh || b != 0ull;
  }
  uint32_t act;
  if (has) act = (had && om[i] == nm[i] && same_nh) ? HSPF_DIFF_SAME : (new_nh ? HSPF_DIFF_INSTALL : HSPF_DIFF_SILENT);
  else act = had ? (old_nh ? HSPF_DIFF_WITHDRAW : HSPF_DIFF_SILENT) : HSPF_DIFF_SAME;
  action[i] = (uint8_t)act;
  flag[i] = (act == HSPF_DIFF_INSTALL || act == HSPF_DIFF_WITHDRAW) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_routes_diff_scatter(size_t count, uint32_t n_pfx, uint32_t n_roots,
                                                             const uint8_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                                             uint32_t *__restrict__ changed, uint32_t *__restrict__ changed_ptr) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (i <= n_roots) changed_ptr[i] = pos[i * (size_t)n_pfx];       // pos[count] = total
  if (i >= count) return;
  if (flag[i]) changed[pos[i]] = (uint32_t)(i % n_pfx);
}

// The hand-off of the diff: record k of the stream = the k-th changed (root, prefix) pair in emission
// order (roots ascending, prefixes ascending inside a root — the order of `changed`).  ROUTE_REC_WORDS + 2 W u32 words:
// [root, prefix, action, metric, entry, 0 | W mask words (lo, hi)] — everything route_install / route_uninstall put on the
// ibus per route (holo-isis/src/ibus/tx.rs:35-110, holo-ospf/src/ibus/tx.rs:32-77) short of the addresses, which the host
// resolves per first-hop slot.  The owning root of position k: the last r with changed_ptr[r] <= k.
constexpr uint32_t ROUTE_REC_WORDS = 6u;
__global__ __launch_bounds__(256) void k_routes_pack(uint32_t n_records, uint32_t n_roots, uint32_t n_pfx, uint32_t W,
                                                     const uint32_t *__restrict__ changed, const uint32_t *__restrict__ changed_ptr,
                                                     const uint8_t *__restrict__ action, const uint32_t *__restrict__ best_metric,
                                                     const uint32_t *__restrict__ best_entry, const uint64_t *__restrict__ nexthop_mask,
                                                     uint32_t *__restrict__ rec) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n_records) return;
  uint32_t lo = 0, hi = n_roots;                       // changed_ptr[lo] <= k < changed_ptr[hi]
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (changed_ptr[mid] <= k) lo = mid; else hi = mid; }
  const uint32_t p = changed[k];
  uint32_t *o = rec + (size_t)k * (ROUTE_REC_WORDS + 2u * W);
  if (p >= n_pfx) {                                     // (a list the caller did not get from hspf_routes_diff_device: nothing is read through it)
    o[0] = lo; o[1] = p; o[2] = 0u; o[3] = 0u; o[4] = 0u; o[5] = 1u;
    for (uint32_t w = 0; w < 2u * W; ++w) o[ROUTE_REC_WORDS + w] = 0u;
    return;
  }
  const size_t i = (size_t)lo * n_pfx + p;
  o[0] = lo; o[1] = p; o[2] = action[i]; o[3] = best_metric[i]; o[4] = best_entry[i]; o[5] = 0u;
  for (uint32_t w = 0; w < W; ++w) {
    const uint64_t m = nexthop_mask[i * W + w];
    o[ROUTE_REC_WORDS + 2u * w] = (uint32_t)m; o[ROUTE_REC_WORDS + 2u * w + 1u] = (uint32_t)(m >> 32);
  }
}

// ---------------------------------------------------------------------------------------------
// Ancestor sets over the SPT's parent DAG: what holo-isis answers with a stack DFS over
// `Vertex.parents` (Spt::is_on_path, holo-isis/src/spf.rs:261-286) for flooding::manet::reflood_list
// (flooding/manet.rs:99-173: "second hops that are not on a shortest path to the LSP originator", "two-hop nodes behind
// this remote neighbour").  For every root of a finished run and a level L: the root's level-L routers (router
// vertices of its SPT with hops == L: L = 1 the first hops / remote-neighbour list, L = 2 the second hops) are
// numbered in ascending vertex index, and anc[root][v] = bit set of the level-L routers that are ancestors of v in the
// parent DAG, or v itself.  is_on_path(a, d) for a level-L router a is then one bit test in anc[root][d].
//
// A link u -> v is a parent link (spf.rs:637-706: every relaxation that reached v at its final distance while v was
// still a candidate) iff both are in the SPT, dist[u] (+) w == dist[v], u may be expanded (the in-CSR holds only links
// of expandable sources; an overloaded source other than the root is excluded unless the run ignores the overload bit)
// and u was popped before v: the static (distance, index) order, i.e. w > 0 or u < v.  Hop-count-like graphs: a network
// whose way in is a zero-cost link from higher-numbered routers has exactly ONE parent, the first tight one in row order
// (= lowest-numbered router; see fused_row_any).  Roots that needed the sequential kernel (dynamic pop order) are not
// handled here: their level count is reported as 0xFFFFFFFF and the caller keeps its own walk.
// Row-major [root][vertex] like the run's outputs; one thread per (root, vertex), sweeps to the fixed point of a
// monotone OR (depth of the SPT below level L).
__device__ __forceinline__ bool anc_is_level(const uint16_t *F, const uint16_t *H, const uint8_t *vflags, uint32_t v, uint32_t level) {
  return (F[v] & 1u) && !(vflags[v] & 1u) && H[v] == level;
}

__global__ __launch_bounds__(256) void k_anc_count(GraphDev g, const uint32_t *__restrict__ roots, const uint16_t *__restrict__ hops,
                                                   const uint16_t *__restrict__ flags, uint32_t level, uint32_t *__restrict__ blockcnt) {
  __shared__ uint32_t s_cnt;
  const uint32_t r = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const bool is = v < g.n && roots[r] != INF && anc_is_level(flags + (size_t)r * g.n, hops + (size_t)r * g.n, g.vflags, v, level);
  const uint64_t b = __ballot(is);
  if ((threadIdx.x & 63u) == 0 && b) atomicAdd(&s_cnt, (uint32_t)__popcll(b));
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[(size_t)r * gridDim.x + blockIdx.x] = s_cnt;
}

__global__ void k_anc_scan(uint32_t n_roots, uint32_t n_blocks, const uint32_t *__restrict__ roots, const uint16_t *__restrict__ flags,
                           uint32_t n, uint32_t *__restrict__ blockcnt, uint32_t *__restrict__ level_count) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_roots) return;
  uint32_t acc = 0;
  for (uint32_t b = 0; b < n_blocks; ++b) { const uint32_t c = blockcnt[(size_t)r * n_blocks + b]; blockcnt[(size_t)r * n_blocks + b] = acc; acc += c; }
  const uint32_t root = roots[r];
  const bool exact = root != INF && (flags[(size_t)r * n + root] & 2u);       // HSPF_RF_EXACT: dynamic pop order
  level_count[r] = exact ? INF : acc;
}

__global__ __launch_bounds__(256) void k_anc_init(GraphDev g, const uint32_t *__restrict__ roots, const uint16_t *__restrict__ hops,
                                                  const uint16_t *__restrict__ flags, uint32_t level, uint32_t W,
                                                  const uint32_t *__restrict__ blockoff, uint32_t *__restrict__ level_rank,
                                                  uint64_t *__restrict__ anc) {
  __shared__ uint32_t s_w[4];
  const uint32_t r = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n = g.n;
  const bool is = v < n && roots[r] != INF && anc_is_level(flags + (size_t)r * n, hops + (size_t)r * n, g.vflags, v, level);
  const uint64_t b = __ballot(is);
  if (lane == 0) s_w[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  uint32_t base = blockoff[(size_t)r * gridDim.x + blockIdx.x];
  for (uint32_t w = 0; w < wave; ++w) base += s_w[w];
  const uint32_t rank = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
  if (v >= n) return;
  const size_t o = (size_t)r * n + v;
  if (level_rank) level_rank[o] = is ? rank : INF;
  for (uint32_t q = 0; q < W; ++q) anc[o * W + q] = (is && (rank >> 6) == q) ? (1ull << (rank & 63u)) : 0ull;
}

__global__ __launch_bounds__(256) void k_anc_sweep(GraphDev g, const uint32_t *__restrict__ roots, const uint32_t *__restrict__ dist,
                                                   const uint16_t *__restrict__ hops, const uint16_t *__restrict__ flags,
                                                   uint32_t level, uint32_t W, uint32_t ignore_ovl, uint32_t hc,
                                                   uint64_t *__restrict__ anc, int *changed, int sweep) {
  if (sweep > 0 && changed[sweep - 1] == 0) return;
  const uint32_t r = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
  const uint32_t n = g.n;
  const uint32_t root = roots[r];
  if (v >= n || root == INF) return;
  const uint32_t *D = dist + (size_t)r * n;
  const uint16_t *H = hops + (size_t)r * n, *F = flags + (size_t)r * n;
  if (!(F[v] & 1u) || v == root || H[v] < level) return;
  if (F[root] & 2u) return;                                         // exact-kernel root: not handled (level_count says so)
  uint64_t *A = anc + (size_t)r * n * W;
  const uint32_t dv = D[v];
  const bool v_net = (g.vflags[v] & 1u) != 0;
  bool zdone = false, any = false;
  for (uint32_t e = g.in_ptr[v]; e < g.in_ptr[v + 1]; ++e) {
    const uint32_t sw = g.in_src[e], u = sw & SRC_MASK, w = g.in_w[e];
    if (!(F[u] & 1u)) continue;
    if (!ignore_ovl && (sw & SRC_NO_TRANSIT) && u != root) continue;
    const uint32_t du = D[u];
    const uint64_t c = (uint64_t)du + w;
    if ((c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c) != dv) continue;       // not tight
    if (w == 0u && u >= v) {                                                    // zero-cost link from a higher-numbered source
      if (!(hc && v_net) || zdone) continue;                                    // static order: never a parent; hop-count-like: the first one only
      zdone = true;
    }
    if (H[u] < level) continue;                                                 // nothing to inherit above the level
    for (uint32_t q = 0; q < W; ++q) {
      const uint64_t pu = A[(size_t)u * W + q], pv = A[(size_t)v * W + q];
      if (pu & ~pv) { A[(size_t)v * W + q] = pv | pu; any = true; }
    }
  }
  if (any) changed[sweep] = 1;
}

}  // namespace hspf
