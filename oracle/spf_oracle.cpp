// oracle/spf_oracle.cpp — CPU restatement of holo's SPF hot loop.  TEST INFRASTRUCTURE ONLY.
//
// This file is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  Nothing under holo_amd/ links or imports it.
//
// What it restates (reference @ /root/reference, v0.9.0):
//   holo-ospf/src/spf.rs:611-721   run_area main loop        (candidate list, relax, ECMP)
//   holo-isis/src/spf.rs:543-706   compute_spt main loop     (same + gates + max path metric)
//   SURVEY.md Appendix A is the merged pseudo-code of the two; this file follows it line by
//   line on the CSR form of the graph defined in include/holo_spf_hip.h (row u = the links
//   `vertex_lsa_links` / `vertex_edges` yield for u, in that order).
//
// Parity pin: tests/test_oracle_golden.py drives this oracle (through the protocol-level
// Python restatements in oracle/isis_ref.py / oracle/ospf_ref.py) over vectors extracted from
// the reference's own conformance fixtures (tests/golden/, made by tools/make_golden.py) and
// checks the resulting RIB against the reference's golden `local-rib`.
//
// Three variants of the same algorithm, selected by `variant`:
//   0  ORACLE_REF    reference-shaped: ordered map keyed (distance, vertex), candidate lookup by
//                    LINEAR SCAN over the map (holo-ospf/src/spf.rs:682-685,
//                    holo-isis/src/spf.rs:656-659) and the two-way check re-walking the
//                    neighbour's links on every edge (:654-664 / :616-627).  O(E*|cand|).
//                    This is the "reference CPU path" stand-in for timing.
//   1  ORACLE_MAP    same ordered map, but candidate lookup through an index and the two-way
//                    check precomputed once.  Same results by construction (vertex ids are
//                    unique in the candidate list); used as the oracle on large graphs.
//   2  ORACLE_HEAP   binary heap with decrease-key on (distance, vertex): the "reasonable CPU
//                    implementation" baseline.  Must equal variants 0/1 bit for bit (tested).
//
// Build: make -C oracle   (g++ -O2 -shared -fPIC)  ->  oracle/liboracle_spf.so

#include <cstdint>
#include <cstring>
#include <map>
#include <atomic>
#include <thread>
#include <vector>
#include <algorithm>
#include <utility>

namespace {

constexpr uint32_t INF = 0xFFFFFFFFu;

constexpr uint8_t VF_NETWORK = 0x01, VF_NO_TRANSIT = 0x02, VF_NO_EXPAND = 0x04;
constexpr uint32_t RUN_NET_NEXTHOPS = 0x01, RUN_IGNORE_OVERLOAD = 0x02;

struct Graph {
  uint32_t n, e;
  const uint32_t *row_ptr, *col, *metric;
  const uint8_t *vflags;
  uint32_t max_path_metric;
};

// holo-ospf/src/spf.rs:38-46 / holo-isis/src/spf.rs:78-88.  `nexthops` holds the global
// edge index of the hops==0 relaxation that created each next hop (IS-IS: a Vec, duplicates
// kept, holo-isis/src/spf.rs:700-703; OSPF: a set, we dedupe on output).
struct Vertex {
  uint32_t id, distance;
  uint16_t hops;
  std::vector<uint32_t> parents;   // pop indices of parents, in relaxation order (IS-IS :677)
  std::vector<uint32_t> nexthops;
};

inline uint32_t sat_add(uint32_t a, uint32_t b) {      // u32::saturating_add
  uint64_t s = (uint64_t)a + b;
  return s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s;
}
inline uint16_t sat_inc16(uint16_t h) { return h == 0xFFFF ? h : (uint16_t)(h + 1); }

// "Check if the LSPs are mutually linked": does `t` list a link back to `v`?  Cost is not
// compared (holo-isis/src/spf.rs:616-627).
inline bool links_back(const Graph &g, uint32_t t, uint32_t v) {
  for (uint32_t k = g.row_ptr[t]; k < g.row_ptr[t + 1]; ++k)
    if (g.col[k] == v) return true;
  return false;
}

struct Out {
  uint32_t *dist; uint16_t *hops; uint16_t *flags; uint32_t *pop_rank;
  uint64_t *mask; uint32_t mask_words;
  uint32_t *n_nexthops;    // Vec length incl. duplicates (IS-IS), optional
  uint32_t *n_parents;     // parents.len(), optional
  uint64_t *work;          // [0] = candidate-scan steps + two-way steps (variant 0 shape)
};

// First-hop slot numbering of include/holo_spf_hip.h, restated independently: H = [root] ++
// BFS over network vertices reachable through network vertices only; slot(p,j)=base(p)+j.
struct Slots {
  std::vector<uint32_t> base;     // per vertex, INF if not in H
  uint32_t total = 0;
  void build(const Graph &g, uint32_t root, const std::vector<uint8_t> &twoway) {
    base.assign(g.n, INF);
    std::vector<uint32_t> q{root};
    base[root] = 0; total = g.row_ptr[root + 1] - g.row_ptr[root];
    for (size_t qi = 0; qi < q.size(); ++qi) {
      uint32_t p = q[qi];
      for (uint32_t k = g.row_ptr[p]; k < g.row_ptr[p + 1]; ++k) {
        uint32_t t = g.col[k];
        if (!twoway[k] || !(g.vflags[t] & VF_NETWORK) || base[t] != INF) continue;
        base[t] = total; total += g.row_ptr[t + 1] - g.row_ptr[t];
        q.push_back(t);
      }
    }
  }
};

// ---- variants 0 and 1: ordered-map candidate list ---------------------------------------
int run_map(const Graph &g, uint32_t root, uint32_t run_flags, bool ref_shape, Out &o,
            const std::vector<uint8_t> &twoway) {
  using Key = std::pair<uint32_t, uint32_t>;            // (distance, VertexId rank)
  std::map<Key, Vertex> cand;                           // BTreeMap<(u32, VertexId), Vertex>
  std::vector<Vertex> spt;                              // in pop order
  std::vector<uint32_t> spt_idx(g.n, INF);              // id -> pop index ("spt.contains")
  std::vector<uint32_t> cand_dist;                      // id -> key.first while on cand list
  std::vector<uint8_t> on_cand;                         // id -> on the cand list (a saturated candidate's distance IS
  if (!ref_shape) { cand_dist.assign(g.n, INF); on_cand.assign(g.n, 0); }   // 0xFFFFFFFF: no sentinel value is free)
  uint64_t work = 0;

  cand.emplace(Key{0, root}, Vertex{root, 0, 0, {}, {}});
  if (!ref_shape) { cand_dist[root] = 0; on_cand[root] = 1; }

  while (!cand.empty()) {
    // pop_first
    auto first = cand.begin();
    Vertex cv = std::move(first->second);
    cand.erase(first);
    if (!ref_shape) on_cand[cv.id] = 0;
    const uint32_t vidx = (uint32_t)spt.size();
    spt_idx[cv.id] = vidx;
    spt.push_back(std::move(cv));
    // NB: spt may reallocate below only through push_back at the top of the loop, so a
    // reference taken here stays valid for the body.
    const uint32_t v = spt[vidx].id, vdist = spt[vidx].distance;
    const uint16_t vhops = spt[vidx].hops;
    const uint8_t vf = g.vflags[v];

    // gates: holo-isis/src/spf.rs:557-604
    if (vf & VF_NO_EXPAND) continue;
    if (vhops != 0 && !(vf & VF_NETWORK) && !(run_flags & RUN_IGNORE_OVERLOAD) &&
        (vf & VF_NO_TRANSIT))
      continue;

    for (uint32_t k = g.row_ptr[v]; k < g.row_ptr[v + 1]; ++k) {
      const uint32_t t = g.col[k], cost = g.metric[k];
      // two-way check
      if (ref_shape) {
        bool back = false;
        for (uint32_t kk = g.row_ptr[t]; kk < g.row_ptr[t + 1]; ++kk) {
          ++work;
          if (g.col[kk] == v) { back = true; break; }
        }
        if (!back) continue;
      } else if (!twoway[k]) continue;
      // already on the SPT?
      if (spt_idx[t] != INF) continue;
      const uint32_t distance = sat_add(vdist, cost);
      if (distance > g.max_path_metric) continue;        // holo-isis/src/spf.rs:637-647
      uint16_t hops = vhops;
      if (!(g.vflags[t] & VF_NETWORK)) hops = sat_inc16(hops);

      // "Check if this vertex is already present on the candidate list."
      if (ref_shape) {
        auto it = cand.begin();
        for (; it != cand.end(); ++it) { ++work; if (it->second.id == t) break; }
        if (it != cand.end()) {
          if (distance < it->second.distance) cand.erase(it);
          else if (distance > it->second.distance) continue;
        }
      } else if (on_cand[t]) {
        if (distance < cand_dist[t]) cand.erase(Key{cand_dist[t], t});
        else if (distance > cand_dist[t]) continue;
      }
      auto ins = cand.emplace(Key{distance, t}, Vertex{t, distance, hops, {}, {}});
      Vertex &c = ins.first->second;                     // or_insert_with: keeps old hops on Equal
      if (!ref_shape) { cand_dist[t] = distance; on_cand[t] = 1; }
      c.parents.push_back(vidx);
      const Vertex &pv = spt[vidx];
      if (vhops == 0) {
        if (!(g.vflags[t] & VF_NETWORK) || (run_flags & RUN_NET_NEXTHOPS)) c.nexthops.push_back(k);
      } else {
        c.nexthops.insert(c.nexthops.end(), pv.nexthops.begin(), pv.nexthops.end());
      }
    }
  }

  // ---- outputs
  Slots slots; slots.build(g, root, twoway);
  if ((slots.total + 63) / 64 > o.mask_words && o.mask) return -5;
  for (uint32_t i = 0; i < g.n; ++i) {
    o.dist[i] = INF;
    if (o.hops) o.hops[i] = 0;
    if (o.flags) o.flags[i] = 0;
    if (o.pop_rank) o.pop_rank[i] = INF;
    if (o.n_nexthops) o.n_nexthops[i] = 0;
    if (o.n_parents) o.n_parents[i] = 0;
  }
  if (o.mask) std::memset(o.mask, 0, sizeof(uint64_t) * (size_t)g.n * o.mask_words);
  // edge index -> owning row (only for edges of H vertices; find by binary search on row_ptr)
  for (uint32_t r = 0; r < spt.size(); ++r) {
    const Vertex &x = spt[r];
    o.dist[x.id] = x.distance;
    if (o.hops) o.hops[x.id] = x.hops;
    if (o.flags) o.flags[x.id] = 1;
    if (o.pop_rank) o.pop_rank[x.id] = r;
    if (o.n_nexthops) o.n_nexthops[x.id] = (uint32_t)x.nexthops.size();
    if (o.n_parents) o.n_parents[x.id] = (uint32_t)x.parents.size();
    if (o.mask)
      for (uint32_t k : x.nexthops) {
        uint32_t p = (uint32_t)(std::upper_bound(g.row_ptr, g.row_ptr + g.n + 1, k) - g.row_ptr) - 1;
        uint32_t s = slots.base[p] + (k - g.row_ptr[p]);
        o.mask[(size_t)x.id * o.mask_words + s / 64] |= 1ull << (s % 64);
      }
  }
  if (o.work) o.work[0] = work;
  return 0;
}

// ---- variant 2: binary heap with decrease-key ----------------------------------------------
// Same semantics; the heap orders (distance, id) exactly like the ordered map, so the pop
// sequence is identical.  Next hops are kept as mask words directly (set semantics).
int run_heap(const Graph &g, uint32_t root, uint32_t run_flags, Out &o,
             const std::vector<uint8_t> &twoway) {
  const uint32_t W = o.mask ? o.mask_words : 0;
  Slots slots; slots.build(g, root, twoway);
  if (o.mask && (slots.total + 63) / 64 > W) return -5;
  std::vector<uint32_t> heap; heap.reserve(1024);
  std::vector<uint32_t> pos(g.n, INF);                   // INF: not on cand list
  std::vector<uint8_t> in_spt(g.n, 0);
  for (uint32_t i = 0; i < g.n; ++i) {
    o.dist[i] = INF;
    if (o.hops) o.hops[i] = 0;
    if (o.flags) o.flags[i] = 0;
    if (o.pop_rank) o.pop_rank[i] = INF;
    if (o.n_nexthops) o.n_nexthops[i] = 0;
    if (o.n_parents) o.n_parents[i] = 0;
  }
  if (o.mask) std::memset(o.mask, 0, sizeof(uint64_t) * (size_t)g.n * W);
  std::vector<uint16_t> hops_local;
  uint16_t *hops = o.hops;
  if (!hops) { hops_local.assign(g.n, 0); hops = hops_local.data(); }
  uint32_t *dist = o.dist;
  auto less = [&](uint32_t a, uint32_t b) {
    return dist[a] < dist[b] || (dist[a] == dist[b] && a < b);
  };
  auto sift_up = [&](uint32_t i) {
    uint32_t x = heap[i];
    while (i > 0) { uint32_t p = (i - 1) / 2; if (!less(x, heap[p])) break; heap[i] = heap[p]; pos[heap[i]] = i; i = p; }
    heap[i] = x; pos[x] = i;
  };
  auto sift_down = [&](uint32_t i) {
    uint32_t x = heap[i]; const uint32_t n = (uint32_t)heap.size();
    for (;;) {
      uint32_t c = 2 * i + 1; if (c >= n) break;
      if (c + 1 < n && less(heap[c + 1], heap[c])) ++c;
      if (!less(heap[c], x)) break;
      heap[i] = heap[c]; pos[heap[i]] = i; i = c;
    }
    heap[i] = x; pos[x] = i;
  };
  dist[root] = 0; hops[root] = 0; heap.push_back(root); pos[root] = 0;
  uint32_t rank = 0;
  while (!heap.empty()) {
    const uint32_t v = heap[0];
    heap[0] = heap.back(); heap.pop_back();
    if (!heap.empty()) sift_down(0);
    pos[v] = INF; in_spt[v] = 1;
    if (o.flags) o.flags[v] = 1;
    if (o.pop_rank) o.pop_rank[v] = rank;
    ++rank;
    const uint8_t vf = g.vflags[v];
    const uint16_t vhops = hops[v];
    if (vf & VF_NO_EXPAND) continue;
    if (vhops != 0 && !(vf & VF_NETWORK) && !(run_flags & RUN_IGNORE_OVERLOAD) && (vf & VF_NO_TRANSIT))
      continue;
    for (uint32_t k = g.row_ptr[v]; k < g.row_ptr[v + 1]; ++k) {
      if (!twoway[k]) continue;
      const uint32_t t = g.col[k];
      if (in_spt[t]) continue;
      const uint32_t distance = sat_add(dist[v], g.metric[k]);
      if (distance > g.max_path_metric) continue;
      const bool on_cand = pos[t] != INF;
      if (on_cand && distance > dist[t]) continue;
      if (!on_cand || distance < dist[t]) {
        // new candidate, or Less: old candidate (and its next hops / parents) is discarded
        dist[t] = distance;
        hops[t] = (g.vflags[t] & VF_NETWORK) ? vhops : sat_inc16(vhops);
        if (o.mask) std::memset(o.mask + (size_t)t * W, 0, sizeof(uint64_t) * W);
        if (o.n_nexthops) o.n_nexthops[t] = 0;
        if (o.n_parents) o.n_parents[t] = 0;
        if (!on_cand) { heap.push_back(t); pos[t] = (uint32_t)heap.size() - 1; }
        sift_up(pos[t]);
      }
      if (o.n_parents) o.n_parents[t] += 1;
      if (vhops == 0) {
        if (!(g.vflags[t] & VF_NETWORK) || (run_flags & RUN_NET_NEXTHOPS)) {
          if (o.mask) { uint32_t s = slots.base[v] + (k - g.row_ptr[v]); o.mask[(size_t)t * W + s / 64] |= 1ull << (s % 64); }
          if (o.n_nexthops) o.n_nexthops[t] += 1;
        }
      } else {
        if (o.mask) for (uint32_t w = 0; w < W; ++w) o.mask[(size_t)t * W + w] |= o.mask[(size_t)v * W + w];
        if (o.n_nexthops) o.n_nexthops[t] += o.n_nexthops[v];
      }
    }
  }
  // vertices that were candidates are all popped eventually; dist of never-reached stays INF
  return 0;
}

}  // namespace

extern "C" {

// Number of mask words the slot numbering needs for `root` (>=1).
uint32_t oracle_mask_words(uint32_t n, uint32_t e, const uint32_t *row_ptr, const uint32_t *col,
                           const uint32_t *metric, const uint8_t *vflags, uint32_t root) {
  Graph g{n, e, row_ptr, col, metric, vflags, INF};
  std::vector<uint8_t> twoway(e);
  for (uint32_t u = 0; u < n; ++u)
    for (uint32_t k = row_ptr[u]; k < row_ptr[u + 1]; ++k) twoway[k] = links_back(g, col[k], u);
  Slots s; s.build(g, root, twoway);
  uint32_t w = (s.total + 63) / 64;
  return w ? w : 1;
}

// Runs `n_roots` SPFs, one after the other (oracle_spf_run, n_threads = 1) or dealt to n_threads workers.  Outputs are
// row-major [n_roots][n]; any of hops/flags/pop_rank/mask/n_nexthops/n_parents/work may be NULL.
// Returns 0, or -1 on bad input, -5 if mask_words is too small.
int oracle_spf_run_mt(uint32_t n, uint32_t e, const uint32_t *row_ptr, const uint32_t *col,
                      const uint32_t *metric, const uint8_t *vflags, uint32_t max_path_metric,
                      const uint32_t *roots, uint32_t n_roots, uint32_t run_flags, int variant,
                      uint32_t *dist, uint16_t *hops, uint16_t *flags, uint32_t *pop_rank,
                      uint64_t *mask, uint32_t mask_words, uint32_t *n_nexthops,
                      uint32_t *n_parents, uint64_t *work, uint32_t n_threads) {
  if (!row_ptr || !dist || (e && (!col || !metric)) || !vflags) return -1;
  Graph g{n, e, row_ptr, col, metric, vflags, max_path_metric};
  std::vector<uint8_t> twoway(e);
  for (uint32_t u = 0; u < n; ++u)
    for (uint32_t k = row_ptr[u]; k < row_ptr[u + 1]; ++k) {
      if (col[k] >= n) return -1;
      twoway[k] = links_back(g, col[k], u);
    }
  // n_threads > 1: the roots are independent runs over the read-only graph; worker t takes roots t, t + T, t + 2T, ...
  // (the all-cores CPU baseline of bench.py and a faster checker for the full-size parity tests; one run is still the
  // same sequential loop)
  std::atomic<int> first_err{0};
  auto one = [&](uint32_t r) -> int {
    const size_t off = (size_t)r * n;
    Out o{dist + off, hops ? hops + off : nullptr, flags ? flags + off : nullptr,
          pop_rank ? pop_rank + off : nullptr, mask ? mask + off * mask_words : nullptr, mask_words,
          n_nexthops ? n_nexthops + off : nullptr, n_parents ? n_parents + off : nullptr,
          work ? work + r : nullptr};
    if (roots[r] == INF) {   // padding root: empty SPT
      for (uint32_t i = 0; i < n; ++i) {
        o.dist[i] = INF; if (o.hops) o.hops[i] = 0; if (o.flags) o.flags[i] = 0;
        if (o.pop_rank) o.pop_rank[i] = INF;
        if (o.n_nexthops) o.n_nexthops[i] = 0;
        if (o.n_parents) o.n_parents[i] = 0;
      }
      if (o.mask) std::memset(o.mask, 0, sizeof(uint64_t) * (size_t)n * mask_words);
      return 0;
    }
    if (roots[r] >= n) return -1;
    return variant == 2 ? run_heap(g, roots[r], run_flags, o, twoway)
                        : run_map(g, roots[r], run_flags, variant == 0, o, twoway);
  };
  const uint32_t T = std::max<uint32_t>(1, std::min<uint32_t>(n_threads, n_roots));
  if (T == 1) {
    for (uint32_t r = 0; r < n_roots; ++r) { const int rc = one(r); if (rc) return rc; }
    return 0;
  }
  std::vector<std::thread> pool;
  for (uint32_t t = 0; t < T; ++t)
    pool.emplace_back([&, t]() {
      for (uint32_t r = t; r < n_roots && first_err.load() == 0; r += T) {
        const int rc = one(r);
        if (rc) { int z = 0; first_err.compare_exchange_strong(z, rc); }
      }
    });
  for (auto &th : pool) th.join();
  return first_err.load();
}

int oracle_spf_run(uint32_t n, uint32_t e, const uint32_t *row_ptr, const uint32_t *col,
                   const uint32_t *metric, const uint8_t *vflags, uint32_t max_path_metric,
                   const uint32_t *roots, uint32_t n_roots, uint32_t run_flags, int variant,
                   uint32_t *dist, uint16_t *hops, uint16_t *flags, uint32_t *pop_rank,
                   uint64_t *mask, uint32_t mask_words, uint32_t *n_nexthops,
                   uint32_t *n_parents, uint64_t *work) {
  return oracle_spf_run_mt(n, e, row_ptr, col, metric, vflags, max_path_metric, roots, n_roots, run_flags, variant, dist, hops,
                           flags, pop_rank, mask, mask_words, n_nexthops, n_parents, work, 1);
}

}  // extern "C"
