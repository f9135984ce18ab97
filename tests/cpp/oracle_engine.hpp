// oracle_engine.hpp — CPU stand-in for the ENGINE ONLY (hspf::host::Engine on oracle/liboracle_spf.so, dlopen'ed).
// TEST INFRASTRUCTURE: used by tests/cpp/host_parity.cpp and tests/cpp/dropin_e2e.cpp (the CPU leg next to the product
// engine); nothing under holo_amd/ or include/ includes it.
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "holo_spf_host.hpp"

using namespace hspf::host;

// ---- CPU stand-in for the engine (tests only) ----------------------------------------------------------------------
typedef int (*oracle_run_t)(uint32_t, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, const uint8_t *, uint32_t,
                            const uint32_t *, uint32_t, uint32_t, int, uint32_t *, uint16_t *, uint16_t *, uint32_t *, uint64_t *,
                            uint32_t, uint32_t *, uint32_t *, uint64_t *);
struct OracleGraph : Graph {
  std::vector<uint32_t> row_ptr, col, metric;
  std::vector<uint8_t> vflags;
  uint32_t max_path;
};
class OracleEngine : public Engine {
 public:
  int variant = 1;                     // oracle variant: 0 reference-shaped, 1 ordered map, 2 binary heap (oracle/spf_oracle.cpp)
  explicit OracleEngine(const std::string &so) {
    void *h = dlopen(so.c_str(), RTLD_NOW);
    if (!h) throw std::runtime_error("dlopen " + so + " (run `make -C oracle`)");
    run_ = (oracle_run_t)dlsym(h, "oracle_spf_run");
    if (!run_) throw std::runtime_error("oracle_spf_run");
  }
  std::unique_ptr<Graph> upload(const std::vector<uint32_t> &rp, const std::vector<uint32_t> &c, const std::vector<uint32_t> &m,
                                const std::vector<uint8_t> &vf, uint32_t mp) override {
    auto g = std::make_unique<OracleGraph>();
    g->row_ptr = rp; g->col = c; g->metric = m; g->vflags = vf; g->max_path = mp;
    return g;
  }
  SlotTable slot_table(Graph &gr, uint32_t root) override {          // restatement of include/holo_spf_hip.h "first-hop slots"
    auto &g = static_cast<OracleGraph &>(gr);
    SlotTable st;
    st.vertex = {root}; st.base = {0};
    st.total = g.row_ptr[root + 1] - g.row_ptr[root];
    std::vector<char> seen(g.vflags.size(), 0);
    seen[root] = 1;
    for (size_t qi = 0; qi < st.vertex.size(); ++qi) {
      const uint32_t p = st.vertex[qi];
      for (uint32_t k = g.row_ptr[p]; k < g.row_ptr[p + 1]; ++k) {
        const uint32_t t = g.col[k];
        bool back = false;
        for (uint32_t k2 = g.row_ptr[t]; k2 < g.row_ptr[t + 1]; ++k2) back |= g.col[k2] == p;
        if (seen[t] || !(g.vflags[t] & HSPF_VF_NETWORK) || !back) continue;
        seen[t] = 1;
        st.vertex.push_back(t); st.base.push_back(st.total);
        st.total += g.row_ptr[t + 1] - g.row_ptr[t];
      }
    }
    return st;
  }
  void patch(Graph &gr, const std::vector<uint32_t> &vertices,
             const std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> &rows, const std::vector<uint8_t> &vflags) override {
    auto &g = static_cast<OracleGraph &>(gr);
    splice_rows(g.row_ptr, g.col, g.metric, g.vflags, vertices, rows, vflags);
    ++patches;
  }
  int patches = 0;
  struct OracleRun : DeviceRun { Tables t; Tables host_tables() override { return t; } };
  std::unique_ptr<DeviceRun> run_device(Graph &gr, const std::vector<uint32_t> &roots, uint32_t run_flags) override {
    auto r = std::make_unique<OracleRun>();
    r->t = run(gr, roots, run_flags);
    r->n_roots = r->t.n_roots; r->n_vertices = r->t.n_vertices; r->mask_words = r->t.mask_words;
    return r;
  }
  // restatement of the per-prefix reduction of hspf_routes_device (include/holo_spf_hip.h)
  RoutesOut routes(DeviceRun &run, const std::vector<uint32_t> &ptr, const std::vector<uint32_t> &vtx, const std::vector<uint32_t> &met, uint32_t flags) override {
    const Tables &t = static_cast<OracleRun &>(run).t;
    const uint32_t P = (uint32_t)ptr.size() - 1, W = t.mask_words, n = t.n_vertices;
    RoutesOut o;
    o.best_metric.assign((size_t)t.n_roots * P, 0xFFFFFFFFu); o.best_entry.assign((size_t)t.n_roots * P, 0xFFFFFFFFu); o.nexthop_mask.assign((size_t)t.n_roots * P * W, 0);
    const bool sat = flags & HSPF_PFX_SATURATING, last = flags & HSPF_PFX_LAST_MIN;
    for (uint32_t r = 0; r < t.n_roots; ++r)
      for (uint32_t p = 0; p < P; ++p) {
        uint32_t best = 0xFFFFFFFFu, ent = 0xFFFFFFFFu;
        std::vector<uint64_t> acc(W, 0);
        for (uint32_t e = ptr[p]; e < ptr[p + 1]; ++e) {
          const size_t i = (size_t)r * n + vtx[e];
          if (!(t.flags[i] & 1)) continue;
          uint32_t m = t.dist[i] + met[e];
          if (sat && m < t.dist[i]) m = 0xFFFFFFFFu;
          if (ent == 0xFFFFFFFFu || m < best || (last && m == best)) { best = m; ent = e; for (uint32_t w = 0; w < W; ++w) acc[w] = t.mask[i * W + w]; }
          else if (m == best) for (uint32_t w = 0; w < W; ++w) acc[w] |= t.mask[i * W + w];
        }
        const size_t oi = (size_t)r * P + p;
        o.best_metric[oi] = best; o.best_entry[oi] = ent;
        for (uint32_t w = 0; w < W; ++w) o.nexthop_mask[oi * W + w] = acc[w];
      }
    return o;
  }
  // the wire step: the tables stay host vectors here; the comparison restates k_routes_diff / k_routes_pack (include/holo_spf_hip.h
  // "RIB diff on device", "the hand-off of the diff")
  struct OracleRoutes : DeviceRoutes { RoutesOut t; RoutesOut host() override { return t; } };
  std::unique_ptr<DeviceRoutes> routes_device(DeviceRun &run, const std::vector<uint32_t> &ptr, const std::vector<uint32_t> &vtx, const std::vector<uint32_t> &met, uint32_t flags) override {
    auto o = std::make_unique<OracleRoutes>();
    o->t = routes(run, ptr, vtx, met, flags & ~(uint32_t)HSPF_PFX_RESIDENT);
    o->n_roots = run.n_roots; o->n_prefixes = (uint32_t)ptr.size() - 1; o->mask_words = run.mask_words;
    return o;
  }
  std::unique_ptr<DeviceRoutes> routes_upload(const RoutesOut &t, uint32_t n_roots, uint32_t n_prefixes, uint32_t mask_words) override {
    auto o = std::make_unique<OracleRoutes>();
    o->t = t; o->n_roots = n_roots; o->n_prefixes = n_prefixes; o->mask_words = mask_words;
    return o;
  }
  // several areas, one RIB: restatement of k_rib_fold (include/holo_spf_hip.h hspf_rib_fold_device) on host vectors
  struct OracleRib : OracleRoutes { std::vector<uint32_t> origin; };
  std::unique_ptr<DeviceRoutes> rib_new(uint32_t n_prefixes, uint32_t mask_words) override {
    auto o = std::make_unique<OracleRib>();
    o->n_roots = 1; o->n_prefixes = n_prefixes; o->mask_words = mask_words;
    o->t.best_metric.assign(n_prefixes, 0xFFFFFFFFu); o->t.best_entry.assign(n_prefixes, 0xFFFFFFFFu); o->t.nexthop_mask.assign((size_t)n_prefixes * mask_words, 0);
    o->origin.assign(n_prefixes, 0);
    return o;
  }
  void rib_fold(DeviceRoutes &rib_set, DeviceRun &run, const std::vector<uint32_t> &ptr, const std::vector<uint32_t> &vtx, const std::vector<uint32_t> &met,
                const std::vector<uint32_t> &org, const std::vector<uint32_t> &map, uint32_t area_index, uint32_t word_offset) override {
    auto &rib = static_cast<OracleRib &>(rib_set);
    const Tables &t = static_cast<OracleRun &>(run).t;
    const uint32_t W = t.mask_words, RW = rib.mask_words;
    for (uint32_t p = 0; p + 1 < ptr.size(); ++p) {
      const uint32_t ip = map[p];
      bool exists = rib.t.best_entry[ip] != 0xFFFFFFFFu;
      uint32_t bm = exists ? rib.t.best_metric[ip] : 0xFFFFFFFFu, bo = exists ? rib.origin[ip] : 0u, be = exists ? HSPF_PFX_KEPT_INIT : 0xFFFFFFFFu;
      std::vector<uint64_t> acc(W, 0);
      for (uint32_t e = ptr[p]; e < ptr[p + 1]; ++e) {
        const uint32_t v = vtx[e] & 0x7FFFFFFFu;
        if (!(t.flags[v] & 1)) continue;
        const uint64_t s = (uint64_t)t.dist[v] + met[e];
        const uint32_t m = s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s;
        if (exists && m > bm) continue;
        if ((vtx[e] & HSPF_PFX_ENTRY_NETWORK) && exists) {
          if (m < bm || (m == bm && org[e] > bo)) exists = false;
          else continue;
        }
        if (!exists || m < bm) { exists = true; bm = m; bo = org[e]; be = e; for (uint32_t w = 0; w < W; ++w) acc[w] = t.mask[(size_t)v * W + w]; }
        else for (uint32_t w = 0; w < W; ++w) acc[w] |= t.mask[(size_t)v * W + w];
      }
      if (be == 0xFFFFFFFFu) continue;
      if (be != HSPF_PFX_KEPT_INIT) for (uint32_t w = 0; w < RW; ++w) rib.t.nexthop_mask[(size_t)ip * RW + w] = 0;
      for (uint32_t w = 0; w < W; ++w) if (word_offset + w < RW) rib.t.nexthop_mask[(size_t)ip * RW + word_offset + w] |= acc[w];
      if (be != HSPF_PFX_KEPT_INIT) { rib.t.best_metric[ip] = bm; rib.origin[ip] = bo; rib.t.best_entry[ip] = (area_index << 24) | be; }
    }
  }
  RouteRecords routes_changed(DeviceRoutes &old_set, DeviceRoutes &new_set) override {
    const RoutesOut &a = static_cast<OracleRoutes &>(old_set).t, &b = static_cast<OracleRoutes &>(new_set).t;
    const uint32_t R = new_set.n_roots, P = new_set.n_prefixes, W = new_set.mask_words;
    RouteRecords out;
    out.mask_words = W;
    for (uint32_t r = 0; r < R; ++r)
      for (uint32_t p = 0; p < P; ++p) {
        const size_t i = (size_t)r * P + p;
        const bool had = a.best_entry[i] != 0xFFFFFFFFu, has = b.best_entry[i] != 0xFFFFFFFFu;
        bool same_nh = true, old_nh = false, new_nh = false;
        for (uint32_t w = 0; w < W; ++w) {
          same_nh = same_nh && a.nexthop_mask[i * W + w] == b.nexthop_mask[i * W + w];
          old_nh = old_nh || a.nexthop_mask[i * W + w] != 0; new_nh = new_nh || b.nexthop_mask[i * W + w] != 0;
        }
        uint32_t act;
        if (has) act = (had && a.best_metric[i] == b.best_metric[i] && same_nh) ? HSPF_DIFF_SAME : (new_nh ? HSPF_DIFF_INSTALL : HSPF_DIFF_SILENT);
        else act = had ? (old_nh ? HSPF_DIFF_WITHDRAW : HSPF_DIFF_SILENT) : HSPF_DIFF_SAME;
        if (act != HSPF_DIFF_INSTALL && act != HSPF_DIFF_WITHDRAW) continue;
        out.words.insert(out.words.end(), {r, p, act, b.best_metric[i], b.best_entry[i], 0u});
        for (uint32_t w = 0; w < W; ++w) { out.words.push_back((uint32_t)b.nexthop_mask[i * W + w]); out.words.push_back((uint32_t)(b.nexthop_mask[i * W + w] >> 32)); }
        out.old_words.insert(out.old_words.end(), {r, p, act, a.best_metric[i], a.best_entry[i], 0u});
        for (uint32_t w = 0; w < W; ++w) { out.old_words.push_back((uint32_t)a.nexthop_mask[i * W + w]); out.old_words.push_back((uint32_t)(a.nexthop_mask[i * W + w] >> 32)); }
      }
    return out;
  }
  Tables run(Graph &gr, const std::vector<uint32_t> &roots, uint32_t run_flags) override {
    auto &g = static_cast<OracleGraph &>(gr);
    Tables t;
    t.n_roots = (uint32_t)roots.size(); t.n_vertices = (uint32_t)g.vflags.size();
    uint32_t words = 1;
    for (uint32_t r : roots) words = std::max(words, (slot_table(gr, r).total + 63) / 64);
    t.mask_words = words;
    const size_t rn = (size_t)t.n_roots * t.n_vertices;
    t.dist.resize(rn); t.hops.resize(rn); t.flags.resize(rn); t.pop_rank.resize(rn); t.mask.resize(rn * words);
    const int rc = run_(t.n_vertices, (uint32_t)g.col.size(), g.row_ptr.data(), g.col.data(), g.metric.data(), g.vflags.data(), g.max_path,
                        roots.data(), t.n_roots, run_flags & 3u, variant, t.dist.data(), t.hops.data(), t.flags.data(), t.pop_rank.data(),
                        t.mask.data(), words, nullptr, nullptr, nullptr);
    if (rc != 0) throw std::runtime_error("oracle_spf_run failed");
    // like the real engine: tell the caller which roots did not pop in the static (distance, index) order
    const uint32_t n = t.n_vertices;
    for (uint32_t j = 0; j < t.n_roots; ++j) {
      std::vector<uint32_t> mem;
      for (uint32_t v = 0; v < n; ++v) if (t.flags[(size_t)j * n + v] & 1) mem.push_back(v);
      auto a = mem, b = mem;
      std::stable_sort(a.begin(), a.end(), [&](uint32_t x, uint32_t y) { return std::make_pair(t.dist[(size_t)j * n + x], x) < std::make_pair(t.dist[(size_t)j * n + y], y); });
      std::stable_sort(b.begin(), b.end(), [&](uint32_t x, uint32_t y) { return t.pop_rank[(size_t)j * n + x] < t.pop_rank[(size_t)j * n + y]; });
      if (a != b) for (uint32_t v : mem) t.flags[(size_t)j * n + v] |= HSPF_RF_EXACT;
    }
    if (!(run_flags & HSPF_RUN_POP_RANK)) t.pop_rank.clear();
    return t;
  }
 private:
  oracle_run_t run_;
};

