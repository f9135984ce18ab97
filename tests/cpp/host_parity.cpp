// tests/cpp/host_parity.cpp — the C++ host side (include/holo_spf_isis.hpp) against the RIBs the reference recorded
// in its conformance fixtures (tests/golden/{isis,isis_steps,ospfv2,ospfv2_steps}/*.json): vector -> Instance / areas ->
// compute_spf -> rows of `local-rib` (include/holo_spf_isis.hpp, include/holo_spf_ospf.hpp).
//
//   host_parity --engine hip    <files...>     the product path: libholo_spf_hip.so through the C ABI (needs an MI355X)
//   host_parity --engine oracle <files...>     CPU stand-in for the ENGINE ONLY (oracle/liboracle_spf.so, dlopen'ed;
//                                              TEST INFRASTRUCTURE): checks the host logic where there is no GPU
// Exit codes: 0 every vector reproduced, 1 mismatch / error, 77 --engine hip without a HIP device.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <numeric>

#include "holo_spf_isis.hpp"
#include "holo_spf_ospf.hpp"
#include "mini_json.hpp"

using namespace hspf::host;
namespace I = hspf::host::isis;
namespace O = hspf::host::ospf;

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
                        roots.data(), t.n_roots, run_flags & 3u, /*MAP*/ 1, t.dist.data(), t.hops.data(), t.flags.data(), t.pop_rank.data(),
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

// ---- vector -> Instance (schema of tools/make_golden.py) -----------------------------------------------------------
static I::SystemId sysid(const std::string &s) {                    // "0000.0000.0001"
  std::string hex;
  for (char c : s) if (c != '.') hex += c;
  I::SystemId out{};
  for (int i = 0; i < 6; ++i) out[i] = (uint8_t)std::stoi(hex.substr(2 * i, 2), nullptr, 16);
  return out;
}
static I::LanId lanid(const std::string &s) {                        // "0000.0000.0001.02"
  return I::LanId{sysid(s.substr(0, 14)), (uint8_t)std::stoi(s.substr(15, 2), nullptr, 16)};
}
static bool has_flag(const J &arr, const char *f) { for (auto &x : arr.arr) if (x.s == f) return true; return false; }

static I::Instance instance_from_vector(const J &vec) {
  I::Instance inst;
  const J &c = vec["config"];
  inst.config.system_id = sysid(c["system_id"].s);
  inst.config.level_type = c["level_type"].s;
  inst.config.metric_type = {{1, c["metric_type"]["1"].s}, {2, c["metric_type"]["2"].s}};
  inst.config.ipv4_enabled = !c["afs"].has("ipv4") || c["afs"]["ipv4"].b;
  inst.config.ipv6_enabled = !c["afs"].has("ipv6") || c["afs"]["ipv6"].b;
  inst.config.mt_ipv6_unicast = c["mt_ipv6_unicast"].b;
  inst.config.max_paths = (uint32_t)c["max_paths"].i();
  inst.config.att_ignore = c["att_ignore"].b;
  for (auto &a : c["area_addrs"].arr) inst.config.area_addrs.push_back(a.s);
  for (auto &i : vec["interfaces"].arr) {
    I::Interface f;
    f.name = i["name"].s; f.interface_type = i["type"].s;
    f.metric = {{1, (uint32_t)i["metric"]["1"].i()}, {2, (uint32_t)i["metric"]["2"].i()}};
    for (auto &a : i["adjacencies"].arr) {
      I::Adjacency d;
      d.system_id = sysid(a["system_id"].s); d.level_usage = a["usage"].s; d.state = a["state"].s;
      for (auto &x : a["ipv4"].arr) d.ipv4_addrs.push_back(x.s);
      for (auto &x : a["ipv6"].arr) d.ipv6_addrs.push_back(x.s);
      d.topologies.clear();
      for (auto &x : a["topologies"].arr) d.topologies.push_back((int)x.i());
      for (auto &x : a["area_addrs"].arr) d.area_addrs.push_back(x.s);
      d.snpa = f.name + "|" + a["system_id"].s + "|" + a["usage"].s;
      f.adjacencies.push_back(d);
    }
    inst.interfaces.push_back(f);
  }
  for (auto &lv : vec["lsdb"].obj) {
    I::Lsdb db;
    for (auto &l : lv.second.arr) {
      I::Lsp p;
      const std::string id = l["id"].s;                              // "0000.0000.0001.00-00"
      const I::LanId lan = lanid(id.substr(0, 17));
      p.system_id = lan.system_id; p.pseudonode = lan.pseudonode;
      p.fragment = (uint8_t)std::stoi(id.substr(18), nullptr, 16);
      if (l.has("seqno")) p.seqno = (uint32_t)l["seqno"].i();
      if (l.has("lifetime")) p.rem_lifetime = (uint16_t)l["lifetime"].i();
      p.overload = has_flag(l["flags"], "ol"); p.att = has_flag(l["flags"], "att");
      if (!l["protocols"].is_null()) { p.protocols_supported = std::vector<int>(); for (auto &x : l["protocols"].arr) p.protocols_supported->push_back((int)x.i()); }
      for (auto &m : l["mt"].arr) p.mt_flags[(int)m["id"].i()] = {has_flag(m["flags"], "ol"), has_flag(m["flags"], "att")};
      for (auto &x : l["is_reach"].arr) p.is_reach.push_back({lanid(x[0].s), (uint32_t)x[1].i()});
      for (auto &x : l["ext_is_reach"].arr) p.ext_is_reach.push_back({lanid(x[0].s), (uint32_t)x[1].i()});
      for (auto &x : l["mt_is_reach"].arr) p.mt_is_reach.push_back({(int)x[0].i(), lanid(x[1].s), (uint32_t)x[2].i()});
      for (auto &x : l["ipv4_int"].arr) p.ipv4_internal.push_back({x[0].s, (uint32_t)x[1].i()});
      for (auto &x : l["ipv4_ext"].arr) p.ipv4_external.push_back({x[0].s, (uint32_t)x[1].i()});
      for (auto &x : l["ext_ipv4"].arr) p.ext_ipv4.push_back({x[0].s, (uint32_t)x[1].i(), x[2].b});
      for (auto &x : l["ipv6"].arr) p.ipv6.push_back({x[0].s, (uint32_t)x[1].i(), x[2].b});
      for (auto &x : l["mt_ipv6"].arr) p.mt_ipv6.push_back({(int)x[0].i(), x[1].s, (uint32_t)x[2].i(), x[3].b});
      db.insert(std::move(p));
    }
    inst.lsdb[std::stoi(lv.first)] = std::move(db);
  }
  return inst;
}

static std::vector<O::Area> areas_from_vector(const J &vec) {        // schema of tools/make_golden_ospf.py
  std::vector<O::Area> out;
  for (auto &a : vec["areas"].arr) {
    O::Area ar;
    ar.area_id = a["area_id"].s;
    for (auto &r : a["routers"].arr) {
      O::RouterLsa l;
      l.adv_rtr = r["adv_rtr"].s; l.maxage = r.has("maxage") && r["maxage"].b;
      for (auto &k : r["links"].arr) l.links.push_back(O::RouterLink{k["type"].s, k["id"].s, k["data"].s, (uint32_t)k["metric"].i()});
      ar.routers.push_back(std::move(l));
    }
    for (auto &nw : a["networks"].arr) {
      O::NetworkLsa l;
      l.lsa_id = nw["lsa_id"].s; l.adv_rtr = nw["adv_rtr"].s; l.mask = nw["mask"].s; l.maxage = nw.has("maxage") && nw["maxage"].b;
      for (auto &x : nw["attached"].arr) l.attached.push_back(x.s);
      ar.networks.push_back(std::move(l));
    }
    for (auto &i : a["interfaces"].arr) {
      O::Interface f;
      f.name = i["name"].s; f.if_type = i["type"].s; f.index = i["index"].i();
      for (auto &nb : i["neighbors"].arr) f.neighbors.push_back(O::Neighbor{nb["router_id"].s, nb["src"].s});
      if (i.has("addrs")) for (auto &x : i["addrs"].arr) f.addrs.push_back(x.s);
      ar.interfaces.push_back(std::move(f));
    }
    out.push_back(std::move(ar));
  }
  return out;
}

// OSPFv2 vector: intra-area rows of the recorded local RIB (virtual-link endpoints are completed after the path by
// area::update_virtual_links and are left to the Python suite's literal restatement)
static bool ospf_rows_equal(const std::vector<O::RibRow> &rows, const J &rib);
static int check_ospf(const J &vec, Engine &eng, const std::string &path) {
  if (vec["has_vlinks"].b) return -1;
  const auto areas = areas_from_vector(vec);
  const auto rows = O::compute_spf_intra_area(vec["router_id"].s, areas, (uint32_t)vec["max_paths"].i(), eng);
  std::vector<const J *> want;
  for (auto &r : vec["rib"].arr) if (r["type"].s == "intra-area") want.push_back(&r);
  std::stable_sort(want.begin(), want.end(), [](const J *a, const J *b) { return parse_ip((*a)["prefix"].s) < parse_ip((*b)["prefix"].s); });
  bool same = want.size() == rows.size();
  for (size_t i = 0; same && i < rows.size(); ++i) {
    const J &w = *want[i];
    same = w["prefix"].s == rows[i].prefix && (uint32_t)w["metric"].i() == rows[i].metric && w["nexthops"].arr.size() == rows[i].nexthops.size();
    for (size_t k = 0; same && k < rows[i].nexthops.size(); ++k) {
      const J &a = w["nexthops"][k][0];
      same = (a.is_null() ? !rows[i].nexthops[k].first : (rows[i].nexthops[k].first && a.s == *rows[i].nexthops[k].first)) &&
             w["nexthops"][k][1].s == rows[i].nexthops[k].second;
    }
    if (!same) std::fprintf(stderr, "MISMATCH %s row %zu: %s metric %u (%zu next hops)\n", path.c_str(), i, rows[i].prefix.c_str(), rows[i].metric, rows[i].nexthops.size());
  }
  if (want.size() != rows.size()) std::fprintf(stderr, "MISMATCH %s: %zu rows, recorded %zu\n", path.c_str(), rows.size(), want.size());
  // the same RIB with both prefix reductions done by the engine (hspf_routes_device, OSPF rule flags)
  if (same && !ospf_rows_equal(O::intra_area_device_routes(vec["router_id"].s, areas, (uint32_t)vec["max_paths"].i(), eng), vec["rib"])) {
    std::fprintf(stderr, "DEVICE ROUTES MISMATCH %s\n", path.c_str());
    return 0;
  }
  return same ? 1 : 0;
}

// flooding::manet: (1) the reference's own hash vectors (manet.rs:205-232); (2) per vector, optional "manet" cases written by
// the Python suite from the literal restatement: {"level", "algo": zero-pruner|modified-manet|mixed, "tn", "lsp": [sys, pn, frag], "want": [sys...]}
static bool check_hash_kat() {
  struct K { uint8_t b[8]; uint16_t h; } kat[] = {{{1, 2, 3, 4, 5, 6, 0, 0x00}, 0x6215}, {{1, 2, 3, 4, 5, 6, 0, 0x07}, 0x6215},
                                                 {{1, 2, 3, 4, 5, 6, 0, 0x0F}, 0x6316}, {{0, 1, 2, 3, 4, 5, 0, 0x01}, 0x410F}};
  for (auto &k : kat) {
    I::SystemId s{}; for (int i = 0; i < 6; ++i) s[i] = k.b[i];
    if (I::flooding::flood_reduction_hash(s, k.b[6], k.b[7]) != k.h) return false;
  }
  return true;
}
static int check_manet(const J &vec, const I::Instance &inst, Engine &eng, const std::string &path) {
  int bad = 0;
  std::map<std::pair<int, std::string>, std::map<I::SystemId, I::flooding::NeighborCache>> caches;
  for (auto &c : vec["manet"].arr) {
    const int level = (int)c["level"].i();
    const std::string algo = c["algo"].s;
    auto key = std::make_pair(level, algo);
    if (!caches.count(key)) {
      std::function<std::string(const I::SystemId &)> f;
      if (algo == "modified-manet") f = [](const I::SystemId &) { return std::string("modified-manet"); };
      else if (algo == "mixed") f = [](const I::SystemId &s) { return std::string((s[5] & 1) ? "modified-manet" : "zero-pruner"); };
      caches[key] = I::flooding::init_cache(level, inst, eng, f);
    }
    const auto got = I::flooding::reflood_list(caches[key], inst.config.system_id, sysid(c["tn"].s), sysid(c["lsp"][0].s),
                                               (uint8_t)c["lsp"][1].i(), (uint8_t)c["lsp"][2].i());
    std::vector<I::SystemId> want;
    for (auto &w : c["want"].arr) want.push_back(sysid(w.s));
    if (got != want) { ++bad; std::fprintf(stderr, "MANET MISMATCH %s tn %s\n", path.c_str(), c["tn"].s.c_str()); }
  }
  return bad;
}

static bool rows_equal(const std::vector<I::RibRow> &rows, const J &rib) {
  std::vector<const J *> want;
  for (auto &r : rib.arr) want.push_back(&r);
  std::stable_sort(want.begin(), want.end(), [](const J *a, const J *b) { return parse_ip((*a)["prefix"].s) < parse_ip((*b)["prefix"].s); });
  if (want.size() != rows.size()) return false;
  for (size_t i = 0; i < rows.size(); ++i) {
    const J &w = *want[i];
    if (w["prefix"].s != rows[i].prefix || (uint32_t)w["metric"].i() != rows[i].metric || (int)w["level"].i() != rows[i].level ||
        w["nexthops"].arr.size() != rows[i].nexthops.size()) return false;
    for (size_t k = 0; k < rows[i].nexthops.size(); ++k)
      if (w["nexthops"][k][0].s != rows[i].nexthops[k].first || w["nexthops"][k][1].s != rows[i].nexthops[k].second) return false;
  }
  return true;
}

// A reference step test replayed as "topology snapshot, then the LSDB after the step": the level graphs built for the
// snapshot are brought forward with row patches (LevelGraph::refresh -> Engine::patch) or rebuilt, must equal graphs
// derived from scratch, and the SPF on them must give the RIB the reference recorded after the step.
static int replay_isis_step(const J &step, const std::string &golden_dir, Engine &eng, int &patched) {
  const std::string src = step["source"].s;                       // "... (snapshot topo2-1/rt6, state ...)"
  const size_t a = src.find("snapshot ");
  if (a == std::string::npos) return -1;
  const size_t sl = src.find('/', a), co = src.find(',', a);
  const std::string topo = src.substr(a + 9, sl - a - 9), rt = src.substr(sl + 1, co - sl - 1);
  const J base = load_json(golden_dir + "/isis/" + topo + "_" + rt + ".json");
  const I::Instance inst0 = instance_from_vector(base), inst1 = instance_from_vector(step);
  I::GraphCache cache;
  if (!rows_equal(I::compute_spf(inst0, eng, &cache), base["rib"])) return 0;
  std::map<int, std::vector<I::LanId>> trig;
  static const I::Lsdb empty;
  for (int level : {1, 2}) {
    auto i0 = inst0.lsdb.find(level), i1 = inst1.lsdb.find(level);
    trig[level] = I::changed_lan_ids(i0 == inst0.lsdb.end() ? empty : i0->second, i1 == inst1.lsdb.end() ? empty : i1->second);
  }
  const int before = cache.patched;
  if (!rows_equal(I::compute_spf(inst1, eng, &cache, &trig), step["rib"])) return 0;
  patched += cache.patched - before;
  for (auto &kv : cache.graphs) {
    const int level = std::get<0>(kv.first), mt = std::get<1>(kv.first);
    const auto levels = inst1.config.levels();
    if (std::find(levels.begin(), levels.end(), level) == levels.end() || !inst1.config.is_topology_enabled(mt)) continue;
    I::LevelGraph fresh(inst1, level, mt < 0 ? std::optional<int>() : std::optional<int>(mt), std::get<2>(kv.first));
    const I::LevelGraph &g = *kv.second;
    if (!(g.vids == fresh.vids) || g.row_ptr != fresh.row_ptr || g.col != fresh.col || g.metric != fresh.metric || g.vflags != fresh.vflags) return 0;
  }
  return 1;
}

static std::vector<O::v3::Area> areas3_from_vector(const J &vec) {
  std::vector<O::v3::Area> out;
  for (auto &a : vec["areas"].arr) {
    O::v3::Area ar;
    ar.area_id = a["area_id"].s;
    for (auto &r : a["routers"].arr) {
      O::v3::RouterLsa l;
      l.adv_rtr = r["adv_rtr"].s; l.lsa_id = (uint32_t)r["lsa_id"].i(); l.maxage = r.has("maxage") && r["maxage"].b;
      for (auto &o : r["options"].arr) l.options.push_back(o.s);
      for (auto &k : r["links"].arr) l.links.push_back(O::v3::RouterLink{k["type"].s, (uint32_t)k["iface_id"].i(), (uint32_t)k["nbr_iface_id"].i(), k["nbr_router_id"].s, (uint32_t)k["metric"].i()});
      ar.routers.push_back(std::move(l));
    }
    for (auto &nw : a["networks"].arr) {
      O::v3::NetworkLsa l; l.adv_rtr = nw["adv_rtr"].s; l.lsa_id = (uint32_t)nw["lsa_id"].i(); l.maxage = nw.has("maxage") && nw["maxage"].b;
      for (auto &x : nw["attached"].arr) l.attached.push_back(x.s);
      ar.networks.push_back(std::move(l));
    }
    for (auto &p : a["iaps"].arr) {
      O::v3::IntraAreaPrefixLsa l;
      l.adv_rtr = p["adv_rtr"].s; l.lsa_id = (uint32_t)p["lsa_id"].i(); l.ref_type = p["ref_type"].s; l.ref_lsa_id = (uint32_t)p["ref_lsa_id"].i();
      l.ref_adv_rtr = p["ref_adv_rtr"].s; l.maxage = p.has("maxage") && p["maxage"].b;
      for (auto &x : p["prefixes"].arr) { O::v3::Prefix q; q.prefix = x["prefix"].s; q.metric = (uint32_t)x["metric"].i(); for (auto &o : x["options"].arr) q.options.push_back(o.s); l.prefixes.push_back(std::move(q)); }
      ar.iaps.push_back(std::move(l));
    }
    for (auto &i : a["interfaces"].arr) {
      O::v3::Interface f; f.name = i["name"].s; f.if_type = i["type"].s; f.index = i["index"].i(); f.iface_id = (uint32_t)i["iface_id"].i();
      for (auto &l : i["link_lsas"].arr) f.link_lsas.push_back(O::v3::LinkLsa{l["adv_rtr"].s, (uint32_t)l["lsa_id"].i(), l["lladdr"].s});
      ar.interfaces.push_back(std::move(f));
    }
    out.push_back(std::move(ar));
  }
  return out;
}

static bool ospf_rows_equal(const std::vector<O::RibRow> &rows, const J &rib) {
  std::vector<const J *> want;
  for (auto &r : rib.arr) if (r["type"].s == "intra-area") want.push_back(&r);
  std::stable_sort(want.begin(), want.end(), [](const J *a, const J *b) { return parse_ip((*a)["prefix"].s) < parse_ip((*b)["prefix"].s); });
  if (want.size() != rows.size()) return false;
  for (size_t i = 0; i < rows.size(); ++i) {
    const J &w = *want[i];
    if (w["prefix"].s != rows[i].prefix || (uint32_t)w["metric"].i() != rows[i].metric || w["nexthops"].arr.size() != rows[i].nexthops.size()) return false;
    for (size_t k = 0; k < rows[i].nexthops.size(); ++k) {
      const J &a = w["nexthops"][k][0];
      if (!(a.is_null() ? !rows[i].nexthops[k].first : (rows[i].nexthops[k].first && a.s == *rows[i].nexthops[k].first)) ||
          w["nexthops"][k][1].s != rows[i].nexthops[k].second) return false;
    }
  }
  return true;
}
static int replay_ospf_step(const J &step, const std::string &golden_dir, Engine &eng, int &patched) {
  if (step["has_vlinks"].b) return -1;
  const std::string src = step["source"].s;
  const size_t a = src.find("snapshot ");
  if (a == std::string::npos) return -1;
  const size_t sl = src.find('/', a), co = src.find(',', a);
  const J base = load_json(golden_dir + "/ospfv2/" + src.substr(a + 9, sl - a - 9) + "_" + src.substr(sl + 1, co - sl - 1) + ".json");
  if (base["has_vlinks"].b) return -1;
  const auto areas0 = areas_from_vector(base), areas1 = areas_from_vector(step);
  O::GraphCache cache;
  if (!ospf_rows_equal(O::compute_spf_intra_area(base["router_id"].s, areas0, (uint32_t)base["max_paths"].i(), eng, &cache), base["rib"])) return 0;
  std::map<std::string, std::vector<O::VertexId>> trig;
  for (auto &n : areas1) for (auto &o : areas0) if (o.area_id == n.area_id) trig[n.area_id] = O::changed_vertex_ids(o, n);
  const int before = cache.patched;
  if (!ospf_rows_equal(O::compute_spf_intra_area(step["router_id"].s, areas1, (uint32_t)step["max_paths"].i(), eng, &cache, &trig), step["rib"])) return 0;
  patched += cache.patched - before;
  for (auto &n : areas1) {
    O::AreaGraph fresh(n);
    const O::AreaGraph &g = *cache.graphs.at(n.area_id);
    if (g.vids != fresh.vids || g.row_ptr != fresh.row_ptr || g.col != fresh.col || g.metric != fresh.metric || g.link_pos != fresh.link_pos) return 0;
  }
  return 1;
}

int main(int argc, char **argv) {
  std::string engine = "hip", oracle_so = "oracle/liboracle_spf.so", golden_dir;
  std::vector<std::string> files;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--engine") && i + 1 < argc) engine = argv[++i];
    else if (!strcmp(argv[i], "--oracle-so") && i + 1 < argc) oracle_so = argv[++i];
    else if (!strcmp(argv[i], "--replay-steps") && i + 1 < argc) golden_dir = argv[++i];
    else files.push_back(argv[i]);
  }
  std::unique_ptr<Engine> eng;
  try {
    if (engine == "hip") {
      if (hspf_device_count() <= 0) { std::printf("no HIP device: the product engine cannot run here\n"); return 77; }
      eng = std::make_unique<HipEngine>(0);
    } else eng = std::make_unique<OracleEngine>(oracle_so);
  } catch (const std::exception &e) { std::fprintf(stderr, "engine: %s\n", e.what()); return 1; }
  int ok = 0, bad = 0, skipped = 0, manet_cases = 0, manet_bad = 0, steps_ok = 0, steps_bad = 0, steps_patched = 0, dev_ok = 0, dev_bad = 0;
  if (!check_hash_kat()) { std::fprintf(stderr, "flood_reduction_hash: reference vectors not reproduced\n"); return 1; }
  for (auto &path : files) {
    try {
      const J vec = load_json(path);
      if (vec["proto"].s == "ospfv2") {
        if (!golden_dir.empty() && vec["source"].s.find("snapshot ") != std::string::npos) {
          const int sr = replay_ospf_step(vec, golden_dir, *eng, steps_patched);
          if (sr > 0) ++steps_ok; else if (sr == 0) { ++steps_bad; std::fprintf(stderr, "STEP REPLAY MISMATCH %s\n", path.c_str()); }
        }
        const int r = check_ospf(vec, *eng, path);
        if (r > 0) ++ok; else if (r == 0) ++bad; else ++skipped;
        continue;
      }
      if (vec["proto"].s == "ospfv3") {
        if (vec["has_vlinks"].b) { ++skipped; continue; }
        const auto rows3 = O::v3::compute_spf_intra_area(vec["router_id"].s, areas3_from_vector(vec), (uint32_t)vec["max_paths"].i(), *eng, vec["af"].s);
        if (ospf_rows_equal(rows3, vec["rib"])) ++ok; else { ++bad; std::fprintf(stderr, "MISMATCH %s (ospfv3)\n", path.c_str()); }
        continue;
      }
      if (vec["proto"].s != "isis") continue;
      if (!golden_dir.empty() && vec["source"].s.find("snapshot ") != std::string::npos) {
        const int r = replay_isis_step(vec, golden_dir, *eng, steps_patched);
        if (r > 0) ++steps_ok; else if (r == 0) { ++steps_bad; std::fprintf(stderr, "STEP REPLAY MISMATCH %s\n", path.c_str()); }
      }
      const I::Instance inst = instance_from_vector(vec);
      if (vec.has("manet")) { const int mb = check_manet(vec, inst, *eng, path); manet_cases += (int)vec["manet"].size(); manet_bad += mb; }
      const auto rows = I::compute_spf(inst, *eng);
      // the same RIB with the prefix attachment done by the engine (hspf_run_device + hspf_routes_device)
      if (rows_equal(I::compute_spf_device_routes(inst, *eng), vec["rib"])) ++dev_ok; else { ++dev_bad; std::fprintf(stderr, "DEVICE ROUTES MISMATCH %s\n", path.c_str()); }
      // recorded rows in BTreeMap<IpNetwork, _> order
      std::vector<const J *> want;
      for (auto &r : vec["rib"].arr) want.push_back(&r);
      std::stable_sort(want.begin(), want.end(), [](const J *a, const J *b) { return parse_ip((*a)["prefix"].s) < parse_ip((*b)["prefix"].s); });
      bool same = want.size() == rows.size();
      for (size_t i = 0; same && i < rows.size(); ++i) {
        const J &w = *want[i];
        same = w["prefix"].s == rows[i].prefix && (uint32_t)w["metric"].i() == rows[i].metric && (int)w["level"].i() == rows[i].level &&
               w["nexthops"].arr.size() == rows[i].nexthops.size();
        for (size_t k = 0; same && k < rows[i].nexthops.size(); ++k)
          same = w["nexthops"][k][0].s == rows[i].nexthops[k].first && w["nexthops"][k][1].s == rows[i].nexthops[k].second;
        if (!same) std::fprintf(stderr, "MISMATCH %s row %zu: %s metric %u level %d (%zu next hops)\n", path.c_str(), i, rows[i].prefix.c_str(),
                                rows[i].metric, rows[i].level, rows[i].nexthops.size());
      }
      if (same) ++ok; else { ++bad; if (want.size() != rows.size()) std::fprintf(stderr, "MISMATCH %s: %zu rows, recorded %zu\n", path.c_str(), rows.size(), want.size()); }
    } catch (const std::exception &e) { ++bad; std::fprintf(stderr, "ERROR %s: %s\n", path.c_str(), e.what()); }
  }
  std::printf("host_parity (%s engine): %d vectors reproduce the recorded local RIB, %d do not, %d skipped (virtual links)\n", engine.c_str(), ok, bad, skipped);
  if (manet_cases) std::printf("host_parity: %d reflood lists checked, %d differ\n", manet_cases, manet_bad);
  if (steps_ok + steps_bad) std::printf("host_parity: %d step tests replayed through patched graphs (%d row-patch refreshes), %d differ\n", steps_ok + steps_bad, steps_patched, steps_bad);
  if (dev_ok + dev_bad) std::printf("host_parity: %d IS-IS RIBs also derived with the prefix attachment on the engine, %d differ\n", dev_ok + dev_bad, dev_bad);
  return (bad || manet_bad || steps_bad || dev_bad) ? 1 : 0;
}
