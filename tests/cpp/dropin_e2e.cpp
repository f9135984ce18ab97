// tests/cpp/dropin_e2e.cpp — what the DROP-IN costs end to end, measured through the compiled host side.
//
// The Rust glue a maintainer adds (rust/holo-isis/src/spf/hip.rs) does, per SPF event, what include/holo_spf_isis.hpp does
// here in C++ (the reference's toolchain is absent from this image): walk the LSDB into a CSR (once, then per changed LSP),
// hand the roots to the engine, take the per-vertex results back, rebuild `Spt` (holo-isis/src/spf.rs:224-242: one ordered-
// map insert per vertex), replay the first-hop slots through the unchanged resolve_nexthop (:956-1010), attach the prefixes
// (compute_routes, :840-949).  This driver builds a synthetic IS-IS level-2 LSDB at LSP level — N routers on an 8-neighbour
// grid plus random chords, 10 N directed adjacencies (BASELINE configs[2] at N = 100 000), wide metrics U[1,100], one
// loopback /32 per router and an anycast /24 on every fifth — and times every stage of that path:
//
//   dropin_e2e --engine hip    [--n N] [--reps K]      the product engine (libholo_spf_hip.so, C ABI; needs an MI355X)
//   dropin_e2e --engine oracle [--n N] [--reps K]      the SAME host code on the CPU stand-in for the engine (oracle/
//                                                      liboracle_spf.so, binary-heap variant; TEST INFRASTRUCTURE)
//   --no-packed            hspf_run (16 B per vertex) instead of hspf_run_packed for the hand-off (A/B)
//   --batch R              also: R roots in ONE run (the shape of flooding::manet::init_cache, manet.rs:47-69), hand-off + rebuild
//   --dump-json F          the instance in the schema of tests/golden/isis/*.json (the literal restatement oracle/isis_ref.py reads it)
//   --dump-rib F           the RIB rows compute_spf produced ("rib" of the same schema)
// One JSON object on stdout.  Exit codes: 0 ok, 1 error, 77 --engine hip without a HIP device.
#include <malloc.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>

#include "holo_spf_isis.hpp"
#include "oracle_engine.hpp"

namespace I = hspf::host::isis;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Rng {                                                        // xorshift64* 
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed ? seed : 1) {}
  uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
  uint32_t below(uint32_t m) { return (uint32_t)(next() % m); }
};

static I::SystemId sysid_of(uint32_t i) {                           // router i: system id = i + 1, big endian (index order = VertexId order)
  const uint64_t v = (uint64_t)i + 1;
  return I::SystemId{(uint8_t)(v >> 40), (uint8_t)(v >> 32), (uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
}
static std::string sysid_text(const I::SystemId &s) {
  char b[32];
  snprintf(b, sizeof b, "%02x%02x.%02x%02x.%02x%02x", s[0], s[1], s[2], s[3], s[4], s[5]);
  return b;
}
static std::string loopback_of(uint32_t i) {
  char b[40];
  snprintf(b, sizeof b, "10.%u.%u.%u/32", (i >> 16) & 0xFF, (i >> 8) & 0xFF, i & 0xFF);
  return b;
}

struct Synth {
  I::Instance inst;
  uint32_t n = 0;
  size_t entries = 0, prefixes = 0;
};

// N routers, 5 N links (10 N directed adjacencies): 8-neighbour grid, then random chords up to the count.
static Synth make_instance(uint32_t n, uint64_t seed) {
  Synth S;
  S.n = n;
  Rng rng(seed);
  uint32_t cols = 1;
  while ((uint64_t)cols * cols < n) ++cols;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> adj(n);    // per router: (neighbour, cost of MY direction), in LSP order
  std::vector<std::pair<uint32_t, uint32_t>> links;
  auto add = [&](uint32_t a, uint32_t b) { if (a != b && a < n && b < n) links.push_back({a, b}); };
  for (uint32_t v = 0; v < n; ++v) {
    const uint32_t r = v / cols, c = v % cols;
    if (c + 1 < cols) add(v, v + 1);
    if (v + cols < n) add(v, v + cols);
    if (c + 1 < cols && v + cols + 1 < n) add(v, v + cols + 1);
    if (c > 0 && v + cols - 1 < n) add(v, v + cols - 1);
    (void)r;
  }
  const size_t want = (size_t)5 * n;
  while (links.size() > want) links.pop_back();
  while (links.size() < want && n > 2) { const uint32_t a = rng.below(n), b = rng.below(n); if (a != b) links.push_back({a, b}); }
  for (auto &l : links) {
    adj[l.first].push_back({l.second, 1 + rng.below(100)});
    adj[l.second].push_back({l.first, 1 + rng.below(100)});
  }
  I::InstanceCfg &cfg = S.inst.config;
  cfg.system_id = sysid_of(0);
  cfg.level_type = "level-2";
  cfg.metric_type = {{1, "wide"}, {2, "wide"}};
  cfg.ipv4_enabled = true; cfg.ipv6_enabled = false; cfg.mt_ipv6_unicast = false;
  cfg.max_paths = 16;
  cfg.area_addrs = {"49.0000"};
  I::Lsdb db;
  for (uint32_t v = 0; v < n; ++v) {
    I::Lsp p;
    p.system_id = sysid_of(v);
    p.protocols_supported = std::vector<int>{I::NLPID_IPV4};
    for (auto &e : adj[v]) p.ext_is_reach.push_back({I::LanId{sysid_of(e.first), 0}, e.second});
    p.ext_ipv4.push_back({loopback_of(v), rng.below(10), false});
    ++S.prefixes;
    if (v % 5 == 0) {                                                 // an anycast /24 shared by routers v and v + 5 k: ties and merges
      char b[40];
      const uint32_t grp = (v / 5) % 50000;
      snprintf(b, sizeof b, "192.%u.%u.0/24", 100 + (grp >> 8), grp & 0xFF);
      p.ext_ipv4.push_back({b, 10, false});
    }
    S.entries += adj[v].size();
    db.insert(std::move(p));
  }
  S.inst.lsdb[2] = std::move(db);
  // self = router 0: one point-to-point circuit per neighbour, metric = the cost its LSP advertises for it
  uint32_t k = 0;
  for (auto &e : adj[0]) {
    I::Interface f;
    char nm[32];
    snprintf(nm, sizeof nm, "eth%04u", k);
    f.name = nm; f.interface_type = "point-to-point";
    f.metric = {{1, e.second}, {2, e.second}};
    I::Adjacency a;
    a.system_id = sysid_of(e.first); a.level_usage = "level-2"; a.state = "up";
    char ip[40];
    snprintf(ip, sizeof ip, "172.16.%u.%u", k >> 8, (k & 0xFF));
    a.ipv4_addrs = {ip}; a.topologies = {0}; a.area_addrs = {"49.0000"};
    a.snpa = f.name + "|" + sysid_text(a.system_id) + "|level-2";
    f.adjacencies.push_back(a);
    S.inst.interfaces.push_back(f);
    ++k;
  }
  return S;
}

static void dump_json(const Synth &S, const std::string &path) {
  std::ofstream o(path);
  const I::InstanceCfg &c = S.inst.config;
  o << "{\"proto\": \"isis\", \"source\": \"tests/cpp/dropin_e2e.cpp synthetic\", \"config\": {\"afs\": {\"ipv4\": true, \"ipv6\": false}, \"area_addrs\": [\"49.0000\"], "
       "\"att_ignore\": false, \"level_type\": \"level-2\", \"max_paths\": " << c.max_paths << ", \"metric_type\": {\"1\": \"wide\", \"2\": \"wide\"}, "
       "\"mt_ipv6_unicast\": false, \"system_id\": \"" << sysid_text(c.system_id) << "\"},\n\"interfaces\": [";
  bool first = true;
  for (auto &f : S.inst.interfaces) {
    o << (first ? "" : ", ") << "{\"name\": \"" << f.name << "\", \"type\": \"" << f.interface_type << "\", \"metric\": {\"1\": " << f.metric.at(1) << ", \"2\": " << f.metric.at(2)
      << "}, \"adjacencies\": [";
    bool fa = true;
    for (auto &a : f.adjacencies) {
      o << (fa ? "" : ", ") << "{\"area_addrs\": [\"49.0000\"], \"ipv4\": [\"" << a.ipv4_addrs[0] << "\"], \"ipv6\": [], \"state\": \"up\", \"system_id\": \"" << sysid_text(a.system_id)
        << "\", \"topologies\": [0], \"usage\": \"level-2\"}";
      fa = false;
    }
    o << "]}";
    first = false;
  }
  o << "],\n\"lsdb\": {\"2\": [";
  first = true;
  for (auto &kv : S.inst.lsdb.at(2).all()) {
    const I::Lsp &l = kv.second;
    o << (first ? "" : ",\n") << "{\"id\": \"" << sysid_text(l.system_id) << ".00-00\", \"flags\": [], \"protocols\": [204], \"mt\": [], \"is_reach\": [], \"mt_is_reach\": [], "
         "\"ipv4_int\": [], \"ipv4_ext\": [], \"ipv6\": [], \"mt_ipv6\": [], \"ext_is_reach\": [";
    bool fe = true;
    for (auto &e : l.ext_is_reach) { o << (fe ? "" : ", ") << "[\"" << sysid_text(e.first.system_id) << ".00\", " << e.second << "]"; fe = false; }
    o << "], \"ext_ipv4\": [";
    fe = true;
    for (auto &e : l.ext_ipv4) { o << (fe ? "" : ", ") << "[\"" << std::get<0>(e) << "\", " << std::get<1>(e) << ", false]"; fe = false; }
    o << "]}";
    first = false;
  }
  o << "]}, \"rib\": []}\n";
}

static void dump_rib(const std::vector<I::RibRow> &rows, const std::string &path) {
  std::ofstream o(path);
  o << "[";
  for (size_t i = 0; i < rows.size(); ++i) {
    o << (i ? ",\n" : "") << "{\"prefix\": \"" << rows[i].prefix << "\", \"metric\": " << rows[i].metric << ", \"level\": " << rows[i].level << ", \"nexthops\": [";
    for (size_t k = 0; k < rows[i].nexthops.size(); ++k)
      o << (k ? ", " : "") << "[\"" << rows[i].nexthops[k].first << "\", \"" << rows[i].nexthops[k].second << "\"]";
    o << "]}";
  }
  o << "]\n";
}

// Forwards to the engine and keeps the time of every call kind (the stages of the drop-in that are the ENGINE's).
class TimedEngine : public Engine {
 public:
  explicit TimedEngine(Engine &e) : e_(e) {}
  double upload_ms = 0, run_ms = 0, patch_ms = 0, slot_ms = 0, run_device_ms = 0, routes_ms = 0;
  void reset() { upload_ms = run_ms = patch_ms = slot_ms = run_device_ms = routes_ms = 0; }
  std::unique_ptr<Graph> upload(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, const std::vector<uint32_t> &c, const std::vector<uint8_t> &d, uint32_t m) override {
    const auto t = Clock::now(); auto r = e_.upload(a, b, c, d, m); upload_ms += ms_since(t); return r;
  }
  std::unique_ptr<Graph> upload_keyed(const std::vector<uint64_t> &vk, const std::vector<uint32_t> &vr, const std::vector<uint64_t> &tk, const std::vector<uint32_t> &tm,
                                      const std::vector<uint8_t> &vf, uint32_t mp, std::vector<uint32_t> &rank, std::vector<uint32_t> &rp, std::vector<uint32_t> &col,
                                      std::vector<uint32_t> &met, std::vector<uint8_t> &fl) override {
    const auto t = Clock::now(); auto r = e_.upload_keyed(vk, vr, tk, tm, vf, mp, rank, rp, col, met, fl); upload_ms += ms_since(t); return r;
  }
  Tables run(Graph &g, const std::vector<uint32_t> &roots, uint32_t fl) override { const auto t = Clock::now(); auto r = e_.run(g, roots, fl); run_ms += ms_since(t); return r; }
  SlotTable slot_table(Graph &g, uint32_t root) override { const auto t = Clock::now(); auto r = e_.slot_table(g, root); slot_ms += ms_since(t); return r; }
  void patch(Graph &g, const std::vector<uint32_t> &v, const std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> &rows, const std::vector<uint8_t> &f) override {
    const auto t = Clock::now(); e_.patch(g, v, rows, f); patch_ms += ms_since(t);
  }
  std::unique_ptr<DeviceRun> run_device(Graph &g, const std::vector<uint32_t> &roots, uint32_t fl) override { const auto t = Clock::now(); auto r = e_.run_device(g, roots, fl); run_device_ms += ms_since(t); return r; }
  RoutesOut routes(DeviceRun &run, const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, const std::vector<uint32_t> &c, uint32_t fl) override {
    const auto t = Clock::now(); auto r = e_.routes(run, a, b, c, fl); routes_ms += ms_since(t); return r;
  }
  std::unique_ptr<DeviceRoutes> routes_device(DeviceRun &run, const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, const std::vector<uint32_t> &c, uint32_t fl) override {
    const auto t = Clock::now(); auto r = e_.routes_device(run, a, b, c, fl); routes_ms += ms_since(t); return r;
  }
  std::unique_ptr<DeviceRoutes> routes_upload(const RoutesOut &t, uint32_t a, uint32_t b, uint32_t c) override { return e_.routes_upload(t, a, b, c); }
  RouteRecords routes_changed(DeviceRoutes &o, DeviceRoutes &n) override { return e_.routes_changed(o, n); }
  std::unique_ptr<DeviceRoutes> rib_new(uint32_t a, uint32_t b) override { return e_.rib_new(a, b); }
  void rib_fold(DeviceRoutes &rib, DeviceRun &run, const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, const std::vector<uint32_t> &c, const std::vector<uint32_t> &d,
                const std::vector<uint32_t> &m, uint32_t ai, uint32_t off) override { e_.rib_fold(rib, run, a, b, c, d, m, ai, off); }
 private:
  Engine &e_;
};

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char **argv) {
  std::string engine = "oracle", json_path, rib_path;
  uint32_t n = 100000, reps = 3, batch = 0;
  bool packed = true;
  std::string self = argv[0];
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(1); } return argv[++i]; };
    if (a == "--engine") engine = next();
    else if (a == "--n") n = (uint32_t)std::stoul(next());
    else if (a == "--reps") reps = (uint32_t)std::stoul(next());
    else if (a == "--batch") batch = (uint32_t)std::stoul(next());
    else if (a == "--no-packed") packed = false;
    else if (a == "--dump-json") json_path = next();
    else if (a == "--dump-rib") rib_path = next();
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
  }
  // A long-running daemon's heap is warm: freed blocks are reused, not handed back to the kernel.  Without this every
  // run's 1.6 MB of table vectors is a fresh mmap whose first touch faults page by page (2 ms per run on the GPU box's
  // micro-VM against 0.09 ms for decoding the words: profiles/r05c_dropin_e2e_hip.json) — the harness, not the path.
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  try {
    std::unique_ptr<Engine> eng;
    HipEngine *hip = nullptr;
    if (engine == "hip") {
      if (hspf_device_count() <= 0) { fprintf(stderr, "no HIP device\n"); return 77; }
      auto h = std::make_unique<HipEngine>(0);
      h->use_packed = packed;
      hip = h.get();
      eng = std::move(h);
    } else {
      const std::string dir = self.substr(0, self.find_last_of('/') == std::string::npos ? 0 : self.find_last_of('/'));
      auto o = std::make_unique<OracleEngine>((dir.empty() ? std::string(".") : dir) + "/../../oracle/liboracle_spf.so");
      o->variant = 2;                                               // binary heap: the fastest CPU loop with identical outputs
      eng = std::move(o);
    }
    TimedEngine te(*eng);
    auto t0 = Clock::now();
    Synth S = make_instance(n, 0x9E3779B97F4A7C15ull);
    const double gen_ms = ms_since(t0);
    if (!json_path.empty()) dump_json(S, json_path);
    const I::Instance &inst = S.inst;

    // ---- stage 1: LSDB -> CSR (first time), upload.  The host walk (every link's target looked up among the vertices on one
    // core) for the record; then what the drop-in does since round 6: the LSDB's records streamed to the engine, which ranks
    // the vertices, resolves the targets and builds the graph (Engine::upload_keyed = hspf_graph_upload_keyed) — the CSR it
    // built must be the host walk's, array by array.
    t0 = Clock::now();
    double csr_host_ms = 0, csr_keyed_engine_ms = 0;
    bool keyed_same = true;
    {
      I::LevelGraph H(inst, 2, I::MT_STANDARD, false);
      csr_host_ms = ms_since(t0);
      if (hip) {
        te.reset();
        I::LevelGraph K(inst, 2, I::MT_STANDARD, false, &te);
        keyed_same = K.vids == H.vids && K.row_ptr == H.row_ptr && K.col == H.col && K.metric == H.metric && K.vflags == H.vflags;
      }
    }
    te.reset();
    t0 = Clock::now();
    I::GraphCache cache;
    if (hip) cache.keyed = &te;
    I::LevelGraph &G = cache.get(inst, 2, I::MT_STANDARD, false, nullptr);
    const double csr_first_ms = ms_since(t0);
    csr_keyed_engine_ms = te.upload_ms;
    t0 = Clock::now();
    G.device(te);
    const double upload_ms = ms_since(t0);

    // ---- stage 2..4: one SPF of the instance's own root (the reference's everyday call), stage by stage
    std::vector<double> run_v, rebuild_v, routes_v, total_v, decode_v, alloc_v, call_v;
    std::vector<I::RibRow> rows;
    size_t spt_size = 0, rib_size = 0;
    for (uint32_t k = 0; k < reps + 1; ++k) {                       // (+1: the first repetition warms the engine up and is dropped)
      te.reset();
      const auto ta = Clock::now();
      I::Spt spt = I::compute_spt(2, inst.config.system_id, true, I::MT_STANDARD, false, inst, te, &G);
      const double spt_ms = ms_since(ta);
      const auto tb = Clock::now();
      std::map<IpKey, I::Route> rib;
      I::compute_routes(2, I::MT_STANDARD, inst, spt, rib);
      const double rt_ms = ms_since(tb);
      spt_size = spt.vertices.size(); rib_size = rib.size();
      if (k == 0) continue;
      run_v.push_back(te.run_ms); rebuild_v.push_back(spt_ms - te.run_ms); routes_v.push_back(rt_ms); total_v.push_back(spt_ms + rt_ms);
      if (hip) { decode_v.push_back(hip->last_handoff.decode_ms); alloc_v.push_back(hip->last_handoff.alloc_ms); call_v.push_back(hip->last_handoff.run_ms); }
    }
    // the whole call as the caller sees it (graph from the cache): compute_spf
    const std::map<int, std::vector<I::LanId>> no_triggers;         // (no LSP changed: the cached graph is kept; without a list the cache rebuilds)
    t0 = Clock::now();
    rows = I::compute_spf(inst, te, &cache, &no_triggers);
    const double compute_spf_ms = ms_since(t0);
    if (!rib_path.empty()) dump_rib(rows, rib_path);
    // the same call with SPT and prefix attachment on the device (graph from the cache as well; checked against `rows`)
    double dev_routes_cached_ms = 0;
    bool dev_routes_cached_same = true;
    {
      t0 = Clock::now();
      auto rows3 = I::compute_spf_device_routes(inst, te, &cache, &no_triggers);
      dev_routes_cached_ms = ms_since(t0);
      dev_routes_cached_same = rows3.size() == rows.size();
      for (size_t i = 0; dev_routes_cached_same && i < rows.size(); ++i)
        dev_routes_cached_same = rows3[i].prefix == rows[i].prefix && rows3[i].metric == rows[i].metric && rows3[i].nexthops == rows[i].nexthops;
    }

    // ---- stage 1b: incremental LSDB -> CSR.  (i) one LSP re-originated with another metric on its first adjacency (cost
    // only), (ii) an adjacency withdrawn by both ends and announced again (structural: two rows change length)
    Synth S2 = S;                                                   // (the twin diffs snapshots; the reference keeps `trigger_lsps`)
    I::Lsdb &db2 = S2.inst.lsdb[2];
    const uint32_t u = n / 3;
    auto lsp_of = [&](I::Lsdb &db, uint32_t v) -> I::Lsp { return db.all().at(I::Lsdb::Key{sysid_of(v), 0, 0}); };
    double inc_cost_ms = 0, inc_cost_patch_ms = 0, inc_struct_ms = 0, inc_struct_patch_ms = 0;
    {
      I::Lsp l = lsp_of(db2, u);
      l.ext_is_reach[0].second = l.ext_is_reach[0].second % 100 + 1;
      l.seqno++;
      db2.insert(l);
      te.reset();
      t0 = Clock::now();
      const bool ok = G.refresh(S2.inst, {I::LanId{sysid_of(u), 0}});
      inc_cost_ms = ms_since(t0); inc_cost_patch_ms = te.patch_ms;
      if (!ok) throw std::runtime_error("refresh (cost) fell back to a rebuild");
    }
    {
      I::Lsp l = lsp_of(db2, u);
      const I::LanId nb = l.ext_is_reach.back().first;
      l.ext_is_reach.pop_back(); l.seqno++;
      db2.insert(l);
      uint32_t w = 0;
      for (int b = 0; b < 6; ++b) w = (w << 8) | nb.system_id[b];
      w -= 1;
      I::Lsp m = lsp_of(db2, w);
      for (size_t i = 0; i < m.ext_is_reach.size(); ++i)
        if (m.ext_is_reach[i].first.system_id == sysid_of(u)) { m.ext_is_reach.erase(m.ext_is_reach.begin() + i); break; }
      m.seqno++;
      db2.insert(m);
      te.reset();
      t0 = Clock::now();
      const bool ok = G.refresh(S2.inst, {I::LanId{sysid_of(u), 0}, I::LanId{sysid_of(w), 0}});
      inc_struct_ms = ms_since(t0); inc_struct_patch_ms = te.patch_ms;
      if (!ok) throw std::runtime_error("refresh (structural) fell back to a rebuild");
    }
    // the patched graph must be the graph of the changed LSDB: SPF on it against a graph derived from scratch
    bool patched_ok = true;
    {
      I::LevelGraph fresh(S2.inst, 2, I::MT_STANDARD, false);
      patched_ok = fresh.row_ptr == G.row_ptr && fresh.col == G.col && fresh.metric == G.metric && fresh.vflags == G.vflags;
      I::Spt a = I::compute_spt(2, inst.config.system_id, true, I::MT_STANDARD, false, S2.inst, te, &G);
      I::Spt b = I::compute_spt(2, inst.config.system_id, true, I::MT_STANDARD, false, S2.inst, te, &fresh);
      patched_ok = patched_ok && a.vertices.size() == b.vertices.size();
      for (auto &kv : a.vertices) {
        const I::Vertex *o = b.get(kv.first);
        patched_ok = patched_ok && o && o->distance == kv.second.distance && o->hops == kv.second.hops && o->nexthops.size() == kv.second.nexthops.size();
      }
    }

    // ---- the path on which nobody rebuilds an Spt: SPT + prefix attachment on the device, one RIB back
    double dev_routes_ms = 0;
    bool dev_routes_same = true;
    {
      te.reset();
      t0 = Clock::now();
      auto rows2 = I::compute_spf_device_routes(inst, te);           // nothing cached: the graph straight from the LSDB records
      dev_routes_ms = ms_since(t0);
      dev_routes_same = rows2.size() == rows.size();
      for (size_t i = 0; dev_routes_same && i < rows.size(); ++i)
        dev_routes_same = rows2[i].prefix == rows[i].prefix && rows2[i].metric == rows[i].metric && rows2[i].nexthops == rows[i].nexthops;
      dev_routes_same = dev_routes_same && dev_routes_cached_same;
    }
    const double dev_routes_engine_ms = te.upload_ms + te.run_device_ms + te.routes_ms + te.slot_ms;
    // ---- the RUNNING instance: graph, prefix table and the previous route tables resident on the device; per LSP change only
    // the changed records come back (RibPipeline = LevelGraph::refresh + hspf_run_device + hspf_routes_device +
    // hspf_routes_diff_device + hspf_routes_pack x 2 + expansion into RouteIpAdd / RouteIpDel).  Checked against the host
    // rule (update_global_rib on two compute_spf RIBs) on the same change.
    double pipe_first_ms = 0, pipe_step_ms = 0;
    size_t pipe_first_msgs = 0, pipe_msgs = 0, pipe_records = 0;
    bool pipe_ok = true;
    I::RibPipeline::Timing pt;
    {
      std::map<std::string, int> ifindex;
      int ix = 1;
      for (auto &f : inst.interfaces) ifindex[f.name] = ix++;
      t0 = Clock::now();
      I::RibPipeline pipe(inst, te, 2, I::MT_STANDARD, ifindex);
      auto first = pipe.step(inst, {});
      pipe_first_ms = ms_since(t0); pipe_first_msgs = first.size();
      auto host_first = I::update_global_rib(rows, {}, ifindex);
      pipe_ok = pipe_ok && first == host_first;
      // an LSP far from the root re-originated with another metric on one adjacency: a handful of routes move
      Synth S3 = S;
      const uint32_t uu = (n / 2) + 7 < n ? (n / 2) + 7 : n - 1;
      std::vector<double> step_v;
      std::vector<I::RibRow> before = rows;
      for (int it = 0; it < 4; ++it) {
        I::Lsp l = S3.inst.lsdb[2].all().at(I::Lsdb::Key{sysid_of(uu), 0, 0});
        const I::Lsp &orig = S.inst.lsdb.at(2).all().at(I::Lsdb::Key{sysid_of(uu), 0, 0});
        for (size_t e = 0; e < l.ext_is_reach.size(); ++e) l.ext_is_reach[e].second = (it & 1) ? orig.ext_is_reach[e].second : 1u;   // every adjacency of the router at cost 1 / back
        l.seqno++;
        S3.inst.lsdb[2].insert(l);
        t0 = Clock::now();
        auto msgs = pipe.step(S3.inst, {I::LanId{sysid_of(uu), 0}});
        step_v.push_back(ms_since(t0));
        if (it == 2) { pt = pipe.last; pipe_msgs = msgs.size(); pipe_records = pt.records; }
        if (it < 2) {                                                 // the host rule on the same change (two full compute_spf: slow, twice is enough)
          auto after = I::compute_spf(S3.inst, *eng);
          pipe_ok = pipe_ok && msgs == I::update_global_rib(after, before, ifindex);
          before = after;
        }
      }
      pipe_step_ms = median(step_v);
    }

    // ---- R roots in one run (flooding::manet::init_cache shape): hand-off and rebuild
    double batch_run_ms = 0, batch_rebuild_ms = 0, batch_decode_ms = 0;
    bool batch_same = true;
    if (batch) {
      std::vector<I::SystemId> rs;
      for (uint32_t i = 0; i < batch; ++i) rs.push_back(sysid_of((uint32_t)((uint64_t)i * n / batch)));
      for (int k = 0; k < 2; ++k) {
        te.reset();
        t0 = Clock::now();
        auto spts = I::compute_spts(2, rs, false, I::MT_STANDARD, false, inst, te, &G);
        const double all = ms_since(t0);
        batch_run_ms = te.run_ms; batch_rebuild_ms = all - te.run_ms;
        if (hip) batch_decode_ms = hip->last_handoff.decode_ms;
        if (k) continue;
        // the roots rebuilt side by side (threads) against one root at a time: vertices, distances, hops, next hops, pop order
        for (size_t i = 0; i < rs.size() && batch_same; ++i) {
          I::Spt one = I::compute_spt(2, rs[i], false, I::MT_STANDARD, false, inst, te, &G);
          batch_same = one.vertices.size() == spts[i].vertices.size() && one.pop_order == spts[i].pop_order;
          auto a = one.vertices.begin();
          auto b = spts[i].vertices.begin();
          for (; batch_same && a != one.vertices.end(); ++a, ++b) {
            batch_same = a->first == b->first && a->second.distance == b->second.distance && a->second.hops == b->second.hops &&
                         a->second.nexthops.size() == b->second.nexthops.size();
            for (size_t h = 0; batch_same && h < a->second.nexthops.size(); ++h) batch_same = a->second.nexthops[h]->system_id == b->second.nexthops[h]->system_id;
          }
        }
      }
      // the same for HOP-COUNT SPTs (flooding::manet::init_cache: mt_id none, every link cost 1 / 0; the rank of a pseudonode is
      // looked up through a per-root cache) — small LSDBs only: the graph is walked out of the LSDB on the host.  With the CPU stand-in
      // only: it checks the twin's threaded rebuild (written in a session without a GPU; the engine's hop-count runs have their own
      // GPU tests: test_cpp_flooding_manet_reflood_lists_on_gpu, tests/test_gpu_golden.py)
      if (n <= 20000 && !hip) {
        auto hc = I::compute_spts(2, rs, false, std::nullopt, true, inst, te);
        for (size_t i = 0; i < rs.size() && batch_same; ++i) {
          I::Spt one = I::compute_spt(2, rs[i], false, std::nullopt, true, inst, te);
          batch_same = one.vertices.size() == hc[i].vertices.size() && one.pop_order == hc[i].pop_order;
          auto a = one.vertices.begin();
          auto b = hc[i].vertices.begin();
          for (; batch_same && a != one.vertices.end(); ++a, ++b)
            batch_same = a->first == b->first && a->second.distance == b->second.distance && a->second.hops == b->second.hops && a->second.nexthops.size() == b->second.nexthops.size();
        }
      }
    }

    const double run = median(run_v), rebuild = median(rebuild_v), routes = median(routes_v), total = median(total_v);
    struct St { const char *name; double ms; } stages[] = {{"run_and_handoff", run}, {"spt_rebuild", rebuild}, {"compute_routes", routes}};
    const St *slow = &stages[0];
    for (auto &s : stages) if (s.ms > slow->ms) slow = &s;
    printf("{\"engine\": \"%s\", \"packed_handoff\": %s, \"n_routers\": %u, \"adjacency_entries\": %zu, \"prefix_entries\": %zu, \"root_neighbours\": %zu, "
           "\"generate_ms\": %.2f, \"lsdb_to_csr_first_ms\": %.2f, \"lsdb_to_csr_first_is\": \"%s\", \"lsdb_to_csr_engine_part_ms\": %.2f, \"lsdb_to_csr_host_walk_ms\": %.2f, "
           "\"keyed_csr_identical_to_host_walk\": %s, \"graph_upload_ms\": %.3f, "
           "\"lsdb_to_csr_incremental\": {\"cost_only_ms\": %.3f, \"cost_only_engine_patch_ms\": %.3f, \"structural_ms\": %.3f, \"structural_engine_patch_ms\": %.3f, \"patched_graph_identical\": %s}, "
           "\"one_root\": {\"run_and_handoff_ms\": %.3f, \"engine_call_ms\": %.3f, \"table_alloc_ms\": %.3f, \"handoff_decode_ms\": %.3f, \"spt_rebuild_ms\": %.2f, \"compute_routes_ms\": %.2f, \"spt_plus_routes_ms\": %.2f, "
           "\"compute_spf_call_ms\": %.2f, \"spt_vertices\": %zu, \"rib_routes\": %zu, \"slowest_stage\": \"%s\"}, "
           "\"device_routes_path\": {\"compute_spf_device_routes_ms\": %.2f, \"graph_from_cache_ms\": %.2f, \"engine_calls_ms\": %.3f, \"same_rib\": %s}, "
           "\"running_instance_pipeline\": {\"first_step_ms\": %.2f, \"first_step_messages\": %zu, \"lsp_change_step_ms\": %.3f, \"stages_ms\": {\"refresh_patch\": %.3f, \"run_device\": %.3f, "
           "\"routes_device\": %.3f, \"slot_nexthops\": %.3f, \"diff_pack\": %.3f, \"expand\": %.3f}, \"records_to_host\": %zu, \"messages\": %zu, \"identical_to_host_rule\": %s}",
           engine.c_str(), (hip && packed) ? "true" : "false", n, S.entries, S.prefixes + (n + 4) / 5, inst.interfaces.size(),
           gen_ms, csr_first_ms, hip ? "records streamed to the engine (hspf_graph_upload_keyed): graph resident when it returns" : "host walk", csr_keyed_engine_ms, csr_host_ms,
           keyed_same ? "true" : "false", upload_ms, inc_cost_ms, inc_cost_patch_ms, inc_struct_ms, inc_struct_patch_ms, patched_ok ? "true" : "false",
           run, median(call_v), median(alloc_v), median(decode_v), rebuild, routes, total, compute_spf_ms, spt_size, rib_size, slow->name,
           dev_routes_ms, dev_routes_cached_ms, dev_routes_engine_ms, dev_routes_same ? "true" : "false",
           pipe_first_ms, pipe_first_msgs, pipe_step_ms, pt.refresh_ms, pt.run_ms, pt.routes_ms, pt.slots_ms, pt.diff_pack_ms, pt.expand_ms, pipe_records, pipe_msgs, pipe_ok ? "true" : "false");
    if (batch) printf(", \"batch\": {\"roots\": %u, \"run_and_handoff_ms\": %.3f, \"handoff_decode_ms\": %.3f, \"spt_rebuild_ms\": %.2f, \"same_as_one_root_at_a_time\": %s}", batch, batch_run_ms, batch_decode_ms, batch_rebuild_ms, batch_same ? "true" : "false");
    printf("}\n");
    return (patched_ok && dev_routes_same && pipe_ok && keyed_same) ? 0 : 1;
  } catch (const std::exception &e) {
    fprintf(stderr, "dropin_e2e: %s\n", e.what());
    return 1;
  }
}
