// tests/cpp/host_parity.cpp — the C++ host side (include/holo_spf_isis.hpp) against the RIBs the reference recorded
// in its conformance fixtures (tests/golden/{isis,isis_steps,ospfv2,ospfv2_steps}/*.json): vector -> Instance / areas ->
// compute_spf -> rows of `local-rib` (include/holo_spf_isis.hpp, include/holo_spf_ospf.hpp).
//
//   host_parity --engine hip    <files...>     the product path: libholo_spf_hip.so through the C ABI (needs an MI355X)
//   host_parity --engine oracle <files...>     CPU stand-in for the ENGINE ONLY (oracle/liboracle_spf.so, dlopen'ed;
//                                              TEST INFRASTRUCTURE): checks the host logic where there is no GPU
// Exit codes: 0 every vector reproduced, 1 mismatch / error, 77 --engine hip without a HIP device.
#include <cstdio>
#include <cstring>
#include <numeric>
#include <set>

#include "holo_spf_isis.hpp"
#include "holo_spf_ospf.hpp"
#include "mini_json.hpp"

using namespace hspf::host;
namespace I = hspf::host::isis;
namespace O = hspf::host::ospf;

#include "oracle_engine.hpp"

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
  inst.config.sr_enabled = c.has("sr_enabled") && c["sr_enabled"].b;
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
      if (l.has("sr_cap") && !l["sr_cap"].is_null()) {               // segment routing (tests/_random_isis.py add_sr)
        I::SrCap cap;
        for (auto &f : l["sr_cap"]["flags"].arr) cap.flags.push_back(f.s);
        for (auto &r : l["sr_cap"]["srgb"].arr) cap.srgb.push_back({(uint32_t)r[0].i(), (uint32_t)r[1].i()});
        p.sr_cap = std::move(cap);
      }
      if (l.has("sr_algos")) for (auto &x : l["sr_algos"].arr) p.sr_algos.push_back((int)x.i());
      if (l.has("prefix_sids"))
        for (auto &kind : l["prefix_sids"].obj)
          for (auto &e : kind.second.obj) {
            I::PrefixSid sid;
            for (auto &f : e.second["flags"].arr) sid.flags.push_back(f.s);
            if (e.second.has("index")) sid.index = (uint32_t)e.second["index"].i();
            if (e.second.has("label")) sid.label = (uint32_t)e.second["label"].i();
            p.prefix_sids[kind.first][std::stoi(e.first)] = std::move(sid);
          }
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
// Lsdb::fragments_from (the forward cursor of the vertex walks) against iter_for_lan_id + zeroth_lsp: ascending visits (the
// stepping path), visits that skip far ahead, repeated and DESCENDING visits (the descent fallback), LAN ids without any LSP.
static bool check_lsdb_cursor() {
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  I::Lsdb db;
  std::vector<I::LanId> lans;
  for (int i = 0; i < 400; ++i) {
    I::LanId lan{};
    lan.system_id[4] = (uint8_t)(rnd() % 40); lan.system_id[5] = (uint8_t)(rnd() % 6);
    lan.pseudonode = rnd() % 3 ? 0 : (uint8_t)(rnd() % 4);
    lans.push_back(lan);
    if (rnd() % 5 == 0) continue;                                       // a LAN id nobody originated
    const int nf = 1 + (int)(rnd() % 12);
    for (int f = 0; f < nf; ++f) {
      if (f == 0 && rnd() % 6 == 0) continue;                          // no zeroth fragment
      I::Lsp l; l.system_id = lan.system_id; l.pseudonode = lan.pseudonode; l.fragment = (uint8_t)f;
      l.seqno = rnd() % 7 ? 1 : 0; l.rem_lifetime = rnd() % 7 ? 100 : 0;
      db.insert(l);
    }
  }
  auto same = [&](I::Lsdb::Cursor &c, const I::LanId &lan) {
    std::vector<const I::Lsp *> got;
    const I::Lsp *z = db.fragments_from(c, lan, got);
    return got == db.iter_for_lan_id(lan) && z == db.zeroth_lsp(lan);
  };
  std::vector<I::LanId> asc = lans;
  std::sort(asc.begin(), asc.end());
  { I::Lsdb::Cursor c; for (auto &lan : asc) if (!same(c, lan)) return false; }                          // ascending, duplicates included
  { I::Lsdb::Cursor c; for (size_t i = 0; i < asc.size(); i += 1 + rnd() % 9) if (!same(c, asc[i])) return false; }   // skipping ahead
  { I::Lsdb::Cursor c; for (size_t i = asc.size(); i-- > 0;) if (!same(c, asc[i])) return false; }       // descending
  { I::Lsdb::Cursor c; for (int i = 0; i < 3000; ++i) if (!same(c, lans[rnd() % lans.size()])) return false; }   // any order
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
    if (w.has("sr_label") != rows[i].sr) return false;                // the SR columns (instances with sr_enabled)
    if (rows[i].sr) {
      auto same = [](const J &j, const std::optional<uint32_t> &v) { return j.is_null() ? !v : (v && (uint32_t)j.i() == *v); };
      if (!same(w["sr_label"], rows[i].sr_label) || w["nexthop_labels"].arr.size() != rows[i].nexthop_labels.size()) return false;
      for (size_t k = 0; k < rows[i].nexthop_labels.size(); ++k) if (!same(w["nexthop_labels"][k], rows[i].nexthop_labels[k])) return false;
    }
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
  // further events on the SAME cache (random chains: "next" holds whole vectors): patch upon patch, rows spliced in place
  if (step.has("next")) {
    I::Instance prev = instance_from_vector(step);
    for (auto &nx : step["next"].arr) {
      const I::Instance cur = instance_from_vector(nx);
      std::map<int, std::vector<I::LanId>> tr;
      for (int level : {1, 2}) {
        const I::Instance &was = prev;
        auto i0 = was.lsdb.find(level), i1 = cur.lsdb.find(level);
        tr[level] = I::changed_lan_ids(i0 == prev.lsdb.end() ? empty : i0->second, i1 == cur.lsdb.end() ? empty : i1->second);
      }
      const int b4 = cache.patched;
      if (!rows_equal(I::compute_spf(cur, eng, &cache, &tr), nx["rib"])) { std::fprintf(stderr, "  replay: RIB differs at a later step of the chain\n"); return 0; }
      patched += cache.patched - b4;
      for (auto &kv : cache.graphs) {
        const int level = std::get<0>(kv.first), mt = std::get<1>(kv.first);
        const auto levels = cur.config.levels();
        if (std::find(levels.begin(), levels.end(), level) == levels.end() || !cur.config.is_topology_enabled(mt)) continue;
        I::LevelGraph fresh(cur, level, mt < 0 ? std::optional<int>() : std::optional<int>(mt), std::get<2>(kv.first));
        const I::LevelGraph &g = *kv.second;
        if (!(g.vids == fresh.vids) || g.row_ptr != fresh.row_ptr || g.col != fresh.col || g.metric != fresh.metric || g.vflags != fresh.vflags) { std::fprintf(stderr, "  replay: a graph patched several times differs from a fresh one\n"); return 0; }
      }
      prev = cur;
    }
  }
  return 1;
}

// The wire step against what the reference RECORDED on the ibus for the step (`ibus_routes`), three ways:
// the host rule (update_global_rib on compute_spf's rows and `rib_before`), the one-shot device form (SPT, attachment,
// comparison with `rib_before`, compaction and packing on the engine: update_global_rib_device), and the running-instance
// pipeline (RibPipeline: first step from an empty RIB on the SNAPSHOT, then the step's LSDB as a set of changed LSPs —
// its messages must turn the snapshot's installed routes into the step's).  1 all equal, 0 a difference, -1 not a wire vector.
static std::vector<I::RibRow> rib_rows(const J &rib) {
  std::vector<I::RibRow> out;
  for (auto &r : rib.arr) {
    I::RibRow row{r["prefix"].s, (uint32_t)r["metric"].i(), (int)r["level"].i(), {}};
    for (auto &nh : r["nexthops"].arr) row.nexthops.push_back({nh[0].s, nh[1].s});
    out.push_back(std::move(row));
  }
  return out;
}
static void dump(const char *what, const std::vector<I::IbusMsg> &v) {
  for (auto &m : v) { std::fprintf(stderr, "    %s %s %s metric %u:", what, m.add ? "add" : "del", m.prefix.c_str(), m.metric); for (auto &n : m.nexthops) std::fprintf(stderr, " (%d, %s)", n.first, n.second.c_str()); std::fprintf(stderr, "\n"); }
}
static int check_isis_wire(const J &vec, const std::string &golden_dir, Engine &eng, size_t &records, size_t &prefixes, int &pipelines) {
  if (!vec.has("ibus_routes") || !vec.has("rib_before")) return -1;
  if (vec["source"].s.find("nb-config-summary") != std::string::npos) return -1;   // summary routes are configuration, not SPF output (as in tests/test_host_isis.py)
  std::map<std::string, int> ifindex;
  for (auto &kv : vec["ifindex"].obj) ifindex[kv.first] = (int)kv.second.i();
  std::vector<I::IbusMsg> want;
  for (auto &m : vec["ibus_routes"].arr) {
    I::IbusMsg w{m["op"].s == "add", m["prefix"].s, m.has("metric") ? (uint32_t)m["metric"].i() : 0u, {}};
    if (m.has("nexthops")) for (auto &nh : m["nexthops"].arr) w.nexthops.push_back({(int)nh[0].i(), nh[1].s});
    want.push_back(std::move(w));
  }
  const I::Instance inst = instance_from_vector(vec);
  const std::vector<I::RibRow> before = rib_rows(vec["rib_before"]);
  if (!(I::update_global_rib(I::compute_spf(inst, eng), before, ifindex) == want)) { std::fprintf(stderr, "  wire: host rule differs\n"); return 0; }
  std::vector<std::pair<int, int>> tabs;
  for (int lv : inst.config.levels()) for (int mt : {I::MT_STANDARD, I::MT_IPV6_UNICAST}) if (inst.config.is_topology_enabled(mt)) tabs.push_back({lv, mt});
  if (tabs.size() != 1) return 1;                                    // (two tables: the L1 / L2 merge is host logic)
  size_t nr = 0, np = 0;
  if (!(I::update_global_rib_device(inst, eng, before, ifindex, &nr, &np) == want)) { std::fprintf(stderr, "  wire: device form differs\n"); return 0; }
  // (records are CANDIDATES; on the recorded fixtures they stay within four of the messages — random instances with parallel links
  // and re-ordered slots, marked "random", may hand more pairs to the host's compare)
  if (nr > np || (np && !vec.has("random") && nr > want.size() + 4)) { std::fprintf(stderr, "  wire: %zu records for %zu messages\n", nr, want.size()); return 0; }
  records += nr; prefixes += np;
  // the running instance: snapshot first, then the step as changed LSPs — when the step kept interfaces and configuration
  const std::string src = vec["source"].s;
  const size_t a = src.find("snapshot ");
  if (golden_dir.empty() || a == std::string::npos) return 1;
  const size_t sl = src.find('/', a), co = src.find(',', a);
  const J base = load_json(golden_dir + "/isis/" + src.substr(a + 9, sl - a - 9) + "_" + src.substr(sl + 1, co - sl - 1) + ".json");
  const I::Instance inst0 = instance_from_vector(base);
  auto same_ifaces = [&]() {
    if (inst0.interfaces.size() != inst.interfaces.size()) return false;
    for (size_t i = 0; i < inst.interfaces.size(); ++i) {
      const I::Interface &x = inst0.interfaces[i], &y = inst.interfaces[i];
      if (x.name != y.name || x.interface_type != y.interface_type || x.metric != y.metric || x.adjacencies.size() != y.adjacencies.size()) return false;
      for (size_t k = 0; k < x.adjacencies.size(); ++k)
        if (x.adjacencies[k].system_id != y.adjacencies[k].system_id || x.adjacencies[k].state != y.adjacencies[k].state || x.adjacencies[k].ipv4_addrs != y.adjacencies[k].ipv4_addrs ||
            x.adjacencies[k].ipv6_addrs != y.adjacencies[k].ipv6_addrs || x.adjacencies[k].level_usage != y.adjacencies[k].level_usage || x.adjacencies[k].topologies != y.adjacencies[k].topologies) return false;
    }
    const I::InstanceCfg &c0 = inst0.config, &c1 = inst.config;
    return c0.level_type == c1.level_type && c0.metric_type == c1.metric_type && c0.ipv4_enabled == c1.ipv4_enabled && c0.ipv6_enabled == c1.ipv6_enabled &&
           c0.mt_ipv6_unicast == c1.mt_ipv6_unicast && c0.att_ignore == c1.att_ignore && c0.max_paths == c1.max_paths && c0.area_addrs == c1.area_addrs;
  };
  if (!same_ifaces()) return 1;
  const int level = tabs[0].first;
  I::RibPipeline pipe(inst0, eng, level, tabs[0].second, ifindex);
  const auto first = pipe.step(inst0, {});
  if (!(first == I::update_global_rib(rib_rows(base["rib"]), {}, ifindex))) { std::fprintf(stderr, "  wire: pipeline first step differs\n"); return 0; }
  static const I::Lsdb empty;
  auto i0 = inst0.lsdb.find(level), i1 = inst.lsdb.find(level);
  const auto trig = I::changed_lan_ids(i0 == inst0.lsdb.end() ? empty : i0->second, i1 == inst.lsdb.end() ? empty : i1->second);
  const auto msgs = pipe.step(inst, trig);
  // (the recorded sequence starts from `rib_before`, which is the snapshot's RIB whenever the step kept everything else)
  if (!(msgs == I::update_global_rib(rib_rows(vec["rib"]), rib_rows(base["rib"]), ifindex))) {
    std::fprintf(stderr, "  wire: pipeline step differs (%zu messages)\n", msgs.size());
    if (getenv("HSPF_PARITY_DUMP")) {
      dump("got ", msgs);
      dump("want", I::update_global_rib(rib_rows(vec["rib"]), rib_rows(base["rib"]), ifindex));
      std::fprintf(stderr, "    changed LAN ids: %zu, pipeline full %d, records %zu\n", trig.size(), (int)pipe.last.full, pipe.last.records);
    }
    return 0;
  }
  // further steps on the SAME pipeline (random chains, tests/test_cpp_driver.py): the previous route tables stay on the engine
  // and the comparison runs there — "next" holds whole vectors (LSDB + the restatement's RIB), interfaces and configuration kept
  if (vec.has("next")) {
    I::Instance prev_inst = instance_from_vector(vec);
    std::vector<I::RibRow> prev_rib = rib_rows(vec["rib"]);
    for (auto &nx : vec["next"].arr) {
      I::Instance cur = instance_from_vector(nx);
      auto p0 = prev_inst.lsdb.find(level), p1 = cur.lsdb.find(level);
      const auto tr = I::changed_lan_ids(p0 == prev_inst.lsdb.end() ? empty : p0->second, p1 == cur.lsdb.end() ? empty : p1->second);
      const auto got = pipe.step(cur, tr);
      const std::vector<I::RibRow> cur_rib = rib_rows(nx["rib"]);
      if (!(got == I::update_global_rib(cur_rib, prev_rib, ifindex))) {
        std::fprintf(stderr, "  wire: pipeline differs at a later step of the chain (%zu messages)\n", got.size());
        if (getenv("HSPF_PARITY_DUMP")) {
          dump("got ", got); dump("want", I::update_global_rib(cur_rib, prev_rib, ifindex));
          std::fprintf(stderr, "    changed LAN ids: %zu, pipeline full %d, records %zu\n", tr.size(), (int)pipe.last.full, pipe.last.records);
        }
        return 0;
      }
      // ... and what the pipeline holds as installed is exactly the RIB's rows with next hops (prefix, metric, next hops)
      {
        std::map<IpKey, const I::RibRow *> inst_rows;
        for (auto &r : cur_rib) if (!r.nexthops.empty()) inst_rows[parse_ip(r.prefix)] = &r;
        const auto held = pipe.rib();
        bool same = inst_rows.size() == held.size();
        for (auto &kv : held) {
          auto it = inst_rows.find(kv.first);
          if (it == inst_rows.end() || it->second->metric != kv.second.metric || !I::detail::same_nexthops(it->second->nexthops, kv.second.nexthops)) {
            same = false;
            if (getenv("HSPF_PARITY_DUMP")) std::fprintf(stderr, "    installed set: %s metric %u (%zu next hops) is not a row of the RIB with these next hops\n", kv.second.prefix.c_str(), kv.second.metric, kv.second.nexthops.size());
          }
        }
        if (!same) { std::fprintf(stderr, "  wire: the pipeline's installed set differs from the RIB after step %zu of the chain (%zu / %zu rows)\n", (size_t)(&nx - &vec["next"].arr[0]) + 1, held.size(), inst_rows.size()); return 0; }
      }
      prev_inst = std::move(cur);
      prev_rib = cur_rib;
    }
  }
  ++pipelines;
  return 1;
}

// OSPFv2: the recorded ibus sequences of the step vectors from the host rule and from ENGINE tables (every area folded into one
// RIB on the engine: Engine::rib_fold = hspf_rib_fold_device).  1 equal, 0 a difference, -1 not a wire vector.
static std::vector<O::RibRow> ospf_rib_rows(const J &rib) {
  std::vector<O::RibRow> out;
  for (auto &r : rib.arr) {
    O::RibRow row{r["prefix"].s, (uint32_t)r["metric"].i(), {}};
    row.type = r.has("type") ? r["type"].s : std::string("intra-area");
    for (auto &nh : r["nexthops"].arr) row.nexthops.push_back({nh[0].is_null() ? std::optional<std::string>() : std::optional<std::string>(nh[0].s), nh[1].s});
    out.push_back(std::move(row));
  }
  return out;
}
static int check_ospf_wire(const J &vec, Engine &eng, size_t &records, size_t &prefixes, int &multi_area) {
  if (!vec.has("ibus_routes") || !vec.has("rib_before")) return -1;
  std::map<std::string, int> ifindex;
  for (auto &kv : vec["ifindex"].obj) ifindex[kv.first] = (int)kv.second.i();
  std::vector<IbusMsg> want;
  for (auto &m : vec["ibus_routes"].arr) {
    IbusMsg w{m["op"].s == "add", m["prefix"].s, m.has("metric") ? (uint32_t)m["metric"].i() : 0u, {}};
    if (m.has("nexthops")) for (auto &nh : m["nexthops"].arr) w.nexthops.push_back({(int)nh[0].i(), nh[1].s});
    want.push_back(std::move(w));
  }
  const auto areas = areas_from_vector(vec);
  const std::vector<O::RibRow> before = ospf_rib_rows(vec["rib_before"]);
  std::vector<O::RibRow> other;
  for (auto &r : ospf_rib_rows(vec["rib"])) if (r.type != "intra-area") other.push_back(r);
  auto rows = O::compute_spf_intra_area(vec["router_id"].s, areas, (uint32_t)vec["max_paths"].i(), eng);
  rows.insert(rows.end(), other.begin(), other.end());
  if (!(O::update_global_rib(rows, before, ifindex) == want)) { std::fprintf(stderr, "  ospf wire: host rule differs\n"); return 0; }
  size_t nr = 0, np = 0;
  if (!(O::update_global_rib_device(vec["router_id"].s, areas, (uint32_t)vec["max_paths"].i(), eng, before, ifindex, other, &nr, &np) == want)) {
    std::fprintf(stderr, "  ospf wire: device form differs\n");
    return 0;
  }
  records += nr; prefixes += np;
  if (areas.size() > 1) ++multi_area;
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
  // further events on the SAME cache (random chains: "next" holds whole vectors): patch upon patch
  if (step.has("next")) {
    auto prev = areas1;
    for (auto &nx : step["next"].arr) {
      const auto cur = areas_from_vector(nx);
      std::map<std::string, std::vector<O::VertexId>> tr;
      for (auto &n : cur) for (auto &o : prev) if (o.area_id == n.area_id) tr[n.area_id] = O::changed_vertex_ids(o, n);
      const int b4 = cache.patched;
      if (!ospf_rows_equal(O::compute_spf_intra_area(nx["router_id"].s, cur, (uint32_t)nx["max_paths"].i(), eng, &cache, &tr), nx["rib"])) { std::fprintf(stderr, "  replay: OSPF RIB differs at a later step of the chain\n"); return 0; }
      patched += cache.patched - b4;
      for (auto &n : cur) {
        O::AreaGraph fresh(n);
        const O::AreaGraph &g = *cache.graphs.at(n.area_id);
        if (g.vids != fresh.vids || g.row_ptr != fresh.row_ptr || g.col != fresh.col || g.metric != fresh.metric || g.link_pos != fresh.link_pos) { std::fprintf(stderr, "  replay: an area graph patched several times differs from a fresh one\n"); return 0; }
      }
      prev = cur;
    }
  }
  return 1;
}

// OSPFv3 from the engine (round 5): the recorded RIB from the ordered fold of every area on the engine, and a wire step — the
// LSDB as recorded against a perturbed copy of it (remote Intra-Area-Prefix and Router-LSA metrics changed) as the RIB held
// before; no OSPFv3 STEP recording exists (the conformance module is commented out upstream; the 44 topology recordings are
// checked by check_cold_start below), so the expected sequence of this perturbed step is the host rule's (pinned by the OSPFv2
// step recordings and the 132 cold-start states: it does not look at the version).
static int check_ospfv3_device(const J &vec, Engine &eng, size_t &records, size_t &prefixes, int &with_msgs) {
  const auto areas = areas3_from_vector(vec);
  const std::string rid = vec["router_id"].s, af = vec["af"].s;
  const uint32_t mp = (uint32_t)vec["max_paths"].i();
  const auto rows = O::v3::compute_spf_intra_area(rid, areas, mp, eng, af);
  if (!(O::v3::intra_area_device_routes(rid, areas, mp, eng, af) == rows)) { std::fprintf(stderr, "  ospfv3: rows from the engine's fold differ\n"); return 0; }
  auto before_areas = areas;
  for (auto &a : before_areas) {
    size_t i = 0;
    for (auto &l : a.iaps) if (l.adv_rtr != rid && i++ % 3 == 0) for (auto &p : l.prefixes) p.metric += 3;
    i = 0;
    for (auto &l : a.routers) if (l.adv_rtr != rid && i++ % 4 == 1) for (auto &k : l.links) k.metric = k.metric % 7 + 1;
  }
  {
    // SpfComputation::{Full, Partial}: the perturbed LSDB first (Full), then the recorded one reached by a change of Intra-Area-
    // Prefix-LSAs only (their router metrics put back by a Full run in between) — the Partial run re-attaches those prefixes
    // from the STORED SPTs, gives the rows of a Full run on the same LSDB and does not call the engine
    auto mid = areas;                                            // recorded router LSAs, perturbed Intra-Area-Prefix-LSAs
    for (size_t ai = 0; ai < mid.size(); ++ai) mid[ai].iaps = before_areas[ai].iaps;
    O::v3::SpfState st(rid, mp, eng, af);
    st.run(before_areas);
    if (!(st.run(mid) == O::v3::compute_spf_intra_area(rid, mid, mp, eng, af))) { std::fprintf(stderr, "  ospfv3 SpfState: full run differs\n"); return 0; }
    std::vector<O::TriggerLsa> trig;
    for (size_t ai = 0; ai < mid.size(); ++ai)
      for (size_t li = 0; li < mid[ai].iaps.size(); ++li) {
        const auto &o = mid[ai].iaps[li], &nw = areas[ai].iaps[li];
        bool differs = o.prefixes.size() != nw.prefixes.size();
        for (size_t k = 0; !differs && k < o.prefixes.size(); ++k) differs = o.prefixes[k].metric != nw.prefixes[k].metric;
        if (!differs) continue;
        O::TriggerLsa t{"intra-area-prefix", {}, {}};
        for (auto &p : nw.prefixes) t.new_prefixes.push_back(p.prefix);
        for (auto &p : o.prefixes) t.old_prefixes.push_back(p.prefix);
        trig.push_back(std::move(t));
      }
    const int runs = st.engine_runs;
    if (!(st.run(areas, &trig) == rows) || st.engine_runs != runs) { std::fprintf(stderr, "  ospfv3 SpfState: partial run differs (or called the engine)\n"); return 0; }
    trig.push_back(O::TriggerLsa{"router", {}, {}});             // ... and a Router-LSA among the triggers makes it Full
    if (!(st.run(areas, &trig) == rows) || st.engine_runs == runs) { std::fprintf(stderr, "  ospfv3 SpfState: full dispatch differs\n"); return 0; }
  }
  std::map<std::string, int> ifindex;
  for (auto &a : areas) for (auto &f : a.interfaces) ifindex[f.name] = (int)f.index;
  const auto before = O::v3::compute_spf_intra_area(rid, before_areas, mp, eng, af);
  for (const auto *old : {&before, &rows}) {                       // a changed LSDB, and an unchanged one (nothing to send)
    const auto want = O::update_global_rib(rows, *old, ifindex);
    size_t nr = 0, np = 0;
    if (!(O::v3::update_global_rib_device(rid, areas, mp, eng, *old, ifindex, af, {}, &nr, &np) == want)) { std::fprintf(stderr, "  ospfv3 wire: device form differs\n"); return 0; }
    records += nr; prefixes += np;
    if (!want.empty()) ++with_msgs;
  }
  return 1;
}

// The recorded COLD-START wire output (tests/golden/wire/<proto>/<topo>_<rt>.json, tools/make_golden_wire.py: the final per-prefix
// state of the topology's `output/ibus.jsonl` — IS-IS 38, OSPFv2 50, OSPFv3 44 with fe80:: next hops): update_global_rib from
// an EMPTY RIB by the host rule and from ENGINE tables must install exactly those routes, in RIB key order.
// 1 equal, 0 a difference, -1 no recording for this vector.
typedef std::map<std::string, std::pair<uint32_t, std::vector<std::pair<int, std::string>>>> WireState;
static bool wire_state_of(const std::vector<IbusMsg> &msgs, WireState &out) {
  for (size_t i = 0; i < msgs.size(); ++i) {
    if (!msgs[i].add || out.count(msgs[i].prefix)) return false;
    if (i && !(parse_ip(msgs[i - 1].prefix) < parse_ip(msgs[i].prefix))) return false;         // BTreeMap<IpNetwork, _> order
    auto nh = msgs[i].nexthops; std::sort(nh.begin(), nh.end());
    out[msgs[i].prefix] = {msgs[i].metric, nh};
  }
  return true;
}
static int check_cold_start(const J &vec, const std::string &path, const std::string &golden_dir, Engine &eng, size_t &records, int &device_forms) {
  if (golden_dir.empty() || vec["source"].s.find("snapshot ") != std::string::npos) return -1;
  const std::string proto = vec["proto"].s, wp = golden_dir + "/wire/" + proto + "/" + path.substr(path.find_last_of('/') + 1);
  { FILE *f = std::fopen(wp.c_str(), "r"); if (!f) return -1; std::fclose(f); }
  const J w = load_json(wp);
  std::map<std::string, int> ifindex;
  for (auto &kv : w["ifindex"].obj) ifindex[kv.first] = (int)kv.second.i();
  WireState want;
  for (auto &r : w["final"].arr) {
    std::vector<std::pair<int, std::string>> nh;
    for (auto &n : r["nexthops"].arr) { if (n[2].arr.size()) return 0; nh.push_back({(int)n[0].i(), n[1].s}); }
    std::sort(nh.begin(), nh.end());
    want[r["prefix"].s] = {(uint32_t)r["metric"].i(), nh};
  }
  if (proto == "isis") {
    const I::Instance inst = instance_from_vector(vec);
    WireState host;
    if (!wire_state_of(I::update_global_rib(I::compute_spf(inst, eng), {}, ifindex), host) || host != want) { std::fprintf(stderr, "  cold start: host rule differs\n"); return 0; }
    if (inst.config.sr_enabled) return 1;
    WireState dev;
    if (!wire_state_of(I::update_global_rib(I::compute_spf_device_routes(inst, eng), {}, ifindex), dev) || dev != want) { std::fprintf(stderr, "  cold start: rows attached on the engine differ\n"); return 0; }
    std::vector<std::pair<int, int>> tabs;
    for (int lv : inst.config.levels()) for (int mt : {I::MT_STANDARD, I::MT_IPV6_UNICAST}) if (inst.config.is_topology_enabled(mt)) tabs.push_back({lv, mt});
    if (tabs.size() != 1) return 1;
    size_t nr = 0, np = 0;
    WireState pk;
    if (!wire_state_of(I::update_global_rib_device(inst, eng, {}, ifindex, &nr, &np), pk) || pk != want || nr > np) { std::fprintf(stderr, "  cold start: device form differs\n"); return 0; }
    records += nr; ++device_forms;
    return 1;
  }
  // OSPFv2 / OSPFv3: the intra-area part is this path's, the other route types come from the recording; at a virtual-link endpoint
  // the intra-area routes THROUGH the link get their next hops from the transit-area step (outside the path): left out
  const bool v3 = proto == "ospfv3";
  const std::string rid = vec["router_id"].s;
  const uint32_t mp = (uint32_t)vec["max_paths"].i();
  std::vector<O::RibRow> rows, other;
  std::vector<IbusMsg> devmsgs;
  size_t nr = 0, np = 0;
  for (auto &r : ospf_rib_rows(vec["rib"])) if (r.type != "intra-area") other.push_back(r);
  if (v3) {
    const auto areas = areas3_from_vector(vec);
    rows = O::v3::compute_spf_intra_area(rid, areas, mp, eng, vec["af"].s);
    devmsgs = O::v3::update_global_rib_device(rid, areas, mp, eng, {}, ifindex, vec["af"].s, other, &nr, &np);
  } else {
    const auto areas = areas_from_vector(vec);
    rows = O::compute_spf_intra_area(rid, areas, mp, eng);
    devmsgs = O::update_global_rib_device(rid, areas, mp, eng, {}, ifindex, other, &nr, &np);
  }
  std::set<std::string> undecided;
  if (vec["has_vlinks"].b) for (auto &r : rows) if (r.nexthops.empty()) undecided.insert(r.prefix);
  for (auto &u : undecided) want.erase(u);
  rows.insert(rows.end(), other.begin(), other.end());
  WireState host, dev;
  if (!wire_state_of(O::update_global_rib(rows, {}, ifindex), host)) return 0;
  if (!wire_state_of(devmsgs, dev)) return 0;
  for (auto &u : undecided) { host.erase(u); dev.erase(u); }
  if (host != want) { std::fprintf(stderr, "  cold start: host rule differs\n"); return 0; }
  if (dev != want) { std::fprintf(stderr, "  cold start: device form differs\n"); return 0; }
  records += nr; ++device_forms;
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
  int wire_ok = 0, wire_bad = 0, wire_pipelines = 0, owire_ok = 0, owire_bad = 0, owire_multi = 0;
  size_t owire_records = 0, owire_prefixes = 0, v3_records = 0, v3_prefixes = 0;
  int v3_ok = 0, v3_bad = 0, v3_msgs = 0;
  size_t wire_records = 0, wire_prefixes = 0, cold_records = 0;
  int cold_ok = 0, cold_bad = 0, cold_dev = 0;
  if (!check_hash_kat()) { std::fprintf(stderr, "flood_reduction_hash: reference vectors not reproduced\n"); return 1; }
  if (!check_lsdb_cursor()) { std::fprintf(stderr, "Lsdb::fragments_from differs from iter_for_lan_id / zeroth_lsp\n"); return 1; }
  for (auto &path : files) {
    try {
      const J vec = load_json(path);
      {
        const int c = check_cold_start(vec, path, golden_dir, *eng, cold_records, cold_dev);
        if (c > 0) ++cold_ok; else if (c == 0) { ++cold_bad; std::fprintf(stderr, "COLD START WIRE STATE MISMATCH %s\n", path.c_str()); }
      }
      if (vec["proto"].s == "ospfv2") {
        if (!golden_dir.empty() && vec["source"].s.find("snapshot ") != std::string::npos) {
          const int sr = replay_ospf_step(vec, golden_dir, *eng, steps_patched);
          if (sr > 0) ++steps_ok; else if (sr == 0) { ++steps_bad; std::fprintf(stderr, "STEP REPLAY MISMATCH %s\n", path.c_str()); }
        }
        {
          const int w = check_ospf_wire(vec, *eng, owire_records, owire_prefixes, owire_multi);
          if (w > 0) ++owire_ok; else if (w == 0) { ++owire_bad; std::fprintf(stderr, "OSPF WIRE STEP MISMATCH %s\n", path.c_str()); }
        }
        const int r = check_ospf(vec, *eng, path);
        if (r > 0) ++ok; else if (r == 0) ++bad; else ++skipped;
        if (r > 0) {                                              // SpfComputation dispatch, OSPFv2: Type-3 / 4 / 5 changes leave the intra-area part alone
          const auto areas = areas_from_vector(vec);
          O::SpfState st(vec["router_id"].s, (uint32_t)vec["max_paths"].i(), *eng);
          const auto r0 = st.run(areas);
          const int runs = st.engine_runs;
          std::vector<O::TriggerLsa> t{O::TriggerLsa{"summary-network", {}, {}}, O::TriggerLsa{"as-external", {}, {}}};
          const bool same = st.run(areas, &t) == r0 && st.engine_runs == runs;
          t.push_back(O::TriggerLsa{"network", {}, {}});
          const bool full = st.run(areas, &t) == r0 && st.engine_runs > runs;
          if (!same || !full) { ++bad; std::fprintf(stderr, "SPF COMPUTATION DISPATCH MISMATCH %s\n", path.c_str()); }
          // the root's Router-LSA goes missing: every area keeps the SPT it had and update_rib_full still folds it (route.rs:157-160)
          auto gone = areas;
          for (auto &a : gone) a.routers.erase(std::remove_if(a.routers.begin(), a.routers.end(), [&](const O::RouterLsa &l) { return l.adv_rtr == vec["router_id"].s; }), a.routers.end());
          if (!(st.run(gone) == r0)) { ++bad; std::fprintf(stderr, "ROOT LSA MISSING: THE STORED SPT'S ROUTES DID NOT STAY %s\n", path.c_str()); }
        }
        continue;
      }
      if (vec["proto"].s == "ospfv3") {
        if (vec["has_vlinks"].b) { ++skipped; continue; }
        const auto rows3 = O::v3::compute_spf_intra_area(vec["router_id"].s, areas3_from_vector(vec), (uint32_t)vec["max_paths"].i(), *eng, vec["af"].s);
        if (ospf_rows_equal(rows3, vec["rib"])) ++ok; else { ++bad; std::fprintf(stderr, "MISMATCH %s (ospfv3)\n", path.c_str()); }
        if (check_ospfv3_device(vec, *eng, v3_records, v3_prefixes, v3_msgs) > 0) ++v3_ok; else { ++v3_bad; std::fprintf(stderr, "OSPFV3 DEVICE ROUTES / WIRE STEP MISMATCH %s\n", path.c_str()); }
        continue;
      }
      if (vec["proto"].s != "isis") continue;
      if (!golden_dir.empty() && vec["source"].s.find("snapshot ") != std::string::npos) {
        const int r = replay_isis_step(vec, golden_dir, *eng, steps_patched);
        if (r > 0) ++steps_ok; else if (r == 0) { ++steps_bad; std::fprintf(stderr, "STEP REPLAY MISMATCH %s\n", path.c_str()); }
      }
      {
        const int w = check_isis_wire(vec, golden_dir, *eng, wire_records, wire_prefixes, wire_pipelines);
        if (w > 0) ++wire_ok; else if (w == 0) { ++wire_bad; std::fprintf(stderr, "WIRE STEP MISMATCH %s\n", path.c_str()); }
      }
      const I::Instance inst = instance_from_vector(vec);
      if (vec.has("manet")) { const int mb = check_manet(vec, inst, *eng, path); manet_cases += (int)vec["manet"].size(); manet_bad += mb; }
      const auto rows = I::compute_spf(inst, *eng);
      // the same RIB with the prefix attachment done by the engine (hspf_run_device + hspf_routes_device)
      // (instances with segment routing keep the host path: the Prefix-SID step needs the Route objects)
      if (!inst.config.sr_enabled) { if (rows_equal(I::compute_spf_device_routes(inst, *eng), vec["rib"])) ++dev_ok; else { ++dev_bad; std::fprintf(stderr, "DEVICE ROUTES MISMATCH %s\n", path.c_str()); } }
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
      if (same && !rows_equal(rows, vec["rib"])) { same = false; std::fprintf(stderr, "MISMATCH %s: SR labels\n", path.c_str()); }
      if (same) ++ok; else { ++bad; if (want.size() != rows.size()) std::fprintf(stderr, "MISMATCH %s: %zu rows, recorded %zu\n", path.c_str(), rows.size(), want.size()); }
    } catch (const std::exception &e) { ++bad; std::fprintf(stderr, "ERROR %s: %s\n", path.c_str(), e.what()); }
  }
  std::printf("host_parity (%s engine): %d vectors reproduce the recorded local RIB, %d do not, %d skipped (virtual links)\n", engine.c_str(), ok, bad, skipped);
  if (manet_cases) std::printf("host_parity: %d reflood lists checked, %d differ\n", manet_cases, manet_bad);
  if (steps_ok + steps_bad) std::printf("host_parity: %d step tests replayed through patched graphs (%d row-patch refreshes), %d differ\n", steps_ok + steps_bad, steps_patched, steps_bad);
  if (dev_ok + dev_bad) std::printf("host_parity: %d IS-IS RIBs also derived with the prefix attachment on the engine, %d differ\n", dev_ok + dev_bad, dev_bad);
  if (wire_ok + wire_bad) std::printf("host_parity: %d recorded ibus sequences (RouteIpAdd / RouteIpDel) reproduced by the host rule AND from engine tables (%zu records for %zu prefixes), %d differ; %d also through the running-instance pipeline\n", wire_ok, wire_records, wire_prefixes, wire_bad, wire_pipelines);
  if (owire_ok + owire_bad) std::printf("host_parity: %d recorded OSPFv2 ibus sequences reproduced by the host rule AND from engine tables (%zu records for %zu prefixes; %d of them two-area instances folded into one RIB on the engine), %d differ\n", owire_ok, owire_records, owire_prefixes, owire_multi, owire_bad);
  if (v3_ok + v3_bad) std::printf("host_parity: %d OSPFv3 RIBs also from the ordered fold on the engine, each with two wire steps against the host rule (%zu records for %zu prefixes, %d sequences with messages) and a Full / Partial / Full sequence of SpfState, %d differ\n", v3_ok, v3_records, v3_prefixes, v3_msgs, v3_bad);
  if (cold_ok + cold_bad) std::printf("host_parity: %d recorded cold-start ibus states (topology output/ibus.jsonl; OSPFv3 included) reproduced by the host rule AND from engine tables (%d through the device comparison and packing, %zu records), %d differ\n", cold_ok, cold_dev, cold_records, cold_bad);
  return (bad || manet_bad || steps_bad || dev_bad || wire_bad || owire_bad || v3_bad || cold_bad) ? 1 : 0;
}
