// holo_spf_isis.hpp — C++17 host side of the IS-IS SPF path on top of the C ABI (include/holo_spf_hip.h).
//
// The compiled-code twin of what a maintainer's Rust patch does around the engine, function for
// function, with the reference's names, argument meaning and failure behaviour:
//
//   vertex_edges()       holo-isis/src/spf.rs:1013-1146   edges of one LSP fragment, TLV order, metric mode
//   LevelGraph           (new)                            LSDB of one level/topology -> hspf_csr, once per generation
//   resolve_nexthop()    holo-isis/src/spf.rs:956-1010    unchanged: first matching interface in name order, used_adjs
//   compute_spt(s)()     holo-isis/src/spf.rs:527-709     SPT loop on the device; first-hop slots replayed in reference
//                        flooding/manet.rs:47-69          order through resolve_nexthop; batched roots = one run
//   vertex_networks()    holo-isis/src/spf.rs:1149-1296   prefixes of one SPT vertex
//   compute_routes()     holo-isis/src/spf.rs:840-949     prefix attachment, ECMP merge, max-paths truncation
//   compute_spf()        holo-isis/src/spf.rs:719-836     per level / topology + L1/L2 merge (route.rs:185-249)
//   flooding::init_cache / reflood_list / flood_reduction_hash   holo-isis/src/flooding/manet.rs:39-194
//
// The engine is reached through hspf::host::Engine (three calls: upload, run, slot_table); the product implementation
// is hspf::host::HipEngine (holo_spf_host.hpp, the C ABI).  There is no CPU SPT loop in this file.  Python twin with the same structure:
// holo_amd/isis.py.  Tests: tests/cpp/host_parity.cpp (recorded RIBs of the reference's conformance fixtures).
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdint>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "holo_spf_host.hpp"

namespace hspf {
namespace host {
namespace isis {

constexpr uint32_t MAX_PATH_METRIC_STANDARD = 1023;          // holo-isis/src/spf.rs:45
constexpr uint32_t MAX_PATH_METRIC_WIDE = 0xFE000000u;       // holo-isis/src/spf.rs:47
constexpr uint32_t MAX_LINK_METRIC_WIDE = 0x00FFFFFFu;       // holo-isis/src/spf.rs:49
constexpr int MT_STANDARD = 0, MT_IPV6_UNICAST = 2;
constexpr int NLPID_IPV4 = 0xCC, NLPID_IPV6 = 0x8E;

using SystemId = std::array<uint8_t, 6>;
struct LanId {
  SystemId system_id{};
  uint8_t pseudonode = 0;
  bool operator<(const LanId &o) const { return std::tie(system_id, pseudonode) < std::tie(o.system_id, o.pseudonode); }
  bool operator==(const LanId &o) const { return system_id == o.system_id && pseudonode == o.pseudonode; }
};
struct VertexId {                     // holo-isis/src/spf.rs:96-100: derive(Ord) => pseudonodes first
  bool non_pseudonode = true;
  LanId lan_id;
  bool operator<(const VertexId &o) const { return std::tie(non_pseudonode, lan_id) < std::tie(o.non_pseudonode, o.lan_id); }
  bool operator==(const VertexId &o) const { return non_pseudonode == o.non_pseudonode && lan_id == o.lan_id; }
};
inline VertexId vertex_id(const LanId &l) { return VertexId{l.pseudonode == 0, l}; }

// ---- the fields of Lsp / LspTlvs / Interface / Adjacency the path reads -----------------------------------------
// Segment routing (compute_routes' Prefix-SID step, holo-isis/src/spf.rs:931-946 -> sr.rs:34-94, 165-300): the SR-Capabilities
// and SR-Algorithm sub-TLVs of a router, the Prefix-SID sub-TLV of a reachability entry.
struct PrefixSid {
  std::vector<std::string> flags;          // "P", "E", "V", "L", ...
  std::optional<uint32_t> index, label;    // one of them (the V / L flags say which the LSP carried)
  bool has(const char *f) const { return std::find(flags.begin(), flags.end(), f) != flags.end(); }
  bool operator==(const PrefixSid &o) const { return flags == o.flags && index == o.index && label == o.label; }
};
struct SrCap {
  std::vector<std::string> flags;          // "I" (MPLS IPv4), "V" (MPLS IPv6)
  std::vector<std::pair<uint32_t, uint32_t>> srgb;   // (first label, range)
  bool has(const char *f) const { return std::find(flags.begin(), flags.end(), f) != flags.end(); }
  bool operator==(const SrCap &o) const { return flags == o.flags && srgb == o.srgb; }
};
struct Lsp {
  SystemId system_id{};
  uint8_t pseudonode = 0, fragment = 0;
  uint32_t seqno = 1;
  uint16_t rem_lifetime = 1200;
  bool overload = false, att = false;
  std::optional<std::vector<int>> protocols_supported;
  std::map<int, std::pair<bool, bool>> mt_flags;                               // mt -> (overload, att)
  std::vector<std::pair<LanId, uint32_t>> is_reach, ext_is_reach;              // TLV 2, TLV 22
  std::vector<std::tuple<int, LanId, uint32_t>> mt_is_reach;                   // TLV 222
  std::vector<std::pair<std::string, uint32_t>> ipv4_internal, ipv4_external;  // TLV 128, 130
  std::vector<std::tuple<std::string, uint32_t, bool>> ext_ipv4, ipv6;         // TLV 135 (+X), 236
  std::vector<std::tuple<int, std::string, uint32_t, bool>> mt_ipv6;           // TLV 237
  std::optional<SrCap> sr_cap;
  std::vector<int> sr_algos;                                                   // 0 = SPF
  std::map<std::string, std::map<int, PrefixSid>> prefix_sids;                 // "ext_ipv4" | "ipv6" | "mt_ipv6" -> entry index -> SID
  const PrefixSid *prefix_sid(const char *kind, int i) const {
    auto k = prefix_sids.find(kind);
    if (k == prefix_sids.end()) return nullptr;
    auto it = k->second.find(i);
    return it == k->second.end() ? nullptr : &it->second;
  }
  LanId lan_id() const { return LanId{system_id, pseudonode}; }
  bool live() const { return seqno != 0 && rem_lifetime != 0; }               // spf.rs:1024-1025
  bool overload_bit(int mt) const { auto it = mt_flags.find(mt); return mt == MT_STANDARD ? overload : (it != mt_flags.end() && it->second.first); }
  bool att_bit(int mt) const { auto it = mt_flags.find(mt); return mt == MT_STANDARD ? att : (it != mt_flags.end() && it->second.second); }
  bool operator==(const Lsp &o) const {
    return system_id == o.system_id && pseudonode == o.pseudonode && fragment == o.fragment && seqno == o.seqno &&
           rem_lifetime == o.rem_lifetime && overload == o.overload && att == o.att && protocols_supported == o.protocols_supported &&
           mt_flags == o.mt_flags && is_reach == o.is_reach && ext_is_reach == o.ext_is_reach && mt_is_reach == o.mt_is_reach &&
           ipv4_internal == o.ipv4_internal && ipv4_external == o.ipv4_external && ext_ipv4 == o.ext_ipv4 && ipv6 == o.ipv6 && mt_ipv6 == o.mt_ipv6 &&
           sr_cap == o.sr_cap && sr_algos == o.sr_algos && prefix_sids == o.prefix_sids;
  }
};
struct Adjacency {
  SystemId system_id{};
  std::string level_usage;             // "level-1" | "level-2" | "level-all"
  std::string state = "up";
  std::vector<std::string> ipv4_addrs, ipv6_addrs, area_addrs;
  std::vector<int> topologies{0};
  std::string snpa;                    // anything unique per adjacency
  bool intersects(int level) const { return level_usage == "level-all" || level_usage == "level-" + std::to_string(level); }
  bool in_topology(int mt) const { return std::find(topologies.begin(), topologies.end(), mt) != topologies.end(); }
};
struct Interface {
  std::string name, interface_type = "broadcast";     // "broadcast" | "point-to-point"
  std::map<int, uint32_t> metric{{1, 10}, {2, 10}};
  std::vector<Adjacency> adjacencies;
};
struct InstanceCfg {
  SystemId system_id{};
  std::string level_type = "level-all";
  std::map<int, std::string> metric_type{{1, "wide"}, {2, "wide"}};
  bool ipv4_enabled = true, ipv6_enabled = true, mt_ipv6_unicast = false, att_ignore = false;
  bool sr_enabled = false;
  uint32_t max_paths = 16;
  std::vector<std::string> area_addrs;
  bool is_af_enabled(bool v6) const { return v6 ? ipv6_enabled : ipv4_enabled; }
  bool is_topology_enabled(int mt) const { return mt == MT_STANDARD ? true : mt_ipv6_unicast; }
  std::vector<int> levels() const {
    if (level_type == "level-1") return {1};
    if (level_type == "level-2") return {2};
    return {1, 2};
  }
};
class Lsdb {                           // LSPs of one level ordered by LSP id (holo-isis/src/collections.rs:657-706)
 public:
  using Key = std::tuple<SystemId, uint8_t, uint8_t>;
  void insert(Lsp l) { Key k{l.system_id, l.pseudonode, l.fragment}; by_id_[k] = std::move(l); }
  const std::map<Key, Lsp> &all() const { return by_id_; }
  std::vector<const Lsp *> iter_for_lan_id(const LanId &lan) const {
    std::vector<const Lsp *> out;
    for (auto it = by_id_.lower_bound(Key{lan.system_id, lan.pseudonode, 0}); it != by_id_.end(); ++it) {
      if (!(std::get<0>(it->first) == lan.system_id) || std::get<1>(it->first) != lan.pseudonode) break;
      out.push_back(&it->second);
    }
    return out;
  }
  std::vector<const Lsp *> iter_for_system_id(const SystemId &sid) const {      // every LSP of a system, pseudonodes included, in LSP-id order
    std::vector<const Lsp *> out;
    for (auto it = by_id_.lower_bound(Key{sid, 0, 0}); it != by_id_.end() && std::get<0>(it->first) == sid; ++it) out.push_back(&it->second);
    return out;
  }
  const Lsp *zeroth_lsp(const LanId &lan) const {                              // spf.rs:1299-1309
    auto it = by_id_.find(Key{lan.system_id, lan.pseudonode, 0});
    return (it != by_id_.end() && it->second.live()) ? &it->second : nullptr;
  }
  // The fragments of `lan` for a caller that visits LAN ids in ASCENDING order (a walk over the vertices of a level graph, per
  // class of vertex): a few steps forward from where the previous visit stood instead of a descent of the tree (two descents
  // per vertex were 40 of the 48 ms a prefix table of 100 000 vertices took).  Any order is answered correctly — a visit that
  // does not lie a few LSPs ahead of the cursor descends.  `out`: every fragment, in LSP-id order, live or not (as
  // iter_for_lan_id); returns the zeroth LSP if it is live (as zeroth_lsp).
  struct Cursor { std::map<Key, Lsp>::const_iterator at; bool set = false; };
  const Lsp *fragments_from(Cursor &c, const LanId &lan, std::vector<const Lsp *> &out) const {
    const Key k{lan.system_id, lan.pseudonode, 0};
    auto it = by_id_.end();
    bool found = false;
    if (c.set && c.at != by_id_.end() && c.at->first < k) {
      it = c.at;
      for (int s = 0; s < 8 && !found; ++s) { ++it; found = it == by_id_.end() || !(it->first < k); }
    }
    if (!found) it = by_id_.lower_bound(k);
    c.at = it; c.set = true;
    out.clear();
    const Lsp *zeroth = nullptr;
    for (; it != by_id_.end(); ++it) {
      if (!(std::get<0>(it->first) == lan.system_id) || std::get<1>(it->first) != lan.pseudonode) break;
      if (std::get<2>(it->first) == 0 && it->second.live()) zeroth = &it->second;
      out.push_back(&it->second);
    }
    return zeroth;
  }
 private:
  std::map<Key, Lsp> by_id_;
};
struct Instance {                      // the slice of InstanceUpView the path reads
  InstanceCfg config;
  std::vector<Interface> interfaces;
  std::map<int, Lsdb> lsdb;
  std::vector<const Interface *> interfaces_by_name() const {                  // collections.rs:258-265
    std::vector<const Interface *> v;
    for (auto &i : interfaces) v.push_back(&i);
    std::sort(v.begin(), v.end(), [](const Interface *a, const Interface *b) { return a->name < b->name; });
    return v;
  }
  bool is_l2_attached_to_backbone(int mt) const {                              // holo-isis/src/instance.rs:577-591
    for (const Interface *i : interfaces_by_name())
      for (const Adjacency &a : i->adjacencies) {
        if (!a.in_topology(mt) || a.state != "up" || !a.intersects(2)) continue;
        bool disjoint = true;
        for (auto &x : a.area_addrs)
          if (std::find(config.area_addrs.begin(), config.area_addrs.end(), x) != config.area_addrs.end()) disjoint = false;
        if (disjoint) return true;
      }
    return false;
  }
};

// ---- LSDB -> CSR -------------------------------------------------------------------------------------------------
// Edges one live fragment contributes, in the reference's order (spf.rs:1026-1127); HopCount mode: cost 0 to a
// pseudonode, 1 to a router (:1131-1146).
template <class F>
inline void for_each_vertex_edge(const Lsp &lsp, std::optional<int> mt_id, bool hopcount, bool std_on, bool wide_on, F &&emit) {
  auto cost = [&](const LanId &nbr, uint32_t m) { return !hopcount ? m : (nbr.pseudonode != 0 ? 0u : 1u); };
  const bool none_or_std = !mt_id || *mt_id == MT_STANDARD;
  if (none_or_std && std_on)
    for (auto &e : lsp.is_reach) emit(e.first, cost(e.first, e.second));
  if ((none_or_std || lsp.pseudonode != 0) && wide_on)
    for (auto &e : lsp.ext_is_reach)
      if (e.second < MAX_LINK_METRIC_WIDE) emit(e.first, cost(e.first, e.second));
  if (mt_id && *mt_id != MT_STANDARD)
    for (auto &e : lsp.mt_is_reach)
      if (std::get<0>(e) == *mt_id && std::get<2>(e) < MAX_LINK_METRIC_WIDE) emit(std::get<1>(e), cost(std::get<1>(e), std::get<2>(e)));
  if (!mt_id)
    for (auto &e : lsp.mt_is_reach)
      if (std::get<2>(e) < MAX_LINK_METRIC_WIDE) emit(std::get<1>(e), cost(std::get<1>(e), std::get<2>(e)));
}
inline std::vector<std::pair<LanId, uint32_t>> vertex_edges(const Lsp &lsp, std::optional<int> mt_id, bool hopcount,
                                                            const std::string &metric_type) {
  std::vector<std::pair<LanId, uint32_t>> out;
  for_each_vertex_edge(lsp, mt_id, hopcount, metric_type == "standard" || metric_type == "both", metric_type == "wide" || metric_type == "both",
                       [&](const LanId &nbr, uint32_t c) { out.push_back({nbr, c}); });
  return out;
}

// CSR form (include/holo_spf_hip.h) of one level's LSDB for one (mt_id, metric mode).  Vertex index = rank in VertexId
// order over the LAN ids that own at least one live fragment; links to LAN ids without any LSP are not listed (they
// can never pass the two-way check).
// VertexId -> vertex index: a binary search in the graph's sorted vertex list with the slice of std::map's interface the callers
// use (find / end / ->second).  (Round 6: the std::map it replaces cost 100 000 node allocations per first extraction.)
class VidIndex {
 public:
  struct It {
    uint32_t second = 0; bool ok = false;
    const It *operator->() const { return this; }
    bool operator==(const It &o) const { return ok == o.ok && (!ok || second == o.second); }
    bool operator!=(const It &o) const { return !(*this == o); }
  };
  explicit VidIndex(const std::vector<VertexId> &v) : v_(&v) {}
  It find(const VertexId &x) const {
    auto it = std::lower_bound(v_->begin(), v_->end(), x);
    return (it != v_->end() && *it == x) ? It{(uint32_t)(it - v_->begin()), true} : It{};
  }
  It end() const { return It{}; }
  size_t count(const VertexId &x) const { return find(x).ok ? 1 : 0; }
 private:
  const std::vector<VertexId> *v_;
};

class LevelGraph {
 public:
  LevelGraph(const LevelGraph &) = delete;
  LevelGraph &operator=(const LevelGraph &) = delete;
  int level;
  std::optional<int> mt_id;
  bool hopcount;
  std::string metric_type;
  std::vector<VertexId> vids;
  VidIndex index{vids};
  std::vector<uint32_t> row_ptr, col, metric;
  std::vector<uint8_t> vflags;
  uint32_t max_path_metric, run_flags;

  // `keyed`: an engine that turns LSDB records into the CSR itself (Engine::upload_keyed = hspf_graph_upload_keyed).  The walk
  // below then only STREAMS the records — per vertex its key (!pseudonode << 56 | LAN id: ascending = VertexId order,
  // holo-isis/src/spf.rs:96-100) and its links as (target key, cost) — and the million target look-ups, the ranking of the
  // vertices and the dropping of links to absent LSPs happen on the device (round 6; 148 ms -> see profiles/r06_notes.md).
  LevelGraph(const Instance &inst, int level_, std::optional<int> mt, bool hop = false, Engine *keyed = nullptr)
      : level(level_), mt_id(mt), hopcount(hop) {
    const InstanceCfg &cfg = inst.config;
    const Lsdb &lsdb = lsdb_of(inst);
    metric_type = cfg.metric_type.at(level);
    cfg_key_ = cfg_key(cfg);
    max_path_metric = metric_type == "standard" ? MAX_PATH_METRIC_STANDARD : MAX_PATH_METRIC_WIDE;   // spf.rs:637-641
    run_flags = mt_id ? 0u : (uint32_t)HSPF_RUN_IGNORE_OVERLOAD;                                     // spf.rs:566-574
    if (keyed && build_keyed(lsdb, cfg, *keyed)) return;
    auto frags = live_fragments(lsdb);
    for (auto &kv : frags) vids.push_back(vertex_id(kv.first));
    std::sort(vids.begin(), vids.end());
    row_ptr.assign(vids.size() + 1, 0);
    vflags.assign(vids.size(), 0);
    for (uint32_t i = 0; i < vids.size(); ++i) {
      auto r = row(vids[i].lan_id, frags, lsdb, cfg);
      col.insert(col.end(), r.col.begin(), r.col.end());
      metric.insert(metric.end(), r.metric.begin(), r.metric.end());
      vflags[i] = r.flags;
      row_ptr[i + 1] = (uint32_t)col.size();
    }
  }
  // Incremental re-derivation after the LSPs of `changed` LAN ids were re-originated, purged or aged out (the reference's
  // `trigger_lsps`, holo-isis/src/spf.rs:144,735): only their rows are rebuilt and — when the graph is on the device —
  // replaced there with hspf_graph_patch.  false (nothing touched) when the change is not a set of row replacements (a
  // vertex appeared or vanished, or the configuration the rows depend on changed): the caller builds a new LevelGraph.
  // Host work is O(changed LSPs x log N) (+ one rebuild of the host CSR arrays when a row changes length): only the fragments of
  // the changed LAN ids are looked at — the vertex set can only change through one of THEM gaining its first or losing its
  // last live fragment (round 5; until then every refresh re-walked the whole LSDB: 55 ms at 100 000 LSPs for a one-LSP change).
  bool refresh(const Instance &inst, const std::vector<LanId> &changed) {
    const InstanceCfg &cfg = inst.config;
    const Lsdb &lsdb = lsdb_of(inst);
    if (cfg_key(cfg) != cfg_key_) return false;
    std::map<LanId, std::vector<const Lsp *>> frags;
    std::set<uint32_t> vs;
    for (auto &lan : changed) {
      std::vector<const Lsp *> live;
      for (const Lsp *l : lsdb.iter_for_lan_id(lan)) if (l->live()) live.push_back(l);
      auto it = index.find(vertex_id(lan));
      if (live.empty() != (it == index.end())) return false;        // a vertex appeared or vanished
      if (it == index.end()) continue;
      vs.insert(it->second);
      frags[lan] = std::move(live);
    }
    if (vs.empty()) return true;
    std::vector<uint32_t> vv(vs.begin(), vs.end());
    std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> rows;
    std::vector<uint8_t> fl;
    bool same_len = true;
    for (uint32_t i : vv) {
      auto r = row(vids[i].lan_id, frags, lsdb, cfg);
      same_len = same_len && r.col.size() == row_ptr[i + 1] - row_ptr[i];
      rows.push_back({r.col, r.metric}); fl.push_back(r.flags);
    }
    if (dev_) dev_engine_->patch(*dev_, vv, rows, fl);
    if (same_len) {                                                   // rows keep their lengths: written in place
      for (size_t j = 0; j < vv.size(); ++j) {
        std::copy(rows[j].first.begin(), rows[j].first.end(), col.begin() + row_ptr[vv[j]]);
        std::copy(rows[j].second.begin(), rows[j].second.end(), metric.begin() + row_ptr[vv[j]]);
        vflags[vv[j]] = fl[j];
      }
    } else splice_rows(row_ptr, col, metric, vflags, vv, rows, fl);
    return true;
  }
  uint32_t n() const { return (uint32_t)vids.size(); }
  Graph &device(Engine &e) {
    if (!dev_ || dev_engine_ != &e) { dev_ = e.upload(row_ptr, col, metric, vflags, max_path_metric); dev_engine_ = &e; }
    return *dev_;
  }
  bool links_back(uint32_t t, uint32_t v) const {
    for (uint32_t k = row_ptr[t]; k < row_ptr[t + 1]; ++k)
      if (col[k] == v) return true;
    return false;
  }
 private:
  static uint64_t key_of(const LanId &lan) {
    uint64_t k = lan.pseudonode == 0 ? 1ull << 56 : 0ull;
    for (int i = 0; i < 6; ++i) k |= (uint64_t)lan.system_id[i] << (48 - 8 * i);
    return k | lan.pseudonode;
  }
  // One pass over the LSDB in ITS order (LspId: fragments of a LAN id are adjacent): records out, CSR back.
  // The pass is bound by the latency of walking the map (a node, its LSP, its TLV vectors: ~3 cache misses per LSP, 23.5 ms for
  // 100 000 LSPs on one core: profiles/r06_notes.md), not by work — so large LSDBs are cut into ranges of whole system ids
  // (splitter keys interpolated between the first and the last system id; any split is correct, an uneven one only slower) and
  // the ranges are streamed by a few threads side by side, each into its own buffers, which are then laid end to end.
  struct KeyedPart {
    std::vector<uint64_t> vkey, tkey;
    std::vector<uint32_t> vrow_len, tmet;        // vrow_len: links per vertex
    std::vector<uint8_t> vfl;
    std::vector<LanId> lans;
  };
  template <typename It>
  void stream_range(It it, It end, const InstanceCfg &cfg, bool std_on, bool wide_on, KeyedPart &o) const {
    const Lsp *zeroth = nullptr;
    size_t row_start = 0;
    auto close = [&]() {                                       // flags of the vertex whose fragments just ended (spf.rs:557-604)
      if (o.lans.empty() || o.vfl.size() == o.lans.size()) return;
      const LanId &lan = o.lans.back();
      const bool is_pn = lan.pseudonode != 0;
      uint8_t f = is_pn ? HSPF_VF_NETWORK : 0;
      const Lsp *z = zeroth;
      if (!z) f |= HSPF_VF_NO_EXPAND;
      else {
        if (!is_pn && mt_id && z->overload_bit(*mt_id)) f |= HSPF_VF_NO_TRANSIT;
        if (mt_id && *mt_id == MT_STANDARD && !is_pn) {
          auto has = [&](int p) { return z->protocols_supported && std::find(z->protocols_supported->begin(), z->protocols_supported->end(), p) != z->protocols_supported->end(); };
          if (!z->protocols_supported || (cfg.ipv4_enabled && !has(NLPID_IPV4)) || (cfg.ipv6_enabled && !has(NLPID_IPV6))) f |= HSPF_VF_NO_EXPAND;
        }
      }
      o.vfl.push_back(f);
      o.vrow_len.push_back((uint32_t)(o.tkey.size() - row_start));
      row_start = o.tkey.size();
    };
    // fragments of a LAN id are adjacent, fragment 0 — the zeroth LSP of spf.rs:1299-1309 when it is live — first
    bool have_lan = false, started = false;
    LanId cur{};
    for (; it != end; ++it) {
      const Lsp &l = it->second;
      const LanId lan = l.lan_id();
      if (!have_lan || !(cur == lan)) { if (started) close(); have_lan = true; cur = lan; started = false; zeroth = nullptr; }
      if (!l.live()) continue;
      if (l.fragment == 0) zeroth = &l;
      if (!started) { started = true; o.lans.push_back(lan); o.vkey.push_back(key_of(lan)); }
      for_each_vertex_edge(l, mt_id, hopcount, std_on, wide_on, [&](const LanId &nbr, uint32_t c) { o.tkey.push_back(key_of(nbr)); o.tmet.push_back(c); });
    }
    if (started) close();
  }
  static uint64_t sid_num(const SystemId &s) { uint64_t x = 0; for (int i = 0; i < 6; ++i) x = (x << 8) | s[i]; return x; }
  static SystemId num_sid(uint64_t x) { SystemId s{}; for (int i = 5; i >= 0; --i) { s[i] = (uint8_t)(x & 0xFF); x >>= 8; } return s; }
  bool build_keyed(const Lsdb &lsdb, const InstanceCfg &cfg, Engine &engine) {
    const auto t_begin = std::chrono::steady_clock::now();
    const bool std_on = metric_type == "standard" || metric_type == "both", wide_on = metric_type == "wide" || metric_type == "both";
    const auto &all = lsdb.all();
    if (all.empty()) return false;
    // ranges of whole system ids, one thread each (HSPF_KEYED_THREADS; default: the cores, at most 16; small LSDBs: one)
    unsigned T = 1;
    if (all.size() >= 4096) {
      T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
      if (const char *e = getenv("HSPF_KEYED_THREADS")) T = std::max(1, atoi(e));
    }
    std::vector<decltype(all.begin())> cut;
    cut.push_back(all.begin());
    if (T > 1) {
      const uint64_t lo = sid_num(std::get<0>(all.begin()->first)), hi = sid_num(std::get<0>(all.rbegin()->first));
      for (unsigned k = 1; k < T; ++k) {
        const uint64_t at = lo + (uint64_t)((unsigned __int128)(hi - lo) * k / T);
        auto it = all.lower_bound(Lsdb::Key{num_sid(at), 0, 0});            // the first LSP of a system id: never inside a LAN id's fragments
        if (it != cut.back() && it != all.end()) cut.push_back(it);
      }
    }
    cut.push_back(all.end());
    const size_t np = cut.size() - 1;
    std::vector<KeyedPart> part(np);
    if (np == 1) stream_range(cut[0], cut[1], cfg, std_on, wide_on, part[0]);
    else {
      std::vector<std::thread> th;
      for (size_t k = 0; k < np; ++k) th.emplace_back([&, k]() { stream_range(cut[k], cut[k + 1], cfg, std_on, wide_on, part[k]); });
      for (auto &t : th) t.join();
    }
    // end to end
    size_t nv = 0, nl = 0;
    std::vector<size_t> v0(np + 1, 0), l0(np + 1, 0);
    for (size_t k = 0; k < np; ++k) { v0[k] = nv; l0[k] = nl; nv += part[k].vkey.size(); nl += part[k].tkey.size(); }
    v0[np] = nv; l0[np] = nl;
    std::vector<uint64_t> vkey(nv), tkey(nl);
    std::vector<uint32_t> vrow(nv + 1), tmet(nl);
    std::vector<uint8_t> vfl(nv);
    std::vector<LanId> lans(nv);
    auto place = [&](size_t k) {
      const KeyedPart &o = part[k];
      std::copy(o.vkey.begin(), o.vkey.end(), vkey.begin() + v0[k]);
      std::copy(o.tkey.begin(), o.tkey.end(), tkey.begin() + l0[k]);
      std::copy(o.tmet.begin(), o.tmet.end(), tmet.begin() + l0[k]);
      std::copy(o.vfl.begin(), o.vfl.end(), vfl.begin() + v0[k]);
      std::copy(o.lans.begin(), o.lans.end(), lans.begin() + v0[k]);
      uint32_t at = (uint32_t)l0[k];
      for (size_t i = 0; i < o.vrow_len.size(); ++i) { vrow[v0[k] + i] = at; at += o.vrow_len[i]; }
    };
    if (np == 1) place(0);
    else {
      std::vector<std::thread> th;
      for (size_t k = 0; k < np; ++k) th.emplace_back([&, k]() { place(k); });
      for (auto &t : th) t.join();
    }
    vrow[nv] = (uint32_t)nl;
    if (lans.empty()) return false;
    const auto t_stream = std::chrono::steady_clock::now();
    std::vector<uint32_t> rank;
    dev_ = engine.upload_keyed(vkey, vrow, tkey, tmet, vfl, max_path_metric, rank, row_ptr, col, metric, vflags);
    if (!dev_) return false;
    const auto t_engine = std::chrono::steady_clock::now();
    dev_engine_ = &engine;
    vids.resize(lans.size());
    for (size_t i = 0; i < lans.size(); ++i) vids[rank[i]] = vertex_id(lans[i]);
    if (getenv("HSPF_KEYED_TIMING"))
      fprintf(stderr, "[LevelGraph keyed] stream %.2f ms, engine %.2f ms, vertex ids %.2f ms\n", std::chrono::duration<double, std::milli>(t_stream - t_begin).count(),
              std::chrono::duration<double, std::milli>(t_engine - t_stream).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_engine).count());
    return true;
  }
  struct Row { std::vector<uint32_t> col, metric; uint8_t flags; };
  using CfgKey = std::tuple<std::string, bool, bool>;       // everything outside the LSDB a row depends on
  CfgKey cfg_key(const InstanceCfg &cfg) const { return {cfg.metric_type.at(level), cfg.ipv4_enabled, cfg.ipv6_enabled}; }
  const Lsdb &lsdb_of(const Instance &inst) const {
    static const Lsdb empty;
    auto li = inst.lsdb.find(level);
    return li == inst.lsdb.end() ? empty : li->second;
  }
  static std::map<LanId, std::vector<const Lsp *>> live_fragments(const Lsdb &lsdb) {
    std::map<LanId, std::vector<const Lsp *>> frags;
    for (auto &kv : lsdb.all())
      if (kv.second.live()) frags[kv.second.lan_id()].push_back(&kv.second);
    return frags;
  }
  // links (fragment, then TLV order: spf.rs:1013-1128) and gate flags (spf.rs:557-604) of one vertex
  Row row(const LanId &lan, std::map<LanId, std::vector<const Lsp *>> &frags, const Lsdb &lsdb, const InstanceCfg &cfg) const {
    Row r;
    for (const Lsp *lsp : frags[lan])
      for (auto &e : vertex_edges(*lsp, mt_id, hopcount, metric_type)) {
        // (binary search in the sorted vertex list: a million lookups per 100 000-router LSDB, contiguous memory instead
        // of the tree's nodes)
        const VertexId vid = vertex_id(e.first);
        auto it = std::lower_bound(vids.begin(), vids.end(), vid);
        if (it != vids.end() && *it == vid) { r.col.push_back((uint32_t)(it - vids.begin())); r.metric.push_back(e.second); }
      }
    const bool is_pn = lan.pseudonode != 0;
    uint8_t f = is_pn ? HSPF_VF_NETWORK : 0;
    const Lsp *z = lsdb.zeroth_lsp(lan);
    if (!z) { r.flags = f | HSPF_VF_NO_EXPAND; return r; }                   // spf.rs:557-561
    if (!is_pn && mt_id && z->overload_bit(*mt_id)) f |= HSPF_VF_NO_TRANSIT; // spf.rs:568-574
    if (mt_id && *mt_id == MT_STANDARD && !is_pn) {                          // spf.rs:582-604
      auto has = [&](int p) { return z->protocols_supported && std::find(z->protocols_supported->begin(), z->protocols_supported->end(), p) != z->protocols_supported->end(); };
      if (!z->protocols_supported || (cfg.ipv4_enabled && !has(NLPID_IPV4)) || (cfg.ipv6_enabled && !has(NLPID_IPV6)))
        f |= HSPF_VF_NO_EXPAND;
    }
    r.flags = f;
    return r;
  }
  CfgKey cfg_key_;
  std::unique_ptr<Graph> dev_;
  Engine *dev_engine_ = nullptr;
};

// ---- SPT ---------------------------------------------------------------------------------------------------------
struct VertexNexthop {                 // holo-isis/src/spf.rs:107-114
  SystemId system_id{};
  std::optional<std::string> iface_name, ipv4, ipv6;
};
struct Vertex {                        // holo-isis/src/spf.rs:78-88
  VertexId id;
  uint32_t distance = 0;
  uint16_t hops = 0;
  std::vector<std::shared_ptr<VertexNexthop>> nexthops;
};
using RankKey = std::array<uint64_t, 4>;

class Spt {                            // holo-isis/src/spf.rs:67-73, 224-297
 public:
  std::map<VertexId, Vertex> vertices;
  std::vector<VertexId> pop_order;
  const Vertex *get(const VertexId &v) const { auto it = vertices.find(v); return it == vertices.end() ? nullptr : &it->second; }
  std::vector<const Vertex *> hops_eq(uint16_t h) const {
    std::vector<const Vertex *> out;
    for (auto &v : pop_order) { const Vertex &x = vertices.at(v); if (v.non_pseudonode && x.hops == h) out.push_back(&x); }
    return out;
  }
  std::vector<const Vertex *> first_hops() const { return hops_eq(1); }
  std::vector<const Vertex *> second_hops() const { return hops_eq(2); }
  // `Vertex.parents` (spf.rs:85, 677), rebuilt on first use from the tight links of the graph
  const std::vector<VertexId> &parents(const VertexId &vid) {
    static const std::vector<VertexId> none;
    if (!parents_built_) { parents_built_ = true; if (rebuild_) rebuild_(parents_); }
    auto it = parents_.find(vid);
    return it == parents_.end() ? none : it->second;
  }
  bool is_on_path(const SystemId &ancestor, const SystemId &descendant) {     // spf.rs:261-286
    const VertexId a = vertex_id(LanId{ancestor, 0}), d = vertex_id(LanId{descendant, 0});
    if (!vertices.count(a) || !vertices.count(d)) return false;
    std::vector<VertexId> stack{d};
    std::set<VertexId> seen;
    while (!stack.empty()) {
      VertexId cur = stack.back(); stack.pop_back();
      if (cur == a) return true;
      if (!seen.insert(cur).second) continue;
      for (auto &p : parents(cur)) stack.push_back(p);
    }
    return false;
  }
  std::function<void(std::map<VertexId, std::vector<VertexId>> &)> rebuild_;
 private:
  bool parents_built_ = false;
  std::map<VertexId, std::vector<VertexId>> parents_;
};

// holo-isis/src/spf.rs:956-1010 (ifaces already in name order)
inline void resolve_nexthop(VertexNexthop &nh, int level, int mt_id, bool parent_is_pseudonode, const SystemId &target,
                            uint32_t link_cost, std::set<std::string> &used_adjs, const std::vector<const Interface *> &ifaces) {
  const std::string want = parent_is_pseudonode ? "broadcast" : "point-to-point";
  for (const Interface *iface : ifaces) {
    if (iface->interface_type != want) continue;
    const Adjacency *adj = nullptr;
    if (parent_is_pseudonode) {
      for (auto &a : iface->adjacencies)
        if (a.level_usage == "level-" + std::to_string(level) && a.system_id == target) { adj = &a; break; }
      if (adj && (!adj->in_topology(mt_id) || adj->state != "up")) adj = nullptr;
    } else {
      auto mi = iface->metric.find(level);
      if (mi == iface->metric.end() || mi->second != link_cost) continue;
      const Adjacency *a = iface->adjacencies.empty() ? nullptr : &iface->adjacencies[0];
      if (a && a->in_topology(mt_id) && a->intersects(level) && a->system_id == target && a->state == "up") adj = a;
    }
    if (!adj || used_adjs.count(adj->snpa)) continue;
    used_adjs.insert(adj->snpa);
    nh.iface_name = iface->name;
    nh.ipv4 = adj->ipv4_addrs.empty() ? std::optional<std::string>() : adj->ipv4_addrs[0];
    nh.ipv6 = adj->ipv6_addrs.empty() ? std::optional<std::string>() : adj->ipv6_addrs[0];
    return;
  }
}

namespace detail {
inline uint32_t sat_add(uint32_t a, uint32_t b) { const uint64_t s = (uint64_t)a + b; return s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s; }
struct RunView {                       // one root's rows of a Tables
  const uint32_t *dist; const uint16_t *hops; const uint16_t *flags; const uint64_t *mask; uint32_t words;
  bool in_spt(uint32_t v) const { return flags[v] & HSPF_RF_IN_SPT; }
};

// Replays the relaxations made from hops == 0 vertices, in the reference's order, to give every first-hop slot its
// VertexNexthop (spf.rs:680-701).  resolve_nexthop is order dependent through `used_adjs`, and the reference calls it
// for EVERY relaxation that is not Ordering::Greater at that moment — also for candidates a later, shorter path
// replaces — so the replay evaluates the candidate-list state at each of those moments from the final distances.
inline std::map<uint32_t, std::shared_ptr<VertexNexthop>>
slot_nexthops(const LevelGraph &g, const SlotTable &st, const RunView &r, const std::function<RankKey(uint32_t)> &rank,
              bool local, int level, const Instance &inst) {
  std::vector<std::pair<uint32_t, uint32_t>> parents;
  for (size_t i = 0; i < st.vertex.size(); ++i)
    if (r.in_spt(st.vertex[i]) && r.hops[st.vertex[i]] == 0) parents.push_back({st.vertex[i], st.base[i]});
  std::stable_sort(parents.begin(), parents.end(), [&](auto &a, auto &b) { return rank(a.first) < rank(b.first); });
  const auto ifaces = inst.interfaces_by_name();
  std::set<std::string> used_adjs;
  std::map<uint32_t, std::shared_ptr<VertexNexthop>> out;
  const uint32_t max_path = g.max_path_metric;
  const bool ignore_ovl = !g.mt_id;
  auto expandable = [&](uint32_t u) {
    const uint8_t f = g.vflags[u];
    if (f & HSPF_VF_NO_EXPAND) return false;
    if (r.hops[u] != 0 && !(f & HSPF_VF_NETWORK) && !ignore_ovl && (f & HSPF_VF_NO_TRANSIT)) return false;
    return true;
  };
  // distance of t on the candidate list just before link `upto_k` of p is processed
  auto cand_before = [&](uint32_t t, uint32_t p, uint32_t upto_k) -> std::optional<uint32_t> {
    std::optional<uint32_t> best;
    const RankKey pk = rank(p);
    std::set<uint32_t> srcs(g.col.begin() + g.row_ptr[t], g.col.begin() + g.row_ptr[t + 1]);   // two-way => u lists t
    for (uint32_t u : srcs) {
      if (!r.in_spt(u) || !expandable(u)) continue;
      if (rank(u) > pk) continue;
      for (uint32_t k = g.row_ptr[u]; k < g.row_ptr[u + 1]; ++k) {
        if (g.col[k] != t) continue;
        if (u == p && k >= upto_k) break;
        const uint32_t d = sat_add(r.dist[u], g.metric[k]);
        if (d <= max_path && (!best || d < *best)) best = d;
      }
    }
    return best;
  };
  for (auto &pb : parents) {
    const uint32_t p = pb.first, base = pb.second;
    if (!expandable(p)) continue;
    for (uint32_t k = g.row_ptr[p]; k < g.row_ptr[p + 1]; ++k) {
      const uint32_t t = g.col[k];
      if (!g.links_back(t, p)) continue;
      if (r.in_spt(t) && rank(t) < rank(p)) continue;                 // already on the SPT
      const uint32_t d = sat_add(r.dist[p], g.metric[k]);
      if (d > max_path) continue;
      const auto cur = cand_before(t, p, k);
      if (cur && d > *cur) continue;                                   // Ordering::Greater
      if (g.vflags[t] & HSPF_VF_NETWORK) continue;                     // pseudonode target: no next hop
      auto nh = std::make_shared<VertexNexthop>();
      nh->system_id = g.vids[t].lan_id.system_id;
      if (local && g.mt_id)
        resolve_nexthop(*nh, level, *g.mt_id, g.vflags[p] & HSPF_VF_NETWORK, nh->system_id, g.metric[k], used_adjs, ifaces);
      out[base + (k - g.row_ptr[p])] = nh;
    }
  }
  return out;
}
}  // namespace detail

// All SPTs of one level/topology for a list of roots with ONE engine run — the shape of flooding::manet::init_cache
// (holo-isis/src/flooding/manet.rs:47-69).
inline std::vector<Spt> compute_spts(int level, const std::vector<SystemId> &root_system_ids, bool local,
                                     std::optional<int> mt_id, bool hopcount, const Instance &inst, Engine &engine,
                                     LevelGraph *graph = nullptr) {
  // the SPTs keep the graph alive for their lazily rebuilt parent lists; a graph passed in must outlive them
  std::shared_ptr<LevelGraph> keep = graph ? std::shared_ptr<LevelGraph>(graph, [](LevelGraph *) {})
                                           : std::make_shared<LevelGraph>(inst, level, mt_id, hopcount);
  const LevelGraph &G = *keep;
  std::vector<Spt> spts(root_system_ids.size());
  std::vector<uint32_t> roots, where;
  for (size_t i = 0; i < root_system_ids.size(); ++i) {
    const VertexId rv = vertex_id(LanId{root_system_ids[i], 0});
    auto it = G.index.find(rv);
    if (it == G.index.end()) {           // root without any LSP: inserted into the SPT and not expanded (spf.rs:552-561)
      spts[i].vertices[rv] = Vertex{rv, 0, 0, {}};
      spts[i].pop_order = {rv};
    } else { roots.push_back(it->second); where.push_back((uint32_t)i); }
  }
  if (roots.empty()) return spts;
  Graph &dev = keep->device(engine);
  auto res = std::make_shared<Tables>(engine.run(dev, roots, G.run_flags));
  const uint32_t n = G.n(), W = res->mask_words;
  // roots the engine ran through its sequential kernel (zero-cost plateaus: the pop order is not the static
  // (distance, id) order) are re-run once more for their exact pop ranks
  std::vector<uint32_t> exact_j;
  for (uint32_t j = 0; j < roots.size(); ++j)
    for (uint32_t v = 0; v < n; ++v)
      if (res->flags[(size_t)j * n + v] & HSPF_RF_EXACT) { exact_j.push_back(j); break; }
  std::shared_ptr<Tables> rr;
  std::map<uint32_t, uint32_t> exact_row;
  if (!exact_j.empty()) {
    std::vector<uint32_t> er;
    for (uint32_t j : exact_j) { exact_row[j] = (uint32_t)er.size(); er.push_back(roots[j]); }
    rr = std::make_shared<Tables>(engine.run(dev, er, G.run_flags | HSPF_RUN_POP_RANK));
  }
  // One root's Spt is a function of its own rows of the tables: several roots (flooding::manet::init_cache hands in every
  // neighbour) are rebuilt side by side — the engine calls (one slot table per root) first, on this thread, then the roots on
  // up to 16 threads (graphs of 4 096 vertices and more; HSPF_KEYED_THREADS as for the LSDB walks; 16 roots at 100 000
  // vertices were 16 x 15.6 ms one after the other).
  std::vector<SlotTable> slot_tables;
  slot_tables.reserve(roots.size());
  for (uint32_t j = 0; j < roots.size(); ++j) slot_tables.push_back(engine.slot_table(dev, roots[j]));
  auto rebuild_root = [&](uint32_t j) {
    detail::RunView r{&res->dist[(size_t)j * n], &res->hops[(size_t)j * n], &res->flags[(size_t)j * n], &res->mask[(size_t)j * n * W], W};
    std::function<RankKey(uint32_t)> rank;
    auto xr = exact_row.find(j);
    if (xr != exact_row.end()) {
      const uint32_t *pr = &rr->pop_rank[(size_t)xr->second * n];
      auto hold = rr;
      rank = [pr, hold](uint32_t v) { return RankKey{pr[v], 0, 0, 0}; };
    } else if (G.hopcount) {
      // Hop-count graphs (spf.rs:1138-1145): links into pseudonodes cost 0, so a pseudonode is put on the candidate
      // list by the lowest-numbered router of its own distance that lists it and, sorting before every router, is
      // popped right after that router.
      auto cache = std::make_shared<std::map<uint32_t, RankKey>>();
      rank = [keep, res, r, cache](uint32_t v) {
        auto it = cache->find(v);
        if (it != cache->end()) return it->second;
        const LevelGraph &g = *keep;
        RankKey k{r.dist[v], v, 0, 0};
        if (g.vflags[v] & HSPF_VF_NETWORK) {
          std::optional<uint32_t> act;
          for (uint32_t e = g.row_ptr[v]; e < g.row_ptr[v + 1]; ++e) {
            const uint32_t u = g.col[e];
            if (r.in_spt(u) && r.dist[u] == r.dist[v] && !(g.vflags[u] & HSPF_VF_NO_EXPAND) && g.links_back(u, v))
              if (!act || u < *act) act = u;
          }
          if (act) k = RankKey{r.dist[v], *act, 1, v};
        }
        (*cache)[v] = k;
        return k;
      };
    } else {
      auto hold = res;
      rank = [r, hold](uint32_t v) { return RankKey{r.dist[v], v, 0, 0}; };   // static order
    }
    const SlotTable &st = slot_tables[j];
    auto slot_nh = detail::slot_nexthops(G, st, r, rank, local, level, inst);
    Spt &s = spts[where[j]];
    std::vector<uint32_t> members;
    for (uint32_t v = 0; v < n; ++v)
      if (r.in_spt(v)) members.push_back(v);
    for (uint32_t v : members) {
      Vertex vx{G.vids[v], r.dist[v], r.hops[v], {}};
      for (uint32_t w = 0; w < W; ++w) {
        uint64_t m = r.mask[(size_t)v * W + w];
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          auto it = slot_nh.find(w * 64 + b);
          if (it != slot_nh.end()) vx.nexthops.push_back(it->second);
        }
      }
      s.vertices.insert_or_assign(s.vertices.end(), vx.id, std::move(vx));       // (vertex indices are VertexId ranks: appended at the end)
    }
    {                                                                            // pop order: every member's rank ONCE, then the sort
      std::vector<std::pair<RankKey, uint32_t>> ranked;
      ranked.reserve(members.size());
      for (uint32_t v : members) ranked.push_back({rank(v), v});
      std::stable_sort(ranked.begin(), ranked.end(), [](const std::pair<RankKey, uint32_t> &a, const std::pair<RankKey, uint32_t> &b) { return a.first < b.first; });
      for (size_t i = 0; i < members.size(); ++i) members[i] = ranked[i].second;
    }
    s.pop_order.reserve(members.size());
    for (uint32_t v : members) s.pop_order.push_back(G.vids[v]);
    s.rebuild_ = [keep, res, r, rank, members](std::map<VertexId, std::vector<VertexId>> &out) {
      const LevelGraph &g = *keep;
      const bool ignore_ovl = !g.mt_id;
      for (uint32_t u : members) {
        const uint8_t f = g.vflags[u];
        if (f & HSPF_VF_NO_EXPAND) continue;
        if (r.hops[u] != 0 && !(f & HSPF_VF_NETWORK) && !ignore_ovl && (f & HSPF_VF_NO_TRANSIT)) continue;
        for (uint32_t k = g.row_ptr[u]; k < g.row_ptr[u + 1]; ++k) {
          const uint32_t t = g.col[k];
          if (!r.in_spt(t) || !(rank(u) < rank(t)) || !g.links_back(t, u)) continue;
          if (detail::sat_add(r.dist[u], g.metric[k]) == r.dist[t]) out[g.vids[t]].push_back(g.vids[u]);
        }
      }
    };
  };
  unsigned T = 1;
  if (roots.size() > 1 && n >= 4096) {
    T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("HSPF_KEYED_THREADS")) T = std::max(1, atoi(e));
    T = std::min<unsigned>(T, (unsigned)roots.size());
  }
  if (T == 1) for (uint32_t j = 0; j < roots.size(); ++j) rebuild_root(j);
  else {
    std::vector<std::exception_ptr> failed(T);
    std::vector<std::thread> th;
    for (unsigned k = 0; k < T; ++k)
      th.emplace_back([&, k]() {
        try { for (uint32_t j = k; j < roots.size(); j += T) rebuild_root(j); } catch (...) { failed[k] = std::current_exception(); }
      });
    for (auto &t : th) t.join();
    for (auto &f : failed) if (f) std::rethrow_exception(f);
  }
  return spts;
}

// holo-isis/src/spf.rs:527-709
inline Spt compute_spt(int level, const SystemId &root, bool local, std::optional<int> mt_id, bool hopcount,
                       const Instance &inst, Engine &engine, LevelGraph *graph = nullptr) {
  return std::move(compute_spts(level, {root}, local, mt_id, hopcount, inst, engine, graph)[0]);
}

// ---- routes --------------------------------------------------------------------------------------------------------
struct Nexthop { std::string addr, iface_name; SystemId system_id{}; std::optional<uint32_t> label; };   // label: SR output label
struct Route {                         // holo-isis/src/route.rs:27-37
  std::string prefix;
  uint32_t metric = 0;
  int level = 0;
  bool external = false, connected = false;
  std::map<IpKey, Nexthop> nexthops;   // BTreeMap<IpAddr, Nexthop>: ECMP order = ascending address
  std::optional<PrefixSid> prefix_sid; // of the network the route was CREATED from (route.rs:78-103)
  std::optional<uint32_t> sr_label;    // SR input label
};
struct Network { std::string prefix; uint32_t metric; bool external; std::optional<PrefixSid> sid; };

// holo-isis/src/spf.rs:1149-1296
// (the fragments of the vertex's LAN id handed in: Lsdb::iter_for_lan_id / fragments_from)
inline std::vector<Network> vertex_networks(const Instance &inst, int level, int mt_id, const std::vector<const Lsp *> &fragments, bool att_bit,
                                            bool l2_attached, bool ipv4_enabled, bool ipv6_enabled) {
  const InstanceCfg &cfg = inst.config;
  const std::string &mt = cfg.metric_type.at(level);
  const bool std_on = mt == "standard" || mt == "both", wide_on = mt == "wide" || mt == "both";
  std::vector<Network> out;
  for (const Lsp *lsp : fragments) {
    if (!lsp->live()) continue;
    if (att_bit && level == 1 && (cfg.level_type == "level-1" || !l2_attached)) {
      if (ipv4_enabled) out.push_back({"0.0.0.0/0", 0, false});
      if (ipv6_enabled) out.push_back({"::/0", 0, false});
    }
    if (mt_id == MT_STANDARD && ipv4_enabled) {
      if (std_on) {
        for (auto &p : lsp->ipv4_internal) out.push_back({p.first, p.second, false});
        for (auto &p : lsp->ipv4_external) out.push_back({p.first, p.second, true});
      }
      auto sid_of = [&](const char *kind, size_t i) -> std::optional<PrefixSid> {
        const PrefixSid *q = lsp->prefix_sid(kind, (int)i);
        return q ? std::optional<PrefixSid>(*q) : std::nullopt;
      };
      if (wide_on)
        for (size_t i = 0; i < lsp->ext_ipv4.size(); ++i) {
          auto &p = lsp->ext_ipv4[i];
          if (std::get<1>(p) <= MAX_PATH_METRIC_WIDE) out.push_back({std::get<0>(p), std::get<1>(p), std::get<2>(p), sid_of("ext_ipv4", i)});
        }
    }
    if (ipv6_enabled) {
      auto sid_of = [&](const char *kind, size_t i) -> std::optional<PrefixSid> {
        const PrefixSid *q = lsp->prefix_sid(kind, (int)i);
        return q ? std::optional<PrefixSid>(*q) : std::nullopt;
      };
      if (mt_id == MT_IPV6_UNICAST) {
        for (size_t i = 0; i < lsp->mt_ipv6.size(); ++i) {
          auto &p = lsp->mt_ipv6[i];
          if (std::get<0>(p) == MT_IPV6_UNICAST && std::get<2>(p) <= MAX_PATH_METRIC_WIDE) out.push_back({std::get<1>(p), std::get<2>(p), std::get<3>(p), sid_of("mt_ipv6", i)});
        }
      } else {
        for (size_t i = 0; i < lsp->ipv6.size(); ++i) {
          auto &p = lsp->ipv6[i];
          if (std::get<1>(p) <= MAX_PATH_METRIC_WIDE) out.push_back({std::get<0>(p), std::get<1>(p), std::get<2>(p), sid_of("ipv6", i)});
        }
      }
    }
  }
  return out;
}
inline std::vector<Network> vertex_networks(const Instance &inst, int level, int mt_id, const LanId &lan, bool att_bit,
                                            bool l2_attached, bool ipv4_enabled, bool ipv6_enabled) {
  auto li = inst.lsdb.find(level);
  if (li == inst.lsdb.end()) return {};
  return vertex_networks(inst, level, mt_id, li->second.iter_for_lan_id(lan), att_bit, l2_attached, ipv4_enabled, ipv6_enabled);
}

inline std::map<IpKey, Nexthop> build_nexthops(const Vertex &v, const std::string &prefix) {     // route.rs:118-142
  const bool v6 = prefix.find(':') != std::string::npos;
  std::map<IpKey, Nexthop> out;
  for (auto &nh : v.nexthops) {
    const auto &addr = v6 ? nh->ipv6 : nh->ipv4;
    if (addr) out[parse_ip(*addr)] = Nexthop{*addr, nh->iface_name.value_or(""), nh->system_id};
  }
  return out;
}

// ---- SR Prefix-SID bookkeeping of compute_routes (holo-isis/src/spf.rs:931-946, sr.rs:34-94, 165-300) ------------------
// The SPT feeds it two bits per route update: local = (vertex.hops == 0), last_hop = (vertex.hops == 1); the rest is label
// arithmetic over the LSDB's SR-Capabilities.  No conformance fixture of the reference has `sr.enabled`: this step is
// checked against the literal restatement (oracle/isis_ref.py) on random instances, as the Python twin is (parity unpinned).
constexpr uint32_t LABEL_EXPLICIT_NULL_V4 = 0, LABEL_EXPLICIT_NULL_V6 = 2, LABEL_IMPLICIT_NULL = 3;
inline const SrCap *sr_cap_of(const Lsdb &lsdb, const SystemId &system_id) {
  for (const Lsp *lsp : lsdb.iter_for_system_id(system_id))
    if (lsp->live() && lsp->sr_cap) return &*lsp->sr_cap;
  return nullptr;
}
inline std::optional<uint32_t> sr_index_to_label(uint32_t index, const std::vector<std::pair<uint32_t, uint32_t>> &srgbs) {   // sr.rs:270-300
  for (auto &r : srgbs) {
    if (index >= r.second) { index -= r.second; continue; }
    return r.first + index;
  }
  return std::nullopt;
}
inline void prefix_sid_update(const Instance &inst, int level, const LanId &adv_rtr, const std::string &prefix, Route &route, bool local, bool last_hop) {
  if (!route.prefix_sid) return;
  const PrefixSid &sid = *route.prefix_sid;
  auto li = inst.lsdb.find(level);
  if (li == inst.lsdb.end()) return;
  const Lsdb &lsdb = li->second;
  bool spf_algo = false;
  for (const Lsp *lsp : lsdb.iter_for_lan_id(adv_rtr))
    spf_algo = spf_algo || (lsp->live() && std::find(lsp->sr_algos.begin(), lsp->sr_algos.end(), 0) != lsp->sr_algos.end());
  if (!spf_algo) return;
  const bool v6 = prefix.find(':') != std::string::npos;
  // input label (sr.rs:165-205)
  if (local && (!sid.has("P") || sid.has("E"))) route.sr_label.reset();
  else if (sid.index) {
    const SrCap *cap = sr_cap_of(lsdb, inst.config.system_id);
    const auto label = cap ? sr_index_to_label(*sid.index, cap->srgb) : std::nullopt;
    if (label) route.sr_label = *label;
  } else route.sr_label = sid.label;
  // output labels (sr.rs:208-267)
  for (auto &kv : route.nexthops) {
    Nexthop &nh = kv.second;
    uint32_t label;
    if (last_hop && !sid.has("P")) label = LABEL_IMPLICIT_NULL;
    else {
      const SrCap *cap = sr_cap_of(lsdb, nh.system_id);
      if (!cap || !cap->has(v6 ? "V" : "I")) continue;
      if (last_hop && sid.has("E")) label = v6 ? LABEL_EXPLICIT_NULL_V6 : LABEL_EXPLICIT_NULL_V4;
      else if (sid.index) {
        const auto l = sr_index_to_label(*sid.index, cap->srgb);
        if (!l) continue;
        label = *l;
      } else label = last_hop ? *sid.label : LABEL_IMPLICIT_NULL;
    }
    nh.label = label;
  }
}

// holo-isis/src/spf.rs:840-949
inline void compute_routes(int level, int mt_id, const Instance &inst, const Spt &spt, std::map<IpKey, Route> &rib) {
  const InstanceCfg &cfg = inst.config;
  const bool l2_attached = inst.is_l2_attached_to_backbone(mt_id);
  const bool ipv4_enabled = cfg.ipv4_enabled && mt_id == MT_STANDARD;
  const bool ipv6_enabled = cfg.ipv6_enabled && (mt_id == MT_STANDARD ? !cfg.is_topology_enabled(MT_IPV6_UNICAST) : true);
  auto li = inst.lsdb.find(level);
  if (li == inst.lsdb.end()) return;
  Lsdb::Cursor cursor[2];                                               // (VertexId order = ascending LAN ids per class: pseudonodes, routers)
  std::vector<const Lsp *> fragments;
  for (auto &kv : spt.vertices) {                                       // Spt::iter: VertexId order
    const Vertex &vertex = kv.second;
    const Lsp *z = li->second.fragments_from(cursor[vertex.id.lan_id.pseudonode == 0], vertex.id.lan_id, fragments);
    if (!z) continue;
    const bool att = !cfg.att_ignore && z->att_bit(mt_id) && !z->overload_bit(mt_id);
    for (auto &net : vertex_networks(inst, level, mt_id, fragments, att, l2_attached, ipv4_enabled, ipv6_enabled)) {
      const IpKey key = parse_ip(net.prefix);
      const uint32_t route_metric = vertex.distance + net.metric;      // route.rs:97, plain `+`
      auto [it, is_new] = rib.try_emplace(key);                        // (one descent of the RIB per network, not three)
      Route *cur;
      if (is_new || route_metric < it->second.metric) {
        it->second = Route{net.prefix, route_metric, level, net.external, vertex.hops == 0, build_nexthops(vertex, net.prefix), net.sid, std::nullopt};
        cur = &it->second;
      } else if (route_metric == it->second.metric) {
        cur = &it->second;
        for (auto &n : build_nexthops(vertex, net.prefix)) cur->nexthops[n.first] = n.second;
      } else continue;
      while (cur->nexthops.size() > cfg.max_paths) cur->nexthops.erase(std::prev(cur->nexthops.end()));   // first k by key
      if (cfg.sr_enabled && cur->prefix_sid) prefix_sid_update(inst, level, vertex.id.lan_id, net.prefix, *cur, vertex.hops == 0, vertex.hops == 1);   // spf.rs:931-946
    }
  }
}

struct RibRow {
  std::string prefix; uint32_t metric; int level; std::vector<std::pair<std::string, std::string>> nexthops;
  bool sr = false;                                             // the SR columns only where SR is on
  std::optional<uint32_t> sr_label; std::vector<std::optional<uint32_t>> nexthop_labels;
};

// LAN ids with an LSP fragment that differs between two LSDB snapshots (what the reference accumulates in `trigger_lsps`)
inline std::vector<LanId> changed_lan_ids(const Lsdb &old_db, const Lsdb &new_db) {
  std::set<LanId> out;
  for (auto &kv : old_db.all()) { auto it = new_db.all().find(kv.first); if (it == new_db.all().end() || !(it->second == kv.second)) out.insert(kv.second.lan_id()); }
  for (auto &kv : new_db.all()) if (!old_db.all().count(kv.first)) out.insert(kv.second.lan_id());
  return std::vector<LanId>(out.begin(), out.end());
}

// Level graphs kept on the device across SPF runs, brought up to date from the changed LSPs.
class GraphCache {
 public:
  int rebuilt = 0, patched = 0;
  Engine *keyed = nullptr;         // set: a (re)build streams the LSDB's records to this engine, which derives the CSR (LevelGraph)
  std::map<std::tuple<int, int, bool>, std::unique_ptr<LevelGraph>> graphs;       // (level, mt_id or -1, hop count)
  LevelGraph &get(const Instance &inst, int level, std::optional<int> mt_id, bool hopcount, const std::vector<LanId> *trigger_lsps) {
    auto key = std::make_tuple(level, mt_id ? *mt_id : -1, hopcount);
    auto it = graphs.find(key);
    if (it != graphs.end() && trigger_lsps && it->second->refresh(inst, *trigger_lsps)) { ++patched; return *it->second; }
    graphs[key] = std::make_unique<LevelGraph>(inst, level, mt_id, hopcount, keyed);
    ++rebuilt;
    return *graphs[key];
  }
};

// Full SPF of every configured level and topology (holo-isis/src/spf.rs:719-836) followed by the L1/L2 merge of
// holo-isis/src/route.rs:185-249; rows like the YANG `local-rib`.
inline std::vector<RibRow> compute_spf(const Instance &inst, Engine &engine, GraphCache *cache = nullptr,
                                       const std::map<int, std::vector<LanId>> *trigger_lsps = nullptr) {
  const InstanceCfg &cfg = inst.config;
  std::map<int, std::map<IpKey, Route>> per_level;
  for (int level : cfg.levels()) {
    std::map<IpKey, Route> rib;
    for (int mt_id : {MT_STANDARD, MT_IPV6_UNICAST}) {
      if (!cfg.is_topology_enabled(mt_id)) continue;
      LevelGraph *graph = nullptr;
      if (cache) {
        static const std::vector<LanId> none;
        const std::vector<LanId> *trig = nullptr;
        if (trigger_lsps) { auto ti = trigger_lsps->find(level); trig = ti == trigger_lsps->end() ? &none : &ti->second; }
        graph = &cache->get(inst, level, mt_id, false, trig);
      }
      Spt spt = compute_spt(level, cfg.system_id, true, mt_id, false, inst, engine, graph);
      compute_routes(level, mt_id, inst, spt, rib);
    }
    per_level[level] = std::move(rib);
  }
  std::map<IpKey, Route> merged;
  if (per_level.size() == 1) merged = std::move(per_level.begin()->second);        // (one level: nothing to merge, nothing to copy)
  else
    for (int level : {2, 1})
      for (auto &kv : per_level[level]) merged[kv.first] = kv.second;
  std::vector<RibRow> rows;
  rows.reserve(merged.size());
  for (auto &kv : merged) {
    RibRow r{kv.second.prefix, kv.second.metric, kv.second.level, {}};
    for (auto &n : kv.second.nexthops) r.nexthops.push_back({n.second.addr, n.second.iface_name});
    if (cfg.sr_enabled) {
      r.sr = true; r.sr_label = kv.second.sr_label;
      for (auto &n : kv.second.nexthops) r.nexthop_labels.push_back(n.second.label);
    }
    rows.push_back(std::move(r));
  }
  return rows;
}

// ---- route derivation with the prefix attachment on the device -------------------------------------
// Every (vertex, prefix, metric) the unchanged vertex_networks() yields, vertices in VertexId order, as a CSR-by-prefix
// table (root independent, built once per LSDB generation); hspf_routes_device reduces it for every root of a run.
struct PrefixTable {
  std::vector<std::string> prefixes;           // BTreeMap<IpNetwork, _> order
  std::vector<IpKey> keys;                     // ... and their parsed form
  std::vector<uint32_t> pfx_ptr, pfx_vertex, pfx_metric;
  std::vector<uint8_t> external;
  using PfxSig = std::vector<std::tuple<std::string, uint32_t, bool>>;     // what one vertex attaches: (prefix, metric, external) in order of appearance
  // position of a prefix in the table (the keys are in ascending order), or -1
  long find(const IpKey &k) const {
    auto it = std::lower_bound(keys.begin(), keys.end(), k);
    return it != keys.end() && *it == k ? (long)(it - keys.begin()) : -1;
  }
  // `sigs` (by vertex index): what every vertex contributed, for the running instance's "did an LSP's prefixes change" check —
  // from the same pass (a second walk of 100 000 vertices cost 55 ms).  Graphs of 4 096 vertices and more: ranges of vertices on
  // up to 16 threads (HSPF_KEYED_THREADS, as the keyed LSDB extraction), rows laid end to end in vertex order afterwards.
  static PrefixTable build(const Instance &inst, int level, int mt_id, const LevelGraph &g, std::vector<PfxSig> *sigs = nullptr) {
    const InstanceCfg &cfg = inst.config;
    const bool l2_attached = inst.is_l2_attached_to_backbone(mt_id);
    const bool v4 = cfg.ipv4_enabled && mt_id == MT_STANDARD;
    const bool v6 = cfg.ipv6_enabled && (mt_id == MT_STANDARD ? !cfg.is_topology_enabled(MT_IPV6_UNICAST) : true);
    struct Row { IpKey key; std::string prefix; uint32_t v, metric; bool ext; };
    const uint32_t n = g.n();
    if (sigs) { sigs->clear(); sigs->resize(n); }
    auto li = inst.lsdb.find(level);
    unsigned T = 1;
    if (n >= 4096) {
      T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
      if (const char *e = getenv("HSPF_KEYED_THREADS")) T = std::max(1, atoi(e));
    }
    std::vector<std::vector<Row>> part(T);
    auto walk = [&](unsigned k) {
      std::vector<Row> &rows = part[k];
      const uint32_t v0 = (uint32_t)((uint64_t)n * k / T), v1 = (uint32_t)((uint64_t)n * (k + 1) / T);
      Lsdb::Cursor cur[2];                                                       // vertices come in VertexId order: ascending LAN ids per class (pseudonode / router)
      std::vector<const Lsp *> frags;
      for (uint32_t v = v0; v < v1; ++v) {
        const LanId lan = g.vids[v].lan_id;
        const Lsp *z = li->second.fragments_from(cur[lan.pseudonode == 0], lan, frags);
        if (!z) continue;                                                        // spf.rs:866-869
        const bool att = !cfg.att_ignore && z->att_bit(mt_id) && !z->overload_bit(mt_id);
        auto nets = vertex_networks(inst, level, mt_id, frags, att, l2_attached, v4, v6);
        if (sigs) { PfxSig &sg = (*sigs)[v]; sg.reserve(nets.size()); for (auto &net : nets) sg.emplace_back(net.prefix, net.metric, net.external); }
        for (auto &net : nets) {
          IpKey key = parse_ip(net.prefix);
          rows.push_back({key, std::move(net.prefix), v, net.metric, net.external});
        }
      }
    };
    const bool tdbg = getenv("HSPF_TWIN_TIMING") != nullptr;           // (stage times on stderr, as compute_spf_device_routes)
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!tdbg) return;
      auto t = std::chrono::steady_clock::now();
      fprintf(stderr, "[twin prefix table]  %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
      t_prev = t;
    };
    if (li != inst.lsdb.end()) {
      if (T == 1) walk(0);
      else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < T; ++k) th.emplace_back(walk, k);
        for (auto &t : th) t.join();
      }
    }
    // by (prefix, vertex, order of appearance): the rows were made in (vertex, appearance) order, so a STABLE sort by prefix of
    // their addresses is that order (the rows themselves — a string each — stay where they are)
    lap("walk of the vertices");
    size_t total = 0;
    for (auto &r : part) total += r.size();
    // (sorted by VALUE: the key packed into three integers whose order is IpKey's — version, the 16 address bytes, length —
    // beside the row's address; comparing through 120 000 pointers cost twice as much)
    struct Ord { uint64_t a, b; uint32_t c; Row *row; };
    std::vector<Ord> order;
    order.reserve(total);
    for (auto &r : part)
      for (auto &x : r) {
        uint64_t a = (uint64_t)(uint8_t)x.key.version, b = 0;
        for (int i = 0; i < 7; ++i) a = (a << 8) | x.key.addr[i];
        for (int i = 7; i < 15; ++i) b = (b << 8) | x.key.addr[i];
        order.push_back({a, b, ((uint32_t)x.key.addr[15] << 16) | (uint32_t)(uint16_t)x.key.len, &x});
      }
    std::stable_sort(order.begin(), order.end(), [](const Ord &x, const Ord &y) { return std::tie(x.a, x.b, x.c) < std::tie(y.a, y.b, y.c); });
    lap("sort");
    PrefixTable t;
    t.pfx_vertex.reserve(total); t.pfx_metric.reserve(total); t.external.reserve(total);
    t.pfx_ptr.push_back(0);
    for (size_t i = 0; i < order.size(); ++i) {
      Row &r = *order[i].row;
      if (i == 0 || !(r.key == order[i - 1].row->key)) { if (i) t.pfx_ptr.push_back((uint32_t)i); t.keys.push_back(r.key); t.prefixes.push_back(std::move(r.prefix)); }
      t.pfx_vertex.push_back(r.v); t.pfx_metric.push_back(r.metric); t.external.push_back(r.ext);
    }
    if (total) t.pfx_ptr.push_back((uint32_t)total);
    lap("lay out");
    return t;
  }
};

// compute_spf (spf.rs:719-836) with BOTH the SPT and the prefix attachment on the device; same rows as compute_spf.
// The graph comes from the cache when there is one (as in compute_spf), else straight from the LSDB records through the engine's
// keyed upload (LevelGraph: 141 ms of host walk -> 9 ms at 100 000 LSPs).  Host work per prefix is a handful of pointer-sized
// moves: the next hop of every first-hop slot is parsed ONCE (not once per prefix that uses it), the routes arrive in prefix
// order and are appended to the RIB's end, and a single-level RIB is moved, not copied, into the rows.
inline std::vector<RibRow> compute_spf_device_routes(const Instance &inst, Engine &engine, GraphCache *cache = nullptr,
                                                     const std::map<int, std::vector<LanId>> *trigger_lsps = nullptr) {
  const InstanceCfg &cfg = inst.config;
  const bool tdbg = getenv("HSPF_TWIN_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tdbg) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[twin device routes] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  std::map<int, std::map<IpKey, Route>> per_level;
  // ONE (level, topology) table — the common shape — has nothing to merge: its prefixes come in RIB order, each once, and
  // go straight into the rows (no intermediate BTreeMap<IpNetwork, Route> with a BTreeMap of next hops per route: two heap
  // nodes per route made and freed again were a third of the host's share of a 120 000-route cold start)
  size_t n_tables = 0;
  for (int mt_id : {MT_STANDARD, MT_IPV6_UNICAST}) n_tables += cfg.is_topology_enabled(mt_id) ? cfg.levels().size() : 0;
  const bool single = n_tables == 1;
  std::vector<RibRow> rows;
  for (int level : cfg.levels()) {
    std::map<IpKey, Route> rib;
    for (int mt_id : {MT_STANDARD, MT_IPV6_UNICAST}) {
      if (!cfg.is_topology_enabled(mt_id)) continue;
      std::unique_ptr<LevelGraph> own;
      LevelGraph *g = nullptr;
      if (cache) {
        static const std::vector<LanId> none;
        const std::vector<LanId> *trig = nullptr;
        if (trigger_lsps) { auto ti = trigger_lsps->find(level); trig = ti == trigger_lsps->end() ? &none : &ti->second; }
        g = &cache->get(inst, level, mt_id, false, trig);
      } else {
        own = std::make_unique<LevelGraph>(inst, level, mt_id, false, &engine);
        g = own.get();
      }
      auto ri = g->index.find(vertex_id(LanId{cfg.system_id, 0}));
      if (ri == g->index.end()) continue;          // root without LSP: SPT = {root}, zeroth LSP missing -> no routes
      lap("level graph");
      const PrefixTable table = PrefixTable::build(inst, level, mt_id, *g);
      lap("prefix table");
      if (table.prefixes.empty()) continue;
      Graph &dev = g->device(engine);
      const uint32_t root = ri->second, n = g->n();
      auto run = engine.run_device(dev, {root}, g->run_flags);
      const RoutesOut ro = engine.routes(*run, table.pfx_ptr, table.pfx_vertex, table.pfx_metric, 0);
      auto res = std::make_shared<Tables>(run->host_tables());
      const uint32_t W = res->mask_words;
      detail::RunView r{res->dist.data(), res->hops.data(), res->flags.data(), res->mask.data(), W};
      std::function<RankKey(uint32_t)> rank;
      bool exact = false;
      for (uint32_t v = 0; v < n; ++v) exact |= (res->flags[v] & HSPF_RF_EXACT) != 0;
      std::shared_ptr<Tables> rr;
      if (exact) {
        rr = std::make_shared<Tables>(engine.run(dev, {root}, g->run_flags | HSPF_RUN_POP_RANK));
        rank = [rr](uint32_t v) { return RankKey{rr->pop_rank[v], 0, 0, 0}; };
      } else rank = [r](uint32_t v) { return RankKey{r.dist[v], v, 0, 0}; };
      lap("engine + tables");
      const auto slot_nh = detail::slot_nexthops(*g, engine.slot_table(dev, root), r, rank, true, level, inst);
      // every slot's next hop, parsed once: [family][slot]
      struct SlotNh { bool has = false; IpKey key; Nexthop nh; };
      std::vector<SlotNh> snh[2];
      snh[0].resize((size_t)W * 64); snh[1].resize((size_t)W * 64);
      for (auto &kv : slot_nh) {
        if (kv.first >= (size_t)W * 64) continue;
        for (int f = 0; f < 2; ++f) {
          const auto &addr = f ? kv.second->ipv6 : kv.second->ipv4;
          if (addr) snh[f][kv.first] = SlotNh{true, parse_ip(*addr), Nexthop{*addr, kv.second->iface_name.value_or(""), kv.second->system_id}};
        }
      }
      lap("slot next hops");
      std::vector<const SlotNh *> pick;
      if (single) rows.reserve(table.prefixes.size());
      for (size_t p = 0; p < table.prefixes.size(); ++p) {
        if (ro.best_entry[p] == 0xFFFFFFFFu) continue;
        const std::string &prefix = table.prefixes[p];
        const uint32_t metric = ro.best_metric[p];
        const IpKey &key = table.keys[p];
        const std::vector<SlotNh> &fam = snh[key.version == 6 ? 1 : 0];
        pick.clear();
        for (uint32_t w = 0; w < W; ++w) {
          uint64_t m = ro.nexthop_mask[p * W + w];
          while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const SlotNh &sn = fam[w * 64 + b];
            if (sn.has) pick.push_back(&sn);
          }
        }
        // BTreeMap<IpAddr, Nexthop>: ascending address, a later slot with the same address replaces the earlier one
        std::stable_sort(pick.begin(), pick.end(), [](const SlotNh *a, const SlotNh *b) { return a->key < b->key; });
        if (single) {
          RibRow row{prefix, metric, level, {}};
          row.nexthops.reserve(std::min<size_t>(pick.size(), cfg.max_paths));
          for (size_t i = 0; i < pick.size() && row.nexthops.size() < cfg.max_paths; ++i) {
            if (i + 1 < pick.size() && pick[i + 1]->key == pick[i]->key) continue;
            row.nexthops.push_back({pick[i]->nh.addr, pick[i]->nh.iface_name});
          }
          rows.push_back(std::move(row));
          continue;
        }
        std::map<IpKey, Nexthop> nhs;
        for (size_t i = 0; i < pick.size(); ++i) {
          if (i + 1 < pick.size() && pick[i + 1]->key == pick[i]->key) continue;
          nhs.emplace_hint(nhs.end(), pick[i]->key, pick[i]->nh);
        }
        Route *cur;
        const bool at_end = rib.empty() || rib.rbegin()->first < key;           // (the table is in prefix order)
        auto it = at_end ? rib.end() : rib.find(key);
        if (it == rib.end() || metric < it->second.metric) {
          const uint32_t v = table.pfx_vertex[ro.best_entry[p]];
          Route fresh{prefix, metric, level, (bool)table.external[ro.best_entry[p]], r.hops[v] == 0, std::move(nhs)};
          if (at_end) cur = &rib.emplace_hint(rib.end(), key, std::move(fresh))->second;
          else cur = &(rib[key] = std::move(fresh));
        } else if (metric == it->second.metric) { cur = &it->second; for (auto &kv : nhs) cur->nexthops[kv.first] = kv.second; }
        else continue;
        while (cur->nexthops.size() > cfg.max_paths) cur->nexthops.erase(std::prev(cur->nexthops.end()));
      }
    }
    lap("rib of the level");
    per_level[level] = std::move(rib);
  }
  if (single) return rows;
  std::map<IpKey, Route> merged;
  if (per_level.size() == 1) merged = std::move(per_level.begin()->second);
  else
    for (int level : {2, 1})
      for (auto &kv : per_level[level]) merged[kv.first] = kv.second;
  rows.reserve(merged.size());
  for (auto &kv : merged) {
    RibRow row{std::move(kv.second.prefix), kv.second.metric, kv.second.level, {}};
    row.nexthops.reserve(kv.second.nexthops.size());
    for (auto &nh : kv.second.nexthops) row.nexthops.push_back({std::move(nh.second.addr), std::move(nh.second.iface_name)});
    rows.push_back(std::move(row));
  }
  lap("merge + rows");
  return rows;
}

// ---- the wire step: update_global_rib, holo-isis/src/route.rs:254-312 ---------------------------------
// What goes on the ibus after an SPF: RouteIpAdd for every route that is new or differs from the RIB held before (metric
// or next hops), nothing for an unchanged one (:268-277), nothing for a route without next hops (CONNECTED, :283-287),
// RouteIpDel for what was installed and is gone (:303-310); adds in prefix order first, then the withdrawals.
using hspf::host::IbusMsg;            // (holo_spf_host.hpp: shared with the OSPF twin)
namespace detail {
inline std::vector<std::pair<int, std::string>> wire_nexthops(const std::vector<std::pair<std::string, std::string>> &nhs, const std::map<std::string, int> &ifindex) {
  std::vector<std::tuple<int, IpKey, std::string>> v;
  for (auto &n : nhs) { IpKey k = parse_ip(n.first); v.push_back({ifindex.at(n.second), k, n.first}); }
  std::sort(v.begin(), v.end(), [](auto &a, auto &b) { return std::tie(std::get<0>(a), std::get<1>(a)) < std::tie(std::get<0>(b), std::get<1>(b)); });
  std::vector<std::pair<int, std::string>> out;
  for (auto &t : v) out.push_back({std::get<0>(t), std::get<2>(t)});
  return out;
}
inline bool same_nexthops(std::vector<std::pair<std::string, std::string>> a, std::vector<std::pair<std::string, std::string>> b) {
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  return a == b;
}
}  // namespace detail

// Host form: new RIB rows (as compute_spf returns them) against the rows held before.
inline std::vector<IbusMsg> update_global_rib(const std::vector<RibRow> &new_rows, const std::vector<RibRow> &old_rows, const std::map<std::string, int> &ifindex) {
  std::map<IpKey, const RibRow *> old;
  for (auto &r : old_rows) old[parse_ip(r.prefix)] = &r;
  std::vector<std::pair<IpKey, const RibRow *>> fresh;
  for (auto &r : new_rows) fresh.push_back({parse_ip(r.prefix), &r});
  std::stable_sort(fresh.begin(), fresh.end(), [](auto &a, auto &b) { return a.first < b.first; });
  std::vector<IbusMsg> msgs;
  for (auto &kr : fresh) {
    const RibRow &r = *kr.second;
    auto it = old.find(kr.first);
    const RibRow *o = it == old.end() ? nullptr : it->second;
    if (it != old.end()) old.erase(it);
    if (o && o->metric == r.metric && detail::same_nexthops(o->nexthops, r.nexthops)) continue;
    if (!r.nexthops.empty()) msgs.push_back(IbusMsg{true, r.prefix, r.metric, detail::wire_nexthops(r.nexthops, ifindex)});
  }
  for (auto &kv : old)
    if (!kv.second->nexthops.empty()) msgs.push_back(IbusMsg{false, kv.second->prefix, 0, {}});
  return msgs;
}

// The record stream of hspf_routes_pack (one root) -> messages.  A record is a CANDIDATE: the device compared metric and
// first-hop slot masks, which is finer than the reference's comparison of next-hop sets (two slots may resolve to one
// adjacency, the host truncates to max-paths) — each is confirmed against the old row here, on the few records only.
inline std::vector<IbusMsg> expand_route_records(const RouteRecords &rec, const std::vector<std::string> &prefixes,
                                                 const std::map<uint32_t, std::shared_ptr<VertexNexthop>> &slot_nh,
                                                 const std::map<IpKey, const RibRow *> &old_rows, const std::map<std::string, int> &ifindex, uint32_t max_paths,
                                                 bool dels_need_old = false, const std::vector<IpKey> *keys = nullptr) {
  std::vector<IbusMsg> adds, dels;
  const uint32_t W = rec.mask_words;
  // every slot's next hop per address family, resolved ONCE (parsed address, interface index): a cold start expands 120 000
  // records over a handful of slots.  `keys`: the parsed prefixes when the caller holds them (PrefixTable::keys).
  struct SlotRes { bool has = false; IpKey key; std::string addr, ifname; int ifx = -1; };
  std::vector<SlotRes> res[2];
  res[0].resize((size_t)W * 64); res[1].resize((size_t)W * 64);
  for (auto &kv : slot_nh) {
    if (kv.first >= (size_t)W * 64) continue;
    for (int f = 0; f < 2; ++f) {
      const auto &addr = f ? kv.second->ipv6 : kv.second->ipv4;
      if (!addr) continue;
      SlotRes &sr = res[f][kv.first];
      sr.has = true; sr.key = parse_ip(*addr); sr.addr = *addr; sr.ifname = kv.second->iface_name.value_or("");
      auto xi = ifindex.find(sr.ifname);
      if (xi != ifindex.end()) sr.ifx = xi->second;
    }
  }
  std::vector<const SlotRes *> pick;
  std::vector<std::pair<std::string, std::string>> keep;
  adds.reserve(rec.count());
  for (size_t k = 0; k < rec.count(); ++k) {
    const uint32_t *r = rec.rec(k);
    const std::string &prefix = prefixes.at(r[1]);
    auto key_of = [&]() { return keys ? (*keys)[r[1]] : parse_ip(prefix); };
    if (r[2] == HSPF_DIFF_WITHDRAW) {
      if (r[4] != 0xFFFFFFFFu) continue;
      if (dels_need_old) { auto oi = old_rows.find(key_of()); if (oi == old_rows.end() || oi->second->nexthops.empty()) continue; }   // (was never installed)
      dels.push_back(IbusMsg{false, prefix, 0, {}});
      continue;
    }
    if (r[2] != HSPF_DIFF_INSTALL) continue;
    const bool v6 = keys ? (*keys)[r[1]].version == 6 : prefix.find(':') != std::string::npos;
    pick.clear();
    for (uint32_t w = 0; w < W; ++w) {
      uint64_t m = (uint64_t)r[HSPF_ROUTE_REC_WORDS + 2 * w] | ((uint64_t)r[HSPF_ROUTE_REC_WORDS + 2 * w + 1] << 32);
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        const SlotRes &sr = res[v6][w * 64 + b];
        if (sr.has) pick.push_back(&sr);
      }
    }
    // BTreeMap<IpAddr, _>: ascending address, a later slot with the same address replaces the earlier one; the first max-paths stay
    std::stable_sort(pick.begin(), pick.end(), [](const SlotRes *a, const SlotRes *b) { return a->key < b->key; });
    size_t np = 0;
    for (size_t i = 0; i < pick.size(); ++i) {
      if (i + 1 < pick.size() && pick[i + 1]->key == pick[i]->key) continue;
      pick[np++] = pick[i];
    }
    pick.resize(std::min<size_t>(np, max_paths));
    if (!old_rows.empty()) {
      auto oi = old_rows.find(key_of());
      if (oi != old_rows.end() && oi->second->metric == r[3]) {
        keep.clear();
        for (const SlotRes *q : pick) keep.push_back({q->addr, q->ifname});
        if (detail::same_nexthops(oi->second->nexthops, keep)) continue;       // the reference's "unchanged" (:268-277)
      }
    }
    // (a route whose next hops did not resolve replaces the old row WITHOUT a message, route.rs:283-301: the fresh route carries
    // no INSTALLED flag, so nothing is uninstalled either)
    if (pick.empty()) continue;
    // BTreeSet<Nexthop> order of the message: (ifindex, address)
    for (const SlotRes *q : pick) if (q->ifx < 0) throw std::out_of_range("expand_route_records: no ifindex for interface '" + q->ifname + "'");
    std::sort(pick.begin(), pick.end(), [](const SlotRes *a, const SlotRes *b) { return std::tie(a->ifx, a->key) < std::tie(b->ifx, b->key); });
    IbusMsg msg{true, prefix, r[3], {}};
    msg.nexthops.reserve(pick.size());
    for (const SlotRes *q : pick) msg.nexthops.push_back({q->ifx, q->addr});
    adds.push_back(std::move(msg));
  }
  adds.insert(adds.end(), dels.begin(), dels.end());
  return adds;
}

// compute_spf + update_global_rib with the SPT, the prefix attachment, the comparison with the RIB held before and the
// compaction of what changed ALL on the device; one record stream comes back.  One (level, topology) table — the shape of
// every IS-IS step fixture; the L1 / L2 merge of a two-level instance is host logic (compute_spf).  `n_records`: how many
// records crossed the bus.  Python twin: holo_amd.routes.update_global_rib_device.
inline std::vector<IbusMsg> update_global_rib_device(const Instance &inst, Engine &engine, const std::vector<RibRow> &rib_before,
                                                     const std::map<std::string, int> &ifindex, size_t *n_records = nullptr, size_t *n_prefixes = nullptr) {
  const InstanceCfg &cfg = inst.config;
  if (n_records) *n_records = 0;
  if (n_prefixes) *n_prefixes = 0;
  std::vector<std::pair<int, int>> tabs;
  for (int lv : cfg.levels()) for (int mt : {MT_STANDARD, MT_IPV6_UNICAST}) if (cfg.is_topology_enabled(mt)) tabs.push_back({lv, mt});
  if (tabs.size() != 1) throw std::invalid_argument("update_global_rib_device: one (level, topology) table only");
  const int level = tabs[0].first, mt_id = tabs[0].second;
  std::map<IpKey, const RibRow *> old_rows;
  for (auto &r : rib_before) old_rows[parse_ip(r.prefix)] = &r;
  LevelGraph g(inst, level, mt_id, false);
  auto ri = g.index.find(vertex_id(LanId{cfg.system_id, 0}));
  PrefixTable table;
  if (ri != g.index.end()) table = PrefixTable::build(inst, level, mt_id, g);
  if (ri == g.index.end() || table.prefixes.empty()) return update_global_rib({}, rib_before, ifindex);   // no SPT / nothing advertised
  // ONE prefix list for both sides: the table's prefixes plus those only the old RIB knows (no entries: no new route)
  std::map<IpKey, std::string> keys;
  for (auto &p : table.prefixes) keys[parse_ip(p)] = p;
  for (auto &kv : old_rows) keys.emplace(kv.first, kv.second->prefix);
  std::vector<std::string> prefixes;
  std::map<IpKey, uint32_t> where;
  for (auto &kv : keys) { where[kv.first] = (uint32_t)prefixes.size(); prefixes.push_back(kv.second); }
  const uint32_t P = (uint32_t)prefixes.size();
  std::vector<uint32_t> ptr(P + 1, 0);
  for (size_t j = 0; j < table.prefixes.size(); ++j) ptr[where[parse_ip(table.prefixes[j])] + 1] = table.pfx_ptr[j + 1] - table.pfx_ptr[j];
  for (uint32_t i = 0; i < P; ++i) ptr[i + 1] += ptr[i];               // (table.prefixes is sorted the same way: the entries keep their order)
  Graph &dev = g.device(engine);
  const uint32_t root = ri->second, n = g.n();
  auto run = engine.run_device(dev, {root}, g.run_flags);
  auto fresh = engine.routes_device(*run, ptr, table.pfx_vertex, table.pfx_metric, 0);
  // first-hop slots -> next hops (needs Interface / Adjacency objects: host, once per slot)
  auto res = std::make_shared<Tables>(run->host_tables());
  const uint32_t W = res->mask_words;
  detail::RunView r{res->dist.data(), res->hops.data(), res->flags.data(), res->mask.data(), W};
  std::function<RankKey(uint32_t)> rank;
  bool exact = false;
  for (uint32_t v = 0; v < n; ++v) exact |= (res->flags[v] & HSPF_RF_EXACT) != 0;
  std::shared_ptr<Tables> rr;
  if (exact) {
    rr = std::make_shared<Tables>(engine.run(dev, {root}, g.run_flags | HSPF_RUN_POP_RANK));
    rank = [rr](uint32_t v) { return RankKey{rr->pop_rank[v], 0, 0, 0}; };
  } else rank = [r](uint32_t v) { return RankKey{r.dist[v], v, 0, 0}; };
  const auto slot_nh = detail::slot_nexthops(g, engine.slot_table(dev, root), r, rank, true, level, inst);
  // the OLD RIB in the same index space: metric, and the slots whose next hop the old route used.  A next hop of the old
  // route that no slot resolves to any more (its adjacency is gone) cannot be expressed: the metric is poisoned so that
  // the pair compares unequal and the host decides.
  RoutesOut old;
  old.best_metric.assign(P, 0xFFFFFFFFu); old.best_entry.assign(P, 0xFFFFFFFFu); old.nexthop_mask.assign((size_t)P * W, 0);
  for (auto &kv : old_rows) {
    const uint32_t i = where[kv.first];
    const RibRow &row = *kv.second;
    const bool v6 = row.prefix.find(':') != std::string::npos;
    std::set<std::pair<std::string, std::string>> want(row.nexthops.begin(), row.nexthops.end()), seen;
    for (auto &sn : slot_nh) {
      const auto &addr = v6 ? sn.second->ipv6 : sn.second->ipv4;
      if (!addr) continue;
      const std::pair<std::string, std::string> key{*addr, sn.second->iface_name.value_or("")};
      if (want.count(key)) { old.nexthop_mask[(size_t)i * W + sn.first / 64] |= 1ull << (sn.first % 64); seen.insert(key); }
    }
    // (more next hops than max-paths allows NOW: the new route will be truncated on the host: not comparable by mask)
    old.best_metric[i] = (seen == want && want.size() <= cfg.max_paths) ? row.metric : 0xFFFFFFFEu;
    old.best_entry[i] = 0;
    bool any = false;
    for (uint32_t w = 0; w < W; ++w) any |= old.nexthop_mask[(size_t)i * W + w] != 0;
    if (!want.empty() && !any) old.nexthop_mask[(size_t)i * W] = 1;      // "it was installed with next hops" (the poisoned metric keeps the pair unequal)
  }
  auto before = engine.routes_upload(old, 1, P, W);
  const RouteRecords rec = engine.routes_changed(*before, *fresh);
  if (n_records) *n_records = rec.count();
  if (n_prefixes) *n_prefixes = P;
  return expand_route_records(rec, prefixes, slot_nh, old_rows, ifindex, cfg.max_paths);
}

// The same step for a RUNNING instance: everything that does not change with one LSP is kept — the level graph on the
// device (rows patched per changed LSP, LevelGraph::refresh), the prefix table on the device (HSPF_PFX_RESIDENT; rebuilt
// only when a changed LSP's prefixes differ), and the route tables of the PREVIOUS run as the "RIB held before" (device
// resident: nothing is uploaded, nothing but the changed records comes back — packed once from the new set and once from the
// old one, so that the host sees what each changed route WAS).  The first-hop slots are resolved again every step (which
// relaxations the root makes depends on distances elsewhere); when a slot comes to mean another next hop, or the prefix list
// changed, the old tables are void and the step compares against the installed routes, record by record.  Which routes ARE
// installed is read off the previous tables at that moment (settle_installed: rows with resolved next hops), not kept from
// the messages: a route that loses its next hops changes the RIB without a message (route.rs:283-301; HSPF_DIFF_SILENT, no
// record) and a set kept from messages would go stale — the random chains of tests/test_cpp_driver.py found exactly that.
// step() = trigger_lsps -> messages.  One (level, topology), local root; interfaces / adjacencies as at
// construction (an adjacency change is a new pipeline).
class RibPipeline {
 public:
  RibPipeline(const Instance &inst, Engine &engine, int level, int mt_id, const std::map<std::string, int> &ifindex)
      : engine_(engine), level_(level), mt_(mt_id), ifindex_(ifindex), graph_(std::make_unique<LevelGraph>(inst, level, mt_id, false, &engine)) {
    rebuild_tables(inst);
  }
  struct Timing { double refresh_ms = 0, run_ms = 0, routes_ms = 0, slots_ms = 0, diff_pack_ms = 0, expand_ms = 0; size_t records = 0; bool full = false; };
  Timing last;
  // First call: every route is new (the RIB before is empty).  Later calls: `changed` = LAN ids whose LSPs differ.
  std::vector<IbusMsg> step(const Instance &inst, const std::vector<LanId> &changed) {
    using C = std::chrono::steady_clock;
    auto ms = [](C::time_point a) { return std::chrono::duration<double, std::milli>(C::now() - a).count(); };
    last = Timing{};
    auto t = C::now();
    if (!changed.empty()) {
      if (!graph_->refresh(inst, changed)) { settle_installed(); graph_ = std::make_unique<LevelGraph>(inst, level_, mt_, false, &engine_); rebuild_tables(inst); last.full = true; }
      else {
        bool pfx = false;
        for (auto &lan : changed) pfx = pfx || prefixes_of(inst, lan) != sig_of(lan);
        if (pfx) { settle_installed(); rebuild_tables(inst); last.full = true; }
      }
    }
    last.refresh_ms = ms(t);
    const InstanceCfg &cfg = inst.config;
    auto ri = graph_->index.find(vertex_id(LanId{cfg.system_id, 0}));
    if (ri == graph_->index.end() || table_.prefixes.empty()) {
      // no root LSP (SPT = {root}, no routes: spf.rs:552-561, 866-869) or nothing advertised any more: the new RIB is empty — every
      // installed route is withdrawn, in RIB order (route.rs:303-310).  (Found by the random LSP changes of tests/test_cpp_driver.py:
      // until round 6 this returned no message and the routes stayed installed.)
      settle_installed();
      std::vector<IbusMsg> msgs;
      for (auto &kv : rib_) if (!kv.second.nexthops.empty()) msgs.push_back(IbusMsg{false, kv.second.prefix, 0, {}});
      rib_.clear();
      return msgs;
    }
    t = C::now();
    Graph &dev = graph_->device(engine_);
    const uint32_t root = ri->second;
    auto run = engine_.run_device(dev, {root}, graph_->run_flags);
    last.run_ms = ms(t); t = C::now();
    auto fresh = engine_.routes_device(*run, table_.pfx_ptr, table_.pfx_vertex, table_.pfx_metric, resident_ ? (uint32_t)HSPF_PFX_RESIDENT : 0u);
    resident_ = true;
    last.routes_ms = ms(t); t = C::now();
    {
      // first-hop slots -> next hops, every step: which relaxations the root makes, and in which order resolve_nexthop hands
      // out adjacencies, depends on distances elsewhere in the graph (spf.rs:680-701), not only on the root's own rows
      // (a view into page-locked memory the run keeps: the replay reads the root's two-hop neighbourhood, not 100 000 rows;
      // the masks are not needed here and stay on the device)
      const TablesView res = run->host_view(false);
      detail::RunView r{res.dist, res.hops, res.flags, res.mask, res.mask_words};
      std::function<RankKey(uint32_t)> rank = [r](uint32_t v) { return RankKey{r.dist[v], v, 0, 0}; };
      bool exact = false;
      for (uint32_t v = 0; v < graph_->n(); ++v) exact |= (res.flags[v] & HSPF_RF_EXACT) != 0;
      std::shared_ptr<Tables> rr;
      if (exact) { rr = std::make_shared<Tables>(engine_.run(dev, {root}, graph_->run_flags | HSPF_RUN_POP_RANK)); rank = [rr](uint32_t v) { return RankKey{rr->pop_rank[v], 0, 0, 0}; }; }
      auto nh = detail::slot_nexthops(*graph_, engine_.slot_table(dev, root), r, rank, true, level_, inst);
      if (prev_ && !same_slots(nh, slot_nh_)) settle_installed();      // a slot means another next hop now: the old masks are void
      slot_nh_ = std::move(nh);
    }
    last.slots_ms = ms(t); t = C::now();
    const uint32_t P = (uint32_t)table_.prefixes.size(), W = fresh->mask_words;
    const bool host_old = !prev_;
    if (host_old) {                                                   // nothing comparable on the device: the routes the host knows to be installed
      RoutesOut old;
      old.best_metric.assign(P, 0xFFFFFFFFu); old.best_entry.assign(P, 0xFFFFFFFFu); old.nexthop_mask.assign((size_t)P * W, 0);
      for (auto &kv : rib_) {
        const long wi = table_.find(kv.first);
        if (wi < 0) continue;
        old.best_metric[wi] = 0xFFFFFFFEu; old.best_entry[wi] = 0;                         // poisoned: every such pair comes back and the host decides
        if (!kv.second.nexthops.empty()) old.nexthop_mask[(size_t)wi * W] = 1;
      }
      prev_ = engine_.routes_upload(old, 1, P, W);
      last.full = true;
    }
    const RouteRecords rec = engine_.routes_changed(*prev_, *fresh);
    last.diff_pack_ms = ms(t); t = C::now();
    last.records = rec.count();
    // the route each record REPLACES: from the host's view after a reset, else from the old set's own record (metric + slot
    // masks resolved like the new ones: equal resolved next hops = the reference's "unchanged", route.rs:268-277)
    std::vector<RibRow> old_store;
    old_store.reserve(rec.count());
    std::map<IpKey, const RibRow *> old_rows;
    for (size_t k = 0; k < rec.count(); ++k) {
      const std::string &prefix = table_.prefixes.at(rec.rec(k)[1]);
      const IpKey &key = table_.keys[rec.rec(k)[1]];
      if (host_old) { auto it = rib_.find(key); if (it != rib_.end()) old_rows[key] = &it->second; continue; }
      const uint32_t *o = rec.old_rec(k);
      if (o[4] == 0xFFFFFFFFu) continue;                               // no route before
      RibRow row{prefix, o[3], level_, {}};
      const bool v6 = prefix.find(':') != std::string::npos;
      std::map<IpKey, std::pair<std::string, std::string>> nhs;
      for (uint32_t w = 0; w < W; ++w) {
        uint64_t m = (uint64_t)o[HSPF_ROUTE_REC_WORDS + 2 * w] | ((uint64_t)o[HSPF_ROUTE_REC_WORDS + 2 * w + 1] << 32);
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          auto it = slot_nh_.find(w * 64 + b);
          if (it == slot_nh_.end()) continue;
          const auto &addr = v6 ? it->second->ipv6 : it->second->ipv4;
          if (addr) nhs[parse_ip(*addr)] = {*addr, it->second->iface_name.value_or("")};
        }
      }
      for (auto &kv : nhs) { if (row.nexthops.size() >= cfg.max_paths) break; row.nexthops.push_back(kv.second); }
      old_store.push_back(std::move(row));
      old_rows[key] = &old_store.back();
    }
    std::vector<IbusMsg> msgs = expand_route_records(rec, table_.prefixes, slot_nh_, old_rows, ifindex_, cfg.max_paths, true, &table_.keys);
    // prefixes the host knows installed and the (rebuilt) table does not list any more
    if (host_old)
      for (auto &kv : rib_) if (table_.find(kv.first) < 0 && !kv.second.nexthops.empty()) msgs.push_back(IbusMsg{false, kv.second.prefix, 0, {}});
    // the withdrawals go out in the order of the OLD RIB (one pass over a BTreeMap, route.rs:303-310), whichever of the two sources
    // above a withdrawal came from
    {
      auto first_del = std::stable_partition(msgs.begin(), msgs.end(), [](const IbusMsg &m) { return m.add; });
      if (msgs.end() - first_del > 1) {
        std::vector<std::pair<IpKey, IbusMsg>> dels;
        for (auto it = first_del; it != msgs.end(); ++it) dels.push_back({parse_ip(it->prefix), std::move(*it)});
        std::stable_sort(dels.begin(), dels.end(), [](const std::pair<IpKey, IbusMsg> &a, const std::pair<IpKey, IbusMsg> &b) { return a.first < b.first; });
        for (size_t i = 0; i < dels.size(); ++i) *(first_del + i) = std::move(dels[i].second);
      }
    }
    rib_.clear();                                                     // (void from here on: settle_installed() reads it off `prev_` when it is needed)
    max_paths_ = cfg.max_paths;
    prev_ = std::move(fresh);
    last.expand_ms = ms(t);
    if (getenv("HSPF_TWIN_TIMING")) fprintf(stderr, "[twin pipeline step]  refresh %.2f run %.2f routes %.2f slots %.2f diff_pack %.2f expand %.2f ms (%zu records)\n", last.refresh_ms, last.run_ms, last.routes_ms, last.slots_ms, last.diff_pack_ms, last.expand_ms, last.records);
    return msgs;
  }
  // The installed routes (rows of the RIB that have next hops), read off the tables of the last step (one copy of the tables
  // to the host: for inspection and tests, not for the per-event path).
  std::map<IpKey, RibRow> rib() const { return read_installed(); }
  LevelGraph &graph() { return *graph_; }

 private:
  static bool same_slots(const std::map<uint32_t, std::shared_ptr<VertexNexthop>> &a, const std::map<uint32_t, std::shared_ptr<VertexNexthop>> &b) {
    if (a.size() != b.size()) return false;
    for (auto ia = a.begin(), ib = b.begin(); ia != a.end(); ++ia, ++ib)
      if (ia->first != ib->first || ia->second->system_id != ib->second->system_id || ia->second->iface_name != ib->second->iface_name ||
          ia->second->ipv4 != ib->second->ipv4 || ia->second->ipv6 != ib->second->ipv6) return false;
    return true;
  }
  using PfxSig = PrefixTable::PfxSig;
  const PfxSig &sig_of(const LanId &lan) const {                      // what the vertex attached when the tables were built (nothing: not a vertex then)
    static const PfxSig none;
    auto vi = graph_->index.find(vertex_id(lan));
    return vi == graph_->index.end() || vi->second >= pfx_sig_.size() ? none : pfx_sig_[vi->second];
  }
  PfxSig prefixes_of(const Instance &inst, const LanId &lan) const {
    PfxSig out;
    auto li = inst.lsdb.find(level_);
    if (li == inst.lsdb.end()) return out;
    const Lsp *z = li->second.zeroth_lsp(lan);
    if (!z) return out;
    const InstanceCfg &cfg = inst.config;
    const bool att = !cfg.att_ignore && z->att_bit(mt_) && !z->overload_bit(mt_);
    const bool v4 = cfg.ipv4_enabled && mt_ == MT_STANDARD;
    const bool v6 = cfg.ipv6_enabled && (mt_ == MT_STANDARD ? !cfg.is_topology_enabled(MT_IPV6_UNICAST) : true);
    for (auto &nw : vertex_networks(inst, level_, mt_, lan, att, inst.is_l2_attached_to_backbone(mt_), v4, v6)) out.push_back({nw.prefix, nw.metric, nw.external});
    return out;
  }
  void rebuild_tables(const Instance &inst) {
    table_ = PrefixTable::build(inst, level_, mt_, *graph_, &pfx_sig_);
    resident_ = false;
  }
  // The tables of the last step -> `rib_` = the rows that resolve to next hops (prefix, metric, next hops as the route held them:
  // ascending address, a later slot with the same address replaces the earlier one, the first max-paths), then the tables are
  // dropped.  One copy of the tables to the host (16 bytes per prefix and mask word) at the few events that void them.
  void settle_installed() {
    if (!prev_) return;                                               // (nothing since the last settle: rib_ stands)
    rib_ = read_installed();
    prev_.reset();
  }
  std::map<IpKey, RibRow> read_installed() const {
    if (!prev_) return rib_;
    std::map<IpKey, RibRow> rows;
    const RoutesOut o = prev_->host();
    const uint32_t W = prev_->mask_words;
    struct SlotRes { bool has = false; IpKey key; std::string addr, ifname; };
    std::vector<SlotRes> res[2];
    res[0].resize((size_t)W * 64); res[1].resize((size_t)W * 64);
    for (auto &kv : slot_nh_) {
      if (kv.first >= (size_t)W * 64) continue;
      for (int f = 0; f < 2; ++f) {
        const auto &addr = f ? kv.second->ipv6 : kv.second->ipv4;
        if (addr) res[f][kv.first] = SlotRes{true, parse_ip(*addr), *addr, kv.second->iface_name.value_or("")};
      }
    }
    std::vector<const SlotRes *> pick;
    const size_t P = std::min<size_t>(table_.prefixes.size(), o.best_entry.size());
    for (size_t p = 0; p < P; ++p) {
      if (o.best_entry[p] == 0xFFFFFFFFu) continue;
      const IpKey &key = table_.keys[p];
      pick.clear();
      for (uint32_t w = 0; w < W; ++w) {
        uint64_t m = o.nexthop_mask[p * W + w];
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          const SlotRes &sr = res[key.version == 6 ? 1 : 0][w * 64 + b];
          if (sr.has) pick.push_back(&sr);
        }
      }
      if (pick.empty()) continue;
      std::stable_sort(pick.begin(), pick.end(), [](const SlotRes *a, const SlotRes *b) { return a->key < b->key; });
      RibRow row{table_.prefixes[p], o.best_metric[p], level_, {}};
      for (size_t i = 0; i < pick.size() && row.nexthops.size() < max_paths_; ++i) {
        if (i + 1 < pick.size() && pick[i + 1]->key == pick[i]->key) continue;
        row.nexthops.push_back({pick[i]->addr, pick[i]->ifname});
      }
      rows.emplace_hint(rows.end(), key, std::move(row));             // (the table is in RIB order)
    }
    return rows;
  }
  Engine &engine_;
  int level_, mt_;
  std::map<std::string, int> ifindex_;
  std::unique_ptr<LevelGraph> graph_;
  PrefixTable table_;
  std::vector<PfxSig> pfx_sig_;                                       // by vertex index of graph_
  std::map<uint32_t, std::shared_ptr<VertexNexthop>> slot_nh_;
  std::unique_ptr<DeviceRoutes> prev_;
  std::map<IpKey, RibRow> rib_;                                       // installed routes as of the last settle_installed(); void while `prev_` holds tables
  uint32_t max_paths_ = 16;
  bool resident_ = false;
};

// ---- flooding::manet (holo-isis/src/flooding/manet.rs) ---------------------------------------------------------------
namespace flooding {

struct NeighborCache {                 // manet.rs:30-35
  Spt spt_hopcount;
  std::map<SystemId, std::string> remote_nbr_list;     // BTreeMap<SystemId, FloodingAlgo>
};

// init_cache (manet.rs:39-97): one hop-count SPT per Up adjacency — ONE batched engine run instead of a sequential
// loop — and, per neighbour, its remote neighbour list (the first hops of that SPT with the flooding algorithm each
// advertises; `flooding_algo_of(system id)` supplies the sub-TLV value, default zero-pruner as in :84).
inline std::map<SystemId, NeighborCache> init_cache(int level, const Instance &inst, Engine &engine,
                                                    const std::function<std::string(const SystemId &)> &flooding_algo_of = {}) {
  std::vector<SystemId> nbrs;
  for (const Interface *iface : inst.interfaces_by_name())
    for (auto &adj : iface->adjacencies)
      if (adj.state == "up" && std::find(nbrs.begin(), nbrs.end(), adj.system_id) == nbrs.end()) nbrs.push_back(adj.system_id);
  std::map<SystemId, NeighborCache> out;
  if (nbrs.empty()) return out;
  std::vector<Spt> spts = compute_spts(level, nbrs, false, std::nullopt, true, inst, engine);
  for (size_t i = 0; i < nbrs.size(); ++i) {
    NeighborCache c;
    c.spt_hopcount = std::move(spts[i]);
    for (const Vertex *v : c.spt_hopcount.first_hops())
      c.remote_nbr_list[v->id.lan_id.system_id] = flooding_algo_of ? flooding_algo_of(v->id.lan_id.system_id) : std::string("zero-pruner");
    out[nbrs[i]] = std::move(c);
  }
  return out;
}

// flood_reduction_hash (manet.rs:190-194): Fletcher-16 (`fletcher` crate: two running sums mod 255, sum2 << 8 | sum1) of
// the 8-byte LSP id with the fragment number shifted right by 3.  Vectors: manet.rs:205-232.
inline uint16_t flood_reduction_hash(const SystemId &system_id, uint8_t pseudonode, uint8_t fragment) {
  uint32_t s1 = 0, s2 = 0;
  auto feed = [&](uint8_t b) { s1 = (s1 + b) % 255; s2 = (s2 + s1) % 255; };
  for (uint8_t b : system_id) feed(b);
  feed(pseudonode);
  feed(fragment >> 3);
  return (uint16_t)((s2 << 8) | s1);
}

// reflood_list (manet.rs:99-173): the neighbours of transmitting neighbour `tn` this router has to re-flood the LSP to;
// every query is answered from the hop-count SPT of the batched run.  BTreeSet order.
inline std::vector<SystemId> reflood_list(std::map<SystemId, NeighborCache> &cache, const SystemId &local_system_id, const SystemId &tn,
                                          const SystemId &lsp_system_id, uint8_t lsp_pseudonode, uint8_t lsp_fragment) {
  auto ci = cache.find(tn);
  if (ci == cache.end() || ci->second.remote_nbr_list.empty()) return {};
  NeighborCache &c = ci->second;
  Spt &spt = c.spt_hopcount;
  std::set<SystemId> thl;                                                             // :120-132
  for (const Vertex *v : spt.second_hops()) {
    const SystemId &s = v->id.lan_id.system_id;
    if (s != lsp_system_id && !spt.is_on_path(s, lsp_system_id)) thl.insert(s);
  }
  std::vector<std::pair<SystemId, std::string>> rnl(c.remote_nbr_list.begin(), c.remote_nbr_list.end());
  const size_t rnum = rnl.size(), n0 = flood_reduction_hash(lsp_system_id, lsp_pseudonode, lsp_fragment) % rnum;     // :135-139
  std::vector<SystemId> out;
  for (size_t k = 0; k < rnum; ++k) {                                                   // circular from index N, :143-170
    if (thl.empty()) break;
    const auto &e = rnl[(n0 + k) % rnum];
    if (e.first == local_system_id) {
      for (auto &t : thl) if (spt.is_on_path(e.first, t)) out.push_back(t);
      break;
    }
    if (e.second != "modified-manet") continue;
    for (auto it = thl.begin(); it != thl.end();) it = spt.is_on_path(e.first, *it) ? thl.erase(it) : std::next(it);
  }
  return out;
}

inline bool should_flood(const Interface &iface, const std::vector<SystemId> &reflood) {     // manet.rs:176-186
  for (auto &a : iface.adjacencies)
    if (a.state == "up" && std::find(reflood.begin(), reflood.end(), a.system_id) != reflood.end()) return true;
  return false;
}

}  // namespace flooding

}  // namespace isis
}  // namespace host
}  // namespace hspf
