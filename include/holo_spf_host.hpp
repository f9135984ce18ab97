// holo_spf_host.hpp — what the C++ host-side twins (holo_spf_isis.hpp, holo_spf_ospf.hpp) share: the three engine calls
// they need, the product implementation of those calls on the C ABI, and IP prefix / address keys in the order of the
// reference's BTreeMap<IpNetwork, _> / BTreeMap<IpAddr, _>.
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "holo_spf_hip.h"

namespace hspf {
namespace host {

// ---- engine seen from the host side ------------------------------------------------------------------------------
struct Tables {                       // row-major [root][vertex] results of one run
  uint32_t n_roots = 0, n_vertices = 0, mask_words = 1;
  std::vector<uint32_t> dist, pop_rank;
  std::vector<uint16_t> hops, flags;
  std::vector<uint64_t> mask;
};
struct SlotTable { std::vector<uint32_t> vertex, base; uint32_t total = 0; };

class Graph {                         // one uploaded graph (device resident for the product engine)
 public:
  virtual ~Graph() = default;
};
// Results of one run left where the engine produced them (HBM for the product engine): input of routes().
// The same tables as plain pointers into memory the run object keeps (valid while it lives): what a caller needs that
// reads a handful of entries (the first-hop slot replay) and must not pay for four std::vector copies per step.
struct TablesView { const uint32_t *dist = nullptr; const uint16_t *hops = nullptr, *flags = nullptr; const uint64_t *mask = nullptr;
                    uint32_t n_roots = 0, n_vertices = 0, mask_words = 1; };
class DeviceRun {
 public:
  virtual ~DeviceRun() = default;
  virtual Tables host_tables() = 0;   // the copy the host side still needs for next-hop resolution
  // distance / hops / flags (and the masks when asked for) on the host, zero-copy for the caller; the default keeps a Tables
  virtual TablesView host_view(bool with_mask) {
    (void)with_mask;
    if (!view_cache_) view_cache_ = std::make_unique<Tables>(host_tables());
    return TablesView{view_cache_->dist.data(), view_cache_->hops.data(), view_cache_->flags.data(), view_cache_->mask.data(), n_roots, n_vertices, mask_words};
  }
  uint32_t n_roots = 0, n_vertices = 0, mask_words = 1;
 private:
  std::unique_ptr<Tables> view_cache_;
};
struct RoutesOut {                    // per (root, prefix), row-major: see hspf_routes in holo_spf_hip.h
  std::vector<uint32_t> best_metric, best_entry;
  std::vector<uint64_t> nexthop_mask;
};
// Route tables of one routes_device() call (or an uploaded set) left where the engine keeps them (HBM for the product
// engine): operands of the RIB comparison (hspf_routes_diff_device).
class DeviceRoutes {
 public:
  virtual ~DeviceRoutes() = default;
  virtual RoutesOut host() = 0;
  uint32_t n_roots = 0, n_prefixes = 0, mask_words = 1;
};
// The record stream of hspf_routes_pack: [count][HSPF_ROUTE_REC_WORDS + 2 mask_words] u32 (root, prefix, action, metric, entry, 0, mask words lo/hi)
struct RouteRecords {
  uint32_t mask_words = 1;
  std::vector<uint32_t> words;
  std::vector<uint32_t> old_words;     // the same records packed from the OLD set (metric, entry, masks of the route held before)
  const uint32_t *old_rec(size_t k) const { return old_words.data() + k * stride(); }
  size_t stride() const { return HSPF_ROUTE_REC_WORDS + 2u * mask_words; }
  size_t count() const { return words.size() / stride(); }
  const uint32_t *rec(size_t k) const { return words.data() + k * stride(); }
};
class Engine {
 public:
  virtual ~Engine() = default;
  virtual std::unique_ptr<Graph> upload(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col,
                                        const std::vector<uint32_t> &metric, const std::vector<uint8_t> &vflags,
                                        uint32_t max_path_metric) = 0;
  // LSDB records -> CSR on the engine (hspf_graph_upload_keyed): vertices by 64-bit key in any order, links as
  // (target key, cost), targets unresolved.  nullptr: the engine has no such path (the caller derives the CSR itself and calls
  // upload).  On success csr_* hold the CSR the engine built (vertex index = rank of the key; links to absent keys dropped) and
  // rank[i] the index of input vertex i.
  virtual std::unique_ptr<Graph> upload_keyed(const std::vector<uint64_t> &, const std::vector<uint32_t> &, const std::vector<uint64_t> &,
                                              const std::vector<uint32_t> &, const std::vector<uint8_t> &, uint32_t, std::vector<uint32_t> &,
                                              std::vector<uint32_t> &, std::vector<uint32_t> &, std::vector<uint32_t> &, std::vector<uint8_t> &) { return nullptr; }
  virtual Tables run(Graph &g, const std::vector<uint32_t> &roots, uint32_t run_flags) = 0;
  virtual SlotTable slot_table(Graph &g, uint32_t root) = 0;
  // whole rows replaced (hspf_graph_patch): vertices strictly ascending, rows[i] = (col, metric) of vertices[i]
  virtual void patch(Graph &g, const std::vector<uint32_t> &vertices,
                     const std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> &rows,
                     const std::vector<uint8_t> &vflags) = 0;
  // device-resident variant (hspf_run_device) + prefix attachment on the device (hspf_routes_device; table CSR by
  // prefix, flags = HSPF_PFX_*)
  virtual std::unique_ptr<DeviceRun> run_device(Graph &g, const std::vector<uint32_t> &roots, uint32_t run_flags) = 0;
  virtual RoutesOut routes(DeviceRun &run, const std::vector<uint32_t> &pfx_ptr, const std::vector<uint32_t> &pfx_vertex,
                           const std::vector<uint32_t> &pfx_metric, uint32_t flags) = 0;
  // The wire step on the device: the same attachment with the tables LEFT on the device, a host table set
  // brought there (the RIB held before), and the comparison + ordered compaction + ONE packed copy of what changed
  // (hspf_routes_diff_device + hspf_routes_pack).  flags may carry HSPF_PFX_RESIDENT (same vectors as the previous call).
  virtual std::unique_ptr<DeviceRoutes> routes_device(DeviceRun &run, const std::vector<uint32_t> &pfx_ptr, const std::vector<uint32_t> &pfx_vertex,
                                                      const std::vector<uint32_t> &pfx_metric, uint32_t flags) = 0;
  virtual std::unique_ptr<DeviceRoutes> routes_upload(const RoutesOut &t, uint32_t n_roots, uint32_t n_prefixes, uint32_t mask_words) = 0;
  virtual RouteRecords routes_changed(DeviceRoutes &old_set, DeviceRoutes &new_set) = 0;
  // Several areas, ONE RIB (hspf_rib_clear_device / hspf_rib_fold_device): an empty instance-wide state over `n_prefixes`
  // prefixes and `mask_words` words of instance-wide first-hop slots, and the ordered fold (HSPF_PFX_ORDERED rules) of one
  // area's table into it — `prefix_map`: area prefix -> instance prefix, the area's slots start at word `word_offset`.
  virtual std::unique_ptr<DeviceRoutes> rib_new(uint32_t n_prefixes, uint32_t mask_words) = 0;
  virtual void rib_fold(DeviceRoutes &rib, DeviceRun &run, const std::vector<uint32_t> &pfx_ptr, const std::vector<uint32_t> &pfx_vertex,
                        const std::vector<uint32_t> &pfx_metric, const std::vector<uint32_t> &pfx_origin, const std::vector<uint32_t> &prefix_map,
                        uint32_t area_index, uint32_t word_offset) = 0;
};

// CSR with the rows of `vertices` (strictly ascending) replaced — the host-side twin of hspf_graph_patch.
inline void splice_rows(std::vector<uint32_t> &row_ptr, std::vector<uint32_t> &col, std::vector<uint32_t> &metric,
                        std::vector<uint8_t> &vflags, const std::vector<uint32_t> &vertices,
                        const std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> &rows,
                        const std::vector<uint8_t> &new_flags) {
  const uint32_t n = (uint32_t)row_ptr.size() - 1;
  if (!vertices.empty() && vertices.size() <= 16) {
    // a handful of rows (the everyday LSP change): IN PLACE, last row first so that the offsets in front stay valid — the tail
    // behind a row moves once by its change in length, nothing is allocated; then one pass over the row starts behind the first
    std::vector<long> d(vertices.size());
    for (size_t j = vertices.size(); j-- > 0;) {
      const uint32_t u = vertices[j];
      const size_t a = row_ptr[u], old_len = row_ptr[u + 1] - a, new_len = rows[j].first.size();
      d[j] = (long)new_len - (long)old_len;
      if (new_len > old_len) {
        col.insert(col.begin() + a + old_len, new_len - old_len, 0u);
        metric.insert(metric.begin() + a + old_len, new_len - old_len, 0u);
      } else if (new_len < old_len) {
        col.erase(col.begin() + a + new_len, col.begin() + a + old_len);
        metric.erase(metric.begin() + a + new_len, metric.begin() + a + old_len);
      }
      std::copy(rows[j].first.begin(), rows[j].first.end(), col.begin() + a);
      std::copy(rows[j].second.begin(), rows[j].second.end(), metric.begin() + a);
      vflags[u] = new_flags[j];
    }
    long shift = 0;
    size_t j = 0;
    for (uint32_t u = vertices[0]; u <= n; ++u) {                        // row u starts behind every replaced row with a lower index
      while (j < vertices.size() && vertices[j] < u) shift += d[j++];
      row_ptr[u] = (uint32_t)((long)row_ptr[u] + shift);
    }
    return;
  }
  // many rows: the stretches BETWEEN the replaced rows move as blocks into exactly sized arrays (not a million appends)
  long delta = 0;
  for (size_t j = 0; j < vertices.size(); ++j) delta += (long)rows[j].first.size() - (long)(row_ptr[vertices[j] + 1] - row_ptr[vertices[j]]);
  const size_t new_len = (size_t)((long)col.size() + delta);
  std::vector<uint32_t> ncol(new_len), nmet(new_len), nrp(n + 1);
  size_t out = 0;                      // write position in the new arrays
  uint32_t from = 0;                   // first vertex of the pending untouched stretch
  auto copy_stretch = [&](uint32_t to) {                                   // rows [from, to) unchanged: one block, offsets shifted
    if (to <= from) return;
    const uint32_t a = row_ptr[from], b = row_ptr[to];
    std::copy(col.begin() + a, col.begin() + b, ncol.begin() + out);
    std::copy(metric.begin() + a, metric.begin() + b, nmet.begin() + out);
    const long shift = (long)out - (long)a;
    for (uint32_t u = from; u < to; ++u) nrp[u] = (uint32_t)((long)row_ptr[u] + shift);
    out += b - a;
  };
  for (size_t j = 0; j < vertices.size(); ++j) {
    const uint32_t u = vertices[j];
    copy_stretch(u);
    nrp[u] = (uint32_t)out;
    std::copy(rows[j].first.begin(), rows[j].first.end(), ncol.begin() + out);
    std::copy(rows[j].second.begin(), rows[j].second.end(), nmet.begin() + out);
    out += rows[j].first.size();
    vflags[u] = new_flags[j];
    from = u + 1;
  }
  copy_stretch(n);
  nrp[n] = (uint32_t)out;
  row_ptr.swap(nrp); col.swap(ncol); metric.swap(nmet);
}

// ---- addresses and prefixes (BTreeMap<IpNetwork, _> / BTreeMap<IpAddr, _> order) -----------------------------------
struct IpKey {                        // (version, 128-bit address, prefix length)
  int version = 4;
  std::array<uint8_t, 16> addr{};
  int len = 0;
  bool operator<(const IpKey &o) const { return std::tie(version, addr, len) < std::tie(o.version, o.addr, o.len); }
  bool operator==(const IpKey &o) const { return version == o.version && addr == o.addr && len == o.len; }
};
// "a.b.c.d[/len]" without a heap allocation (the general routine below makes six strings per address: a quarter of the compiled
// twin's route computation at 120 000 prefixes was this).  false: not plain dotted-quad text — the caller takes the general routine.
inline bool parse_ip4_fast(const std::string &text, IpKey &k) {
  const char *p = text.data(), *end = p + text.size();
  uint32_t v = 0;
  for (int i = 0; i < 4; ++i) {
    uint32_t o = 0; int nd = 0;
    while (p != end && *p >= '0' && *p <= '9' && nd < 3) { o = o * 10u + (uint32_t)(*p - '0'); ++p; ++nd; }
    if (nd == 0 || o > 255u) return false;
    v = (v << 8) | o;
    if (i < 3) { if (p == end || *p != '.') return false; ++p; }
  }
  int len = 32;
  if (p != end) {
    if (*p != '/') return false;
    ++p;
    int nd = 0; len = 0;
    while (p != end && *p >= '0' && *p <= '9' && nd < 2) { len = len * 10 + (*p - '0'); ++p; ++nd; }
    if (nd == 0 || p != end || len > 32) return false;
  }
  if (len < 32) v = len == 0 ? 0u : (v & (0xFFFFFFFFu << (32 - len)));       // network address: host bits cleared (strict = false)
  k.version = 4; k.addr.fill(0); k.len = len;
  k.addr[12] = (uint8_t)(v >> 24); k.addr[13] = (uint8_t)(v >> 16); k.addr[14] = (uint8_t)(v >> 8); k.addr[15] = (uint8_t)v;
  return true;
}
inline IpKey parse_ip(const std::string &text) {             // "a.b.c.d[/len]" or IPv6 text form (with "::")
  IpKey k;
  if (parse_ip4_fast(text, k)) return k;
  k = IpKey{};
  std::string a = text;
  const size_t slash = a.find('/');
  int len = -1;
  if (slash != std::string::npos) { len = std::stoi(a.substr(slash + 1)); a = a.substr(0, slash); }
  if (a.find(':') == std::string::npos) {
    k.version = 4;
    size_t pos = 0;
    for (int i = 0; i < 4; ++i) {
      const size_t dot = a.find('.', pos);
      k.addr[12 + i] = (uint8_t)std::stoi(a.substr(pos, dot == std::string::npos ? std::string::npos : dot - pos));
      pos = dot == std::string::npos ? a.size() : dot + 1;
    }
    k.len = len < 0 ? 32 : len;
    if (k.len < 32) {                                        // network address: host bits cleared (strict = false)
      uint32_t v = ((uint32_t)k.addr[12] << 24) | ((uint32_t)k.addr[13] << 16) | ((uint32_t)k.addr[14] << 8) | k.addr[15];
      v = k.len == 0 ? 0 : (v & (0xFFFFFFFFu << (32 - k.len)));
      k.addr[12] = v >> 24; k.addr[13] = v >> 16; k.addr[14] = v >> 8; k.addr[15] = v;
    }
    return k;
  }
  k.version = 6;
  std::vector<uint16_t> head, tail;
  bool in_tail = false;
  size_t pos = 0;
  if (a.rfind("::", 0) == 0) { in_tail = true; pos = 2; }
  while (pos < a.size()) {
    size_t c = a.find(':', pos);
    std::string grp = a.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
    if (grp.empty()) { in_tail = true; pos = c + 1; continue; }
    (in_tail ? tail : head).push_back((uint16_t)std::stoul(grp, nullptr, 16));
    if (c == std::string::npos) break;
    if (c + 1 < a.size() && a[c + 1] == ':') { in_tail = true; pos = c + 2; } else pos = c + 1;
  }
  std::vector<uint16_t> g(8, 0);
  for (size_t i = 0; i < head.size() && i < 8; ++i) g[i] = head[i];
  for (size_t i = 0; i < tail.size() && i < 8; ++i) g[8 - tail.size() + i] = tail[i];
  for (int i = 0; i < 8; ++i) { k.addr[2 * i] = g[i] >> 8; k.addr[2 * i + 1] = g[i] & 0xFF; }
  k.len = len < 0 ? 128 : len;
  for (int bit = k.len; bit < 128; ++bit) k.addr[bit / 8] &= ~(uint8_t)(0x80u >> (bit % 8));
  return k;
}


// ---- the wire step's message (what goes on the ibus per route: RouteIpAdd / RouteIpDel) --------------------------------
struct IbusMsg {
  bool add = true;
  std::string prefix;
  uint32_t metric = 0;
  std::vector<std::pair<int, std::string>> nexthops;       // (ifindex, address), as the reference's BTreeSet<Nexthop> orders them
  bool operator==(const IbusMsg &o) const { return add == o.add && prefix == o.prefix && (!add || (metric == o.metric && nexthops == o.nexthops)); }
};

// ---- the product engine: libholo_spf_hip.so through the C ABI ---------------------------------------------------------
class HipGraph : public Graph {
 public:
  HipGraph(hspf_ctx *c, hspf_graph *g) : ctx(c), g(g) {}
  ~HipGraph() override { if (g) hspf_graph_free(ctx, g); }
  hspf_ctx *ctx;
  hspf_graph *g;
};
// Device and page-locked host blocks of the objects below, kept for reuse: a running instance makes the same few allocations
// every SPF event (run tables, route tables, staging), and hipMalloc / hipFree cost 10-50 us each and synchronise the device
// (the LSP-change pipeline of RibPipeline: 0.94 -> ~0.7 ms, profiles/r05_notes.md).  Blocks are handed out by exact size class
// (rounded up to 64 KB); every engine call that reads or writes them is synchronous, so a returned block is idle.  Like the
// engine and its context the pool belongs to ONE thread (the protocol instance's): no locking.
class HipPool {
 public:
  ~HipPool() {
    for (auto &kv : dev_) for (void *p : kv.second) (void)hipFree(p);
    for (auto &kv : pin_) for (void *p : kv.second) (void)hipHostFree(p);
  }
  static size_t cls(size_t bytes) { return (std::max<size_t>(bytes, 1) + 65535) & ~size_t(65535); }
  void *dev(size_t bytes) {
    auto &fl = dev_[cls(bytes)];
    if (!fl.empty()) { void *p = fl.back(); fl.pop_back(); return p; }
    void *p = nullptr;
    if (hipMalloc(&p, cls(bytes)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
    return p;
  }
  void dev_free(void *p, size_t bytes) { if (p) dev_[cls(bytes)].push_back(p); }
  void *pinned(size_t bytes) {
    auto &fl = pin_[cls(bytes)];
    if (!fl.empty()) { void *p = fl.back(); fl.pop_back(); return p; }
    void *p = nullptr;
    if (hipHostMalloc(&p, cls(bytes), hipHostMallocDefault) != hipSuccess) throw std::runtime_error("hipHostMalloc failed");
    return p;
  }
  void pinned_free(void *p, size_t bytes) { if (p) pin_[cls(bytes)].push_back(p); }
 private:
  std::map<size_t, std::vector<void *>> dev_, pin_;
};
class HipDeviceRun : public DeviceRun {             // dist / hops / flags / masks of one run in device buffers of the engine's pool
 public:
  explicit HipDeviceRun(std::shared_ptr<HipPool> pool) : pool_(std::move(pool)) {}
  ~HipDeviceRun() override {
    pool_->dev_free(dist, rn() * 4); pool_->dev_free(hops, rn() * 2); pool_->dev_free(flags, rn() * 2); pool_->dev_free(mask, rn() * 8 * mask_words);
    pool_->pinned_free(stage_, stage_bytes_);
  }
  void alloc() {
    dist = (uint32_t *)pool_->dev(rn() * 4); hops = (uint16_t *)pool_->dev(rn() * 2); flags = (uint16_t *)pool_->dev(rn() * 2);
    mask = (uint64_t *)pool_->dev(rn() * 8 * mask_words);
  }
  Tables host_tables() override {
    Tables t;
    t.n_roots = n_roots; t.n_vertices = n_vertices; t.mask_words = mask_words;
    t.dist.resize(rn()); t.hops.resize(rn()); t.flags.resize(rn()); t.mask.resize(rn() * mask_words);
    if (hipMemcpy(t.dist.data(), dist, rn() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(t.hops.data(), hops, rn() * 2, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(t.flags.data(), flags, rn() * 2, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(t.mask.data(), mask, rn() * 8 * mask_words, hipMemcpyDeviceToHost) != hipSuccess)
      throw std::runtime_error("hipMemcpy of the run tables failed");
    return t;
  }
  // into ONE page-locked block of the pool (bus speed, no std::vector to size and fill): [mask (when asked for) | dist | hops | flags]
  TablesView host_view(bool with_mask) override {
    const size_t mb = with_mask ? rn() * 8 * mask_words : 0, need = mb + rn() * 8;
    if (!stage_ || stage_bytes_ < need || (with_mask && !stage_mask_)) {
      pool_->pinned_free(stage_, stage_bytes_);
      stage_ = (char *)pool_->pinned(need); stage_bytes_ = need; stage_mask_ = with_mask; staged_ = false;
    }
    char *pm = stage_, *pd = stage_ + (stage_mask_ ? rn() * 8 * mask_words : 0), *ph = pd + rn() * 4, *pf = ph + rn() * 2;
    if (!staged_) {
      bool ok = hipMemcpy(pd, dist, rn() * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(ph, hops, rn() * 2, hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(pf, flags, rn() * 2, hipMemcpyDeviceToHost) == hipSuccess;
      if (ok && stage_mask_) ok = hipMemcpy(pm, mask, rn() * 8 * mask_words, hipMemcpyDeviceToHost) == hipSuccess;
      if (!ok) throw std::runtime_error("hipMemcpy of the run tables failed");
      staged_ = true;
    }
    return TablesView{(const uint32_t *)pd, (const uint16_t *)ph, (const uint16_t *)pf, stage_mask_ ? (const uint64_t *)pm : nullptr, n_roots, n_vertices, mask_words};
  }
  uint32_t *dist = nullptr; uint16_t *hops = nullptr, *flags = nullptr; uint64_t *mask = nullptr;
 private:
  size_t rn() const { return (size_t)n_roots * n_vertices; }
  std::shared_ptr<HipPool> pool_;
  char *stage_ = nullptr; size_t stage_bytes_ = 0; bool stage_mask_ = false, staged_ = false;
};
class HipDeviceRoutes : public DeviceRoutes {       // best_metric / best_entry / nexthop_mask of one table set in device buffers of the engine's pool
 public:
  explicit HipDeviceRoutes(std::shared_ptr<HipPool> pool) : pool_(std::move(pool)) {}
  ~HipDeviceRoutes() override {
    pool_->dev_free(bm, rp() * 4); pool_->dev_free(be, rp() * 4); pool_->dev_free(nm, rp() * 8 * mask_words); pool_->dev_free(org, std::max<size_t>(n_prefixes, 1) * 4);
  }
  RoutesOut host() override {
    RoutesOut o;
    const size_t n = (size_t)n_roots * n_prefixes;
    o.best_metric.resize(n); o.best_entry.resize(n); o.nexthop_mask.resize(n * mask_words);
    if (n && (hipMemcpy(o.best_metric.data(), bm, n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(o.best_entry.data(), be, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
              hipMemcpy(o.nexthop_mask.data(), nm, n * 8 * mask_words, hipMemcpyDeviceToHost) != hipSuccess))
      throw std::runtime_error("hipMemcpy of the route tables failed");
    return o;
  }
  bool alloc() {
    bm = (uint32_t *)pool_->dev(rp() * 4); be = (uint32_t *)pool_->dev(rp() * 4); nm = (uint64_t *)pool_->dev(rp() * 8 * mask_words);
    return true;
  }
  void alloc_origin() { org = (uint32_t *)pool_->dev(std::max<size_t>(n_prefixes, 1) * 4); }
  hspf_routes raw() const { return hspf_routes{bm, be, nm}; }
  uint32_t *bm = nullptr, *be = nullptr; uint64_t *nm = nullptr;
  uint32_t *org = nullptr;             // rib_new only: the owners' origins (hspf_rib_device.origin)
 private:
  size_t rp() const { return std::max<size_t>((size_t)n_roots * n_prefixes, 1); }
  std::shared_ptr<HipPool> pool_;
};
class HipEngine : public Engine {
 public:
  explicit HipEngine(int device = 0) {
    const int rc = hspf_init(device, &ctx_);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_init: ") + hspf_strerror(rc));   // no CPU fallback
  }
  ~HipEngine() override { if (diff_) (void)hipFree(diff_); if (pin_) hspf_host_free(ctx_, pin_); if (ctx_) hspf_shutdown(ctx_); }
  hspf_ctx *raw() const { return ctx_; }
  std::unique_ptr<Graph> upload(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col,
                                const std::vector<uint32_t> &metric, const std::vector<uint8_t> &vflags,
                                uint32_t max_path_metric) override {
    hspf_csr csr{(uint32_t)vflags.size(), (uint32_t)col.size(), row_ptr.data(), col.data(), metric.data(), vflags.data(), max_path_metric};
    hspf_graph *g = nullptr;
    const int rc = hspf_graph_upload(ctx_, &csr, &g);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_graph_upload: ") + hspf_last_error(ctx_));
    return std::make_unique<HipGraph>(ctx_, g);
  }
  std::unique_ptr<Graph> upload_keyed(const std::vector<uint64_t> &vkey, const std::vector<uint32_t> &vrow, const std::vector<uint64_t> &tkey,
                                      const std::vector<uint32_t> &tmet, const std::vector<uint8_t> &vfl, uint32_t max_path_metric, std::vector<uint32_t> &rank,
                                      std::vector<uint32_t> &row_ptr, std::vector<uint32_t> &col, std::vector<uint32_t> &metric, std::vector<uint8_t> &vflags) override {
    const uint32_t n = (uint32_t)vkey.size();
    hspf_keyed_lsdb k{n, (uint32_t)tkey.size(), vkey.data(), vrow.data(), tkey.data(), tmet.data(), vfl.data(), max_path_metric};
    hspf_graph *g = nullptr;
    rank.resize(n);
    const int rc = hspf_graph_upload_keyed(ctx_, &k, &g, rank.data());
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_graph_upload_keyed: ") + hspf_last_error(ctx_));
    auto out = std::make_unique<HipGraph>(ctx_, g);
    // the CSR as the engine built it: the twins' slot replay, refresh and Spt queries walk it on the host
    auto fetch = [&](uint32_t which, void *dst, size_t bytes) {
      size_t got = 0;
      if (hspf_graph_export(ctx_, g, which, dst, bytes, &got) != HSPF_OK || got != bytes) throw std::runtime_error(std::string("hspf_graph_export: ") + hspf_last_error(ctx_));
    };
    const uint32_t e = hspf_graph_n_edges(g);
    row_ptr.resize((size_t)n + 1); col.resize(e); metric.resize(e); vflags.resize(n);
    fetch(HSPF_GX_ROW_PTR, row_ptr.data(), ((size_t)n + 1) * 4);
    if (e) { fetch(HSPF_GX_COL, col.data(), (size_t)e * 4); fetch(HSPF_GX_METRIC, metric.data(), (size_t)e * 4); }
    fetch(HSPF_GX_VFLAGS, vflags.data(), n);
    return out;
  }
  // One run, tables on the host.  Since ABI 7 through the PACKED hand-off (hspf_run_packed: one word per (root, vertex) into a
  // page-locked buffer the engine keeps, a quarter of hspf_run's bytes over the bus) and decoded into the twins' tables
  // here; runs whose results do not fit packed words (more than 24 first-hop slots), and runs that ask for the pop order,
  // take hspf_run as before.  `last_handoff` says which way the last run went and what it cost.
  struct Handoff { bool packed = false; uint32_t word_bytes = 0; double run_ms = 0, alloc_ms = 0, decode_ms = 0; };   // alloc: the twins' own table vectors
  Handoff last_handoff;
  Tables run(Graph &gr, const std::vector<uint32_t> &roots, uint32_t run_flags) override {
    hspf_graph *g = static_cast<HipGraph &>(gr).g;
    Tables t;
    t.n_roots = (uint32_t)roots.size();
    t.n_vertices = hspf_graph_n_vertices(g);
    const size_t rn = (size_t)t.n_roots * t.n_vertices;
    last_handoff = Handoff{};
    const auto t0 = std::chrono::steady_clock::now();
    if (!(run_flags & HSPF_RUN_POP_RANK) && use_packed) {
      if (pin_cap_ < rn * 8) {
        if (pin_) hspf_host_free(ctx_, pin_);
        pin_ = nullptr; pin_cap_ = 0;
        if (hspf_host_alloc(ctx_, rn * 8, &pin_) != HSPF_OK) throw std::runtime_error(std::string("hspf_host_alloc: ") + hspf_last_error(ctx_));
        pin_cap_ = rn * 8;
      }
      hspf_packed_layout ly{};
      std::vector<uint8_t> status(t.n_roots, 0);
      const int rc = hspf_run_packed(ctx_, g, roots.data(), t.n_roots, run_flags, pin_, pin_cap_, &ly, status.data());
      if (rc == HSPF_OK) {
        const auto t1 = std::chrono::steady_clock::now();
        t.mask_words = 1;
        t.dist.resize(rn); t.hops.resize(rn); t.flags.resize(rn); t.mask.resize(rn);
        const auto t1b = std::chrono::steady_clock::now();
        for (uint32_t r = 0; r < t.n_roots; ++r) {
          const uint16_t ex = (status[r] & HSPF_ROOT_EXACT) ? (uint16_t)HSPF_RF_EXACT : (uint16_t)0;
          const size_t o = (size_t)r * t.n_vertices;
          if (ly.word_bytes == 4) decode_row<uint32_t>((const uint32_t *)pin_ + o, ly, ex, t, o);
          else decode_row<uint64_t>((const uint64_t *)pin_ + o, ly, ex, t, o);
        }
        const auto t2 = std::chrono::steady_clock::now();
        last_handoff = Handoff{true, ly.word_bytes, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t1b - t1).count(),
                               std::chrono::duration<double, std::milli>(t2 - t1b).count()};
        return t;
      }
      if (rc != HSPF_E_NO_PACKED) throw std::runtime_error(std::string("hspf_run_packed: ") + hspf_last_error(ctx_));
    }
    int rc = hspf_mask_words(ctx_, g, roots.data(), t.n_roots, &t.mask_words);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_mask_words: ") + hspf_last_error(ctx_));
    t.dist.resize(rn); t.hops.resize(rn); t.flags.resize(rn); t.mask.resize(rn * t.mask_words);
    if (run_flags & HSPF_RUN_POP_RANK) t.pop_rank.resize(rn);
    hspf_result out{t.dist.data(), t.hops.data(), t.flags.data(), t.mask.data(), t.mask_words,
                    (run_flags & HSPF_RUN_POP_RANK) ? t.pop_rank.data() : nullptr};
    rc = hspf_run(ctx_, g, roots.data(), t.n_roots, run_flags, &out);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_run: ") + hspf_last_error(ctx_));
    last_handoff.run_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return t;
  }
  bool use_packed = true;              // false: hspf_run for every run (the A/B of tests and of the end-to-end timing)
  SlotTable slot_table(Graph &gr, uint32_t root) override {
    hspf_graph *g = static_cast<HipGraph &>(gr).g;
    SlotTable st;
    const int cnt = hspf_slot_table(ctx_, g, root, nullptr, nullptr, 0, &st.total);
    if (cnt < 0) throw std::runtime_error(std::string("hspf_slot_table: ") + hspf_last_error(ctx_));
    st.vertex.resize(cnt); st.base.resize(cnt);
    hspf_slot_table(ctx_, g, root, st.vertex.data(), st.base.data(), (uint32_t)cnt, &st.total);
    return st;
  }
  void patch(Graph &gr, const std::vector<uint32_t> &vertices,
             const std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> &rows, const std::vector<uint8_t> &vflags) override {
    std::vector<uint32_t> rp{0}, c, m;
    for (auto &r : rows) { c.insert(c.end(), r.first.begin(), r.first.end()); m.insert(m.end(), r.second.begin(), r.second.end()); rp.push_back((uint32_t)c.size()); }
    if (c.empty()) { c.push_back(0); m.push_back(0); }              // non-NULL pointers for an all-empty delta
    hspf_rows d{(uint32_t)vertices.size(), vertices.data(), rp.data(), c.data(), m.data(), vflags.data()};
    const int rc = hspf_graph_patch(ctx_, static_cast<HipGraph &>(gr).g, &d);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_graph_patch: ") + hspf_last_error(ctx_));
  }
  std::unique_ptr<DeviceRun> run_device(Graph &gr, const std::vector<uint32_t> &roots, uint32_t run_flags) override {
    hspf_graph *g = static_cast<HipGraph &>(gr).g;
    auto r = std::make_unique<HipDeviceRun>(pool_);
    r->n_roots = (uint32_t)roots.size(); r->n_vertices = hspf_graph_n_vertices(g);
    int rc = hspf_mask_words(ctx_, g, roots.data(), r->n_roots, &r->mask_words);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_mask_words: ") + hspf_last_error(ctx_));
    r->alloc();
    hspf_result out{r->dist, r->hops, r->flags, r->mask, r->mask_words, nullptr};
    rc = hspf_run_device(ctx_, g, roots.data(), r->n_roots, run_flags, &out);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_run_device: ") + hspf_last_error(ctx_));
    return r;
  }
  RoutesOut routes(DeviceRun &run, const std::vector<uint32_t> &pfx_ptr, const std::vector<uint32_t> &pfx_vertex,
                   const std::vector<uint32_t> &pfx_metric, uint32_t flags) override {
    auto &r = static_cast<HipDeviceRun &>(run);
    const uint32_t P = (uint32_t)pfx_ptr.size() - 1;
    RoutesOut o;
    const size_t rp = (size_t)r.n_roots * P;
    o.best_metric.assign(rp, 0xFFFFFFFFu); o.best_entry.assign(rp, 0xFFFFFFFFu); o.nexthop_mask.assign(rp * r.mask_words, 0);
    if (P == 0) return o;
    uint32_t *bm = (uint32_t *)pool_->dev(rp * 4), *be = (uint32_t *)pool_->dev(rp * 4);
    uint64_t *nm = (uint64_t *)pool_->dev(rp * 8 * r.mask_words);
    static const uint32_t zero = 0;
    hspf_prefix_table tab{P, (uint32_t)pfx_vertex.size(), pfx_ptr.data(), pfx_vertex.empty() ? &zero : pfx_vertex.data(),
                          pfx_metric.empty() ? &zero : pfx_metric.data(), flags};
    hspf_routes ro{bm, be, nm};
    const int rc = hspf_routes_device(ctx_, r.n_vertices, r.n_roots, r.mask_words, r.dist, r.flags, r.mask, &tab, &ro);
    bool ok = rc == HSPF_OK && hipMemcpy(o.best_metric.data(), bm, rp * 4, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(o.best_entry.data(), be, rp * 4, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(o.nexthop_mask.data(), nm, rp * 8 * r.mask_words, hipMemcpyDeviceToHost) == hipSuccess;
    pool_->dev_free(bm, rp * 4); pool_->dev_free(be, rp * 4); pool_->dev_free(nm, rp * 8 * r.mask_words);
    if (!ok) throw std::runtime_error(std::string("hspf_routes_device: ") + hspf_last_error(ctx_));
    return o;
  }
  std::unique_ptr<DeviceRoutes> routes_device(DeviceRun &run, const std::vector<uint32_t> &pfx_ptr, const std::vector<uint32_t> &pfx_vertex,
                                              const std::vector<uint32_t> &pfx_metric, uint32_t flags) override {
    auto &r = static_cast<HipDeviceRun &>(run);
    auto o = std::make_unique<HipDeviceRoutes>(pool_);
    o->n_roots = r.n_roots; o->n_prefixes = (uint32_t)pfx_ptr.size() - 1; o->mask_words = r.mask_words;
    if (!o->alloc()) throw std::runtime_error("hipMalloc of the route tables failed");
    if (o->n_prefixes == 0) return o;
    static const uint32_t zero = 0;
    hspf_prefix_table tab{o->n_prefixes, (uint32_t)pfx_vertex.size(), pfx_ptr.data(), pfx_vertex.empty() ? &zero : pfx_vertex.data(),
                          pfx_metric.empty() ? &zero : pfx_metric.data(), flags};
    hspf_routes ro = o->raw();
    const int rc = hspf_routes_device(ctx_, r.n_vertices, r.n_roots, r.mask_words, r.dist, r.flags, r.mask, &tab, &ro);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_routes_device: ") + hspf_last_error(ctx_));
    return o;
  }
  std::unique_ptr<DeviceRoutes> routes_upload(const RoutesOut &t, uint32_t n_roots, uint32_t n_prefixes, uint32_t mask_words) override {
    auto o = std::make_unique<HipDeviceRoutes>(pool_);
    o->n_roots = n_roots; o->n_prefixes = n_prefixes; o->mask_words = mask_words;
    const size_t rp = (size_t)n_roots * n_prefixes;
    if (t.best_metric.size() != rp || t.best_entry.size() != rp || t.nexthop_mask.size() != rp * mask_words) throw std::runtime_error("routes_upload: table sizes");
    if (!o->alloc()) throw std::runtime_error("hipMalloc of the route tables failed");
    if (rp && (hipMemcpy(o->bm, t.best_metric.data(), rp * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(o->be, t.best_entry.data(), rp * 4, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(o->nm, t.nexthop_mask.data(), rp * 8 * mask_words, hipMemcpyHostToDevice) != hipSuccess))
      throw std::runtime_error("hipMemcpy of the route tables failed");
    return o;
  }
  std::unique_ptr<DeviceRoutes> rib_new(uint32_t n_prefixes, uint32_t mask_words) override {
    auto o = std::make_unique<HipDeviceRoutes>(pool_);
    o->n_roots = 1; o->n_prefixes = n_prefixes; o->mask_words = mask_words;
    o->alloc(); o->alloc_origin();
    const hspf_rib_device rib{n_prefixes, mask_words, o->bm, o->be, o->nm, o->org};
    const int rc = hspf_rib_clear_device(ctx_, &rib);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_rib_clear_device: ") + hspf_last_error(ctx_));
    return o;
  }
  void rib_fold(DeviceRoutes &rib_set, DeviceRun &run, const std::vector<uint32_t> &pfx_ptr, const std::vector<uint32_t> &pfx_vertex,
                const std::vector<uint32_t> &pfx_metric, const std::vector<uint32_t> &pfx_origin, const std::vector<uint32_t> &prefix_map,
                uint32_t area_index, uint32_t word_offset) override {
    auto &o = static_cast<HipDeviceRoutes &>(rib_set);
    auto &r = static_cast<HipDeviceRun &>(run);
    if (pfx_ptr.size() <= 1) return;
    static const uint32_t zero = 0;
    hspf_prefix_table tab{(uint32_t)pfx_ptr.size() - 1, (uint32_t)pfx_vertex.size(), pfx_ptr.data(), pfx_vertex.empty() ? &zero : pfx_vertex.data(),
                          pfx_metric.empty() ? &zero : pfx_metric.data(), HSPF_PFX_SATURATING | HSPF_PFX_ORDERED, pfx_origin.empty() ? &zero : pfx_origin.data(),
                          nullptr, nullptr, nullptr};
    const hspf_rib_device rib{o.n_prefixes, o.mask_words, o.bm, o.be, o.nm, o.org};
    const int rc = hspf_rib_fold_device(ctx_, r.n_vertices, r.mask_words, r.dist, r.flags, r.mask, &tab, prefix_map.data(), area_index, word_offset, &rib);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_rib_fold_device: ") + hspf_last_error(ctx_));
  }
  RouteRecords routes_changed(DeviceRoutes &old_set, DeviceRoutes &new_set) override {
    auto &a = static_cast<HipDeviceRoutes &>(old_set);
    auto &b = static_cast<HipDeviceRoutes &>(new_set);
    if (a.n_roots != b.n_roots || a.n_prefixes != b.n_prefixes || a.mask_words != b.mask_words) throw std::runtime_error("routes_changed: the two sets differ in shape");
    RouteRecords out;
    out.mask_words = b.mask_words;
    const size_t rp = (size_t)b.n_roots * b.n_prefixes;
    if (rp == 0) return out;
    // scratch of the comparison (action bytes, changed list, per-root bounds): kept by the engine, grown on demand
    const size_t need = rp + rp * 4 + ((size_t)b.n_roots + 1) * 4 + 64;
    if (diff_cap_ < need) {
      if (diff_) (void)hipFree(diff_);
      diff_ = nullptr; diff_cap_ = 0;
      if (hipMalloc(&diff_, need) != hipSuccess) throw std::runtime_error("hipMalloc of the diff scratch failed");
      diff_cap_ = need;
    }
    uint32_t *changed = (uint32_t *)diff_, *cptr = changed + rp;
    uint8_t *action = (uint8_t *)(cptr + b.n_roots + 1);
    hspf_routes ro = a.raw(), rn = b.raw();
    int rc = hspf_routes_diff_device(ctx_, b.n_roots, b.n_prefixes, b.mask_words, &ro, &rn, action, changed, cptr);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_routes_diff_device: ") + hspf_last_error(ctx_));
    const uint32_t k = hspf_routes_diff_count(ctx_);
    out.words.assign((size_t)k * out.stride(), 0u);
    if (k) {
      rc = hspf_routes_pack(ctx_, b.n_roots, b.n_prefixes, b.mask_words, &rn, action, changed, cptr, k, out.words.data());
      if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_routes_pack: ") + hspf_last_error(ctx_));
      out.old_words.assign(out.words.size(), 0u);                    // the same list packed from the old set: what the route was
      rc = hspf_routes_pack(ctx_, b.n_roots, b.n_prefixes, b.mask_words, &ro, action, changed, cptr, k, out.old_words.data());
      if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_routes_pack (old set): ") + hspf_last_error(ctx_));
    }
    return out;
  }
 private:
  void *diff_ = nullptr;
  size_t diff_cap_ = 0;
  template <typename WT>
  static void decode_row(const WT *w, const hspf_packed_layout &ly, uint16_t exact_flag, Tables &t, size_t o) {
    const WT nr = (WT)ly.not_reached, hm = (WT)ly.hops_mask, mm = (WT)((1ull << ly.mask_bits) - 1ull);
    const uint32_t ds = ly.dist_shift, hs = ly.hops_shift, n = t.n_vertices;
    uint32_t *d = t.dist.data() + o; uint16_t *h = t.hops.data() + o, *f = t.flags.data() + o; uint64_t *m = t.mask.data() + o;
    for (uint32_t v = 0; v < n; ++v) {
      const WT x = w[v];
      const bool in = x < nr;
      d[v] = in ? (uint32_t)(x >> ds) : 0xFFFFFFFFu;
      h[v] = in ? (uint16_t)((x >> hs) & hm) : (uint16_t)0;
      f[v] = in ? (uint16_t)(HSPF_RF_IN_SPT | exact_flag) : (uint16_t)0;
      m[v] = in ? (uint64_t)(x & mm) : 0ull;
    }
  }
  hspf_ctx *ctx_ = nullptr;
  void *pin_ = nullptr;
  size_t pin_cap_ = 0;
  std::shared_ptr<HipPool> pool_ = std::make_shared<HipPool>();      // (shared with the run / route objects: they may outlive the engine)
};

}  // namespace host
}  // namespace hspf
