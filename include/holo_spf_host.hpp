// holo_spf_host.hpp — what the C++ host-side twins (holo_spf_isis.hpp, holo_spf_ospf.hpp) share: the three engine calls
// they need, the product implementation of those calls on the C ABI, and IP prefix / address keys in the order of the
// reference's BTreeMap<IpNetwork, _> / BTreeMap<IpAddr, _>.
#pragma once
#include <algorithm>
#include <array>
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
class DeviceRun {
 public:
  virtual ~DeviceRun() = default;
  virtual Tables host_tables() = 0;   // the copy the host side still needs for next-hop resolution
  uint32_t n_roots = 0, n_vertices = 0, mask_words = 1;
};
struct RoutesOut {                    // per (root, prefix), row-major: see hspf_routes in holo_spf_hip.h
  std::vector<uint32_t> best_metric, best_entry;
  std::vector<uint64_t> nexthop_mask;
};
class Engine {
 public:
  virtual ~Engine() = default;
  virtual std::unique_ptr<Graph> upload(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col,
                                        const std::vector<uint32_t> &metric, const std::vector<uint8_t> &vflags,
                                        uint32_t max_path_metric) = 0;
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
};

// CSR with the rows of `vertices` (strictly ascending) replaced — the host-side twin of hspf_graph_patch.
inline void splice_rows(std::vector<uint32_t> &row_ptr, std::vector<uint32_t> &col, std::vector<uint32_t> &metric,
                        std::vector<uint8_t> &vflags, const std::vector<uint32_t> &vertices,
                        const std::vector<std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> &rows,
                        const std::vector<uint8_t> &new_flags) {
  const uint32_t n = (uint32_t)row_ptr.size() - 1;
  std::vector<uint32_t> nrp(n + 1, 0), ncol, nmet;
  size_t j = 0;
  for (uint32_t u = 0; u < n; ++u) {
    nrp[u] = (uint32_t)ncol.size();
    if (j < vertices.size() && vertices[j] == u) {
      ncol.insert(ncol.end(), rows[j].first.begin(), rows[j].first.end());
      nmet.insert(nmet.end(), rows[j].second.begin(), rows[j].second.end());
      vflags[u] = new_flags[j];
      ++j;
    } else {
      ncol.insert(ncol.end(), col.begin() + row_ptr[u], col.begin() + row_ptr[u + 1]);
      nmet.insert(nmet.end(), metric.begin() + row_ptr[u], metric.begin() + row_ptr[u + 1]);
    }
  }
  nrp[n] = (uint32_t)ncol.size();
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
inline IpKey parse_ip(const std::string &text) {             // "a.b.c.d[/len]" or IPv6 text form (with "::")
  IpKey k;
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


// ---- the product engine: libholo_spf_hip.so through the C ABI ---------------------------------------------------------
class HipGraph : public Graph {
 public:
  HipGraph(hspf_ctx *c, hspf_graph *g) : ctx(c), g(g) {}
  ~HipGraph() override { if (g) hspf_graph_free(ctx, g); }
  hspf_ctx *ctx;
  hspf_graph *g;
};
class HipDeviceRun : public DeviceRun {             // dist / hops / flags / masks of one run in plain hipMalloc buffers
 public:
  ~HipDeviceRun() override { for (void *p : {(void *)dist, (void *)hops, (void *)flags, (void *)mask}) if (p) (void)hipFree(p); }
  Tables host_tables() override {
    Tables t;
    t.n_roots = n_roots; t.n_vertices = n_vertices; t.mask_words = mask_words;
    const size_t rn = (size_t)n_roots * n_vertices;
    t.dist.resize(rn); t.hops.resize(rn); t.flags.resize(rn); t.mask.resize(rn * mask_words);
    if (hipMemcpy(t.dist.data(), dist, rn * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(t.hops.data(), hops, rn * 2, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(t.flags.data(), flags, rn * 2, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(t.mask.data(), mask, rn * 8 * mask_words, hipMemcpyDeviceToHost) != hipSuccess)
      throw std::runtime_error("hipMemcpy of the run tables failed");
    return t;
  }
  uint32_t *dist = nullptr; uint16_t *hops = nullptr, *flags = nullptr; uint64_t *mask = nullptr;
};
class HipEngine : public Engine {
 public:
  explicit HipEngine(int device = 0) {
    const int rc = hspf_init(device, &ctx_);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_init: ") + hspf_strerror(rc));   // no CPU fallback
  }
  ~HipEngine() override { if (ctx_) hspf_shutdown(ctx_); }
  std::unique_ptr<Graph> upload(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col,
                                const std::vector<uint32_t> &metric, const std::vector<uint8_t> &vflags,
                                uint32_t max_path_metric) override {
    hspf_csr csr{(uint32_t)vflags.size(), (uint32_t)col.size(), row_ptr.data(), col.data(), metric.data(), vflags.data(), max_path_metric};
    hspf_graph *g = nullptr;
    const int rc = hspf_graph_upload(ctx_, &csr, &g);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_graph_upload: ") + hspf_last_error(ctx_));
    return std::make_unique<HipGraph>(ctx_, g);
  }
  Tables run(Graph &gr, const std::vector<uint32_t> &roots, uint32_t run_flags) override {
    hspf_graph *g = static_cast<HipGraph &>(gr).g;
    Tables t;
    t.n_roots = (uint32_t)roots.size();
    t.n_vertices = hspf_graph_n_vertices(g);
    int rc = hspf_mask_words(ctx_, g, roots.data(), t.n_roots, &t.mask_words);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_mask_words: ") + hspf_last_error(ctx_));
    const size_t rn = (size_t)t.n_roots * t.n_vertices;
    t.dist.resize(rn); t.hops.resize(rn); t.flags.resize(rn); t.mask.resize(rn * t.mask_words);
    if (run_flags & HSPF_RUN_POP_RANK) t.pop_rank.resize(rn);
    hspf_result out{t.dist.data(), t.hops.data(), t.flags.data(), t.mask.data(), t.mask_words,
                    (run_flags & HSPF_RUN_POP_RANK) ? t.pop_rank.data() : nullptr};
    rc = hspf_run(ctx_, g, roots.data(), t.n_roots, run_flags, &out);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_run: ") + hspf_last_error(ctx_));
    return t;
  }
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
    auto r = std::make_unique<HipDeviceRun>();
    r->n_roots = (uint32_t)roots.size(); r->n_vertices = hspf_graph_n_vertices(g);
    int rc = hspf_mask_words(ctx_, g, roots.data(), r->n_roots, &r->mask_words);
    if (rc != HSPF_OK) throw std::runtime_error(std::string("hspf_mask_words: ") + hspf_last_error(ctx_));
    const size_t rn = (size_t)r->n_roots * r->n_vertices;
    if (hipMalloc((void **)&r->dist, rn * 4) != hipSuccess || hipMalloc((void **)&r->hops, rn * 2) != hipSuccess ||
        hipMalloc((void **)&r->flags, rn * 2) != hipSuccess || hipMalloc((void **)&r->mask, rn * 8 * r->mask_words) != hipSuccess)
      throw std::runtime_error("hipMalloc of the run tables failed");
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
    uint32_t *bm = nullptr, *be = nullptr; uint64_t *nm = nullptr;
    if (hipMalloc((void **)&bm, rp * 4) != hipSuccess || hipMalloc((void **)&be, rp * 4) != hipSuccess || hipMalloc((void **)&nm, rp * 8 * r.mask_words) != hipSuccess)
      throw std::runtime_error("hipMalloc of the route tables failed");
    static const uint32_t zero = 0;
    hspf_prefix_table tab{P, (uint32_t)pfx_vertex.size(), pfx_ptr.data(), pfx_vertex.empty() ? &zero : pfx_vertex.data(),
                          pfx_metric.empty() ? &zero : pfx_metric.data(), flags};
    hspf_routes ro{bm, be, nm};
    const int rc = hspf_routes_device(ctx_, r.n_vertices, r.n_roots, r.mask_words, r.dist, r.flags, r.mask, &tab, &ro);
    bool ok = rc == HSPF_OK && hipMemcpy(o.best_metric.data(), bm, rp * 4, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(o.best_entry.data(), be, rp * 4, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(o.nexthop_mask.data(), nm, rp * 8 * r.mask_words, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(bm); (void)hipFree(be); (void)hipFree(nm);
    if (!ok) throw std::runtime_error(std::string("hspf_routes_device: ") + hspf_last_error(ctx_));
    return o;
  }
 private:
  hspf_ctx *ctx_ = nullptr;
};

}  // namespace host
}  // namespace hspf
