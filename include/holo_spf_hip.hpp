// holo_spf_hip.hpp — C++17 RAII convenience layer over the C ABI of include/holo_spf_hip.h.
//
// Header only, no dependency beyond the C header and the HIP runtime the caller already links for its
// device buffers.  It is the compiled-language twin of the safe Rust wrapper sketched in
// INTEGRATION.md §3 (Engine: Send, not Sync; Graph freed on drop; errors as codes, never exceptions
// across the boundary — this layer turns a non-zero code into hspf::Error for C++ callers).
#ifndef HOLO_SPF_HIP_HPP
#define HOLO_SPF_HIP_HPP

#include <cstdint>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <algorithm>
#include <vector>

#include "holo_spf_hip.h"

namespace hspf {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &what) : std::runtime_error(what + ": " + hspf_strerror(c)), code(c) {}
};

struct Tables {                       // row-major [root][vertex] host results of one run
  uint32_t n_roots = 0, n_vertices = 0, mask_words = 1;
  std::vector<uint32_t> dist, pop_rank;
  std::vector<uint16_t> hops, flags;
  std::vector<uint64_t> mask;
  hspf_stats stats{};
};

// Page-locked host memory from hspf_host_alloc (results cross the bus at full speed into it), freed with the object.
class PinnedBuffer {
 public:
  PinnedBuffer() = default;
  PinnedBuffer(hspf_ctx *ctx, size_t bytes) : ctx_(ctx), bytes_(bytes) {
    const int rc = hspf_host_alloc(ctx, bytes, &p_);
    if (rc != HSPF_OK) throw Error(rc, "hspf_host_alloc");
  }
  PinnedBuffer(PinnedBuffer &&o) noexcept : ctx_(o.ctx_), p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
  PinnedBuffer &operator=(PinnedBuffer &&o) noexcept { if (this != &o) { reset(); ctx_ = o.ctx_; p_ = o.p_; bytes_ = o.bytes_; o.p_ = nullptr; o.bytes_ = 0; } return *this; }
  PinnedBuffer(const PinnedBuffer &) = delete;
  PinnedBuffer &operator=(const PinnedBuffer &) = delete;
  ~PinnedBuffer() { reset(); }
  void reset() { if (p_) hspf_host_free(ctx_, p_); p_ = nullptr; bytes_ = 0; }
  void *data() const { return p_; }
  size_t size() const { return bytes_; }
 private:
  hspf_ctx *ctx_ = nullptr;
  void *p_ = nullptr;
  size_t bytes_ = 0;
};

// Packed results of one run (ABI 7): ONE word per (root, vertex) in a page-locked buffer + the run's field positions.
// A vertex is decoded where it is looked at (the rebuild of `Vertex{distance, hops, nexthops}` touches each once).
struct PackedTables {
  uint32_t n_roots = 0, n_vertices = 0;
  hspf_packed_layout layout{};
  PinnedBuffer words;
  std::vector<uint8_t> root_status;            // HSPF_ROOT_EXACT per root
  hspf_stats stats{};
  uint64_t word(uint32_t r, uint32_t v) const { return hspf_packed_word(&layout, words.data(), (size_t)r * n_vertices + v); }
  bool in_spt(uint32_t r, uint32_t v) const { return hspf_packed_in_spt(&layout, word(r, v)) != 0; }
  uint32_t dist(uint32_t r, uint32_t v) const { return hspf_packed_dist(&layout, word(r, v)); }
  uint16_t hops(uint32_t r, uint32_t v) const { return hspf_packed_hops(&layout, word(r, v)); }
  uint64_t mask(uint32_t r, uint32_t v) const { return hspf_packed_mask(&layout, word(r, v)); }
};

class Engine;

// The engine context, shared by the Engine and every Graph made from it: a Graph that outlives its Engine (members
// declared in the wrong order, a Graph moved out of the Engine's scope) keeps the context alive until it has freed its
// device arrays, instead of handing hspf_graph_free a dangling ctx.
struct CtxHolder {
  hspf_ctx *ctx = nullptr;
  CtxHolder() = default;
  CtxHolder(const CtxHolder &) = delete;
  CtxHolder &operator=(const CtxHolder &) = delete;
  ~CtxHolder() { if (ctx) hspf_shutdown(ctx); }
};

class Graph {
 public:
  Graph(Graph &&o) noexcept : holder_(std::move(o.holder_)), ctx_(o.ctx_), g_(o.g_) { o.g_ = nullptr; }
  Graph(const Graph &) = delete;
  Graph &operator=(const Graph &) = delete;
  ~Graph() { if (g_) hspf_graph_free(ctx_, g_); }
  uint32_t n_vertices() const { return hspf_graph_n_vertices(g_); }
  uint32_t n_links() const { return hspf_graph_n_edges(g_); }
  uint32_t n_links_kept() const { return hspf_graph_n_edges_kept(g_); }
  hspf_graph *raw() const { return g_; }

  // One re-originated / purged LSP or LSA = one replaced row (hspf_graph_patch).
  struct Row {
    uint32_t vertex;
    std::vector<uint32_t> col, metric;
    uint8_t vflags;
  };
  // Rows in any order; duplicates of a vertex are an error (HSPF_E_INVAL from the library).
  void patch(std::vector<Row> rows) {
    std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.vertex < b.vertex; });
    std::vector<uint32_t> vertex, row_ptr{0}, col, metric;
    std::vector<uint8_t> vflags;
    for (const Row &r : rows) {
      if (r.col.size() != r.metric.size()) throw Error(HSPF_E_INVAL, "Graph::patch: col/metric length mismatch");
      vertex.push_back(r.vertex);
      col.insert(col.end(), r.col.begin(), r.col.end());
      metric.insert(metric.end(), r.metric.begin(), r.metric.end());
      row_ptr.push_back((uint32_t)col.size());
      vflags.push_back(r.vflags);
    }
    if (col.empty()) { col.push_back(0); metric.push_back(0); }      // non-NULL pointers for an all-empty delta
    hspf_rows d{(uint32_t)vertex.size(), vertex.data(), row_ptr.data(), col.data(), metric.data(), vflags.data()};
    const int rc = hspf_graph_patch(ctx_, g_, &d);
    if (rc != HSPF_OK) throw Error(rc, std::string("hspf_graph_patch (") + hspf_last_error(ctx_) + ")");
  }

 private:
  friend class Engine;
  Graph(std::shared_ptr<CtxHolder> h, hspf_graph *g) : holder_(std::move(h)), ctx_(holder_->ctx), g_(g) {}
  std::shared_ptr<CtxHolder> holder_;
  hspf_ctx *ctx_;
  hspf_graph *g_;
};

class Engine {
 public:
  explicit Engine(int device = 0) : holder_(std::make_shared<CtxHolder>()) {
    const int rc = hspf_init(device, &holder_->ctx);
    if (rc != HSPF_OK) throw Error(rc, "hspf_init");
    ctx_ = holder_->ctx;
  }
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;
  ~Engine() = default;                               // the context goes with the last holder (this, or a surviving Graph)
  hspf_ctx *raw() const { return ctx_; }

  Graph upload(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col,
               const std::vector<uint32_t> &metric, const std::vector<uint8_t> &vflags, uint32_t max_path_metric) {
    hspf_csr csr{(uint32_t)vflags.size(), (uint32_t)col.size(), row_ptr.data(), col.data(), metric.data(), vflags.data(),
                 max_path_metric};
    hspf_graph *g = nullptr;
    const int rc = hspf_graph_upload(ctx_, &csr, &g);
    if (rc != HSPF_OK) throw Error(rc, std::string("hspf_graph_upload (") + hspf_last_error(ctx_) + ")");
    return Graph(holder_, g);
  }

  uint32_t mask_words(const Graph &g, const std::vector<uint32_t> &roots) {
    uint32_t w = 1;
    const int rc = hspf_mask_words(ctx_, g.raw(), roots.data(), (uint32_t)roots.size(), &w);
    if (rc != HSPF_OK) throw Error(rc, "hspf_mask_words");
    return w;
  }

  // (H vertices, slot bases, total slots) of one root — see the header comment of hspf_result.
  std::pair<std::vector<uint32_t>, std::vector<uint32_t>> slot_table(const Graph &g, uint32_t root, uint32_t *total = nullptr) {
    uint32_t tot = 0;
    const int cnt = hspf_slot_table(ctx_, g.raw(), root, nullptr, nullptr, 0, &tot);
    if (cnt < 0) throw Error(cnt, "hspf_slot_table");
    std::vector<uint32_t> hv(cnt), hb(cnt);
    hspf_slot_table(ctx_, g.raw(), root, hv.data(), hb.data(), (uint32_t)cnt, &tot);
    if (total) *total = tot;
    return {hv, hb};
  }

  Tables run(const Graph &g, const std::vector<uint32_t> &roots, uint32_t run_flags = 0) {
    Tables t;
    t.n_roots = (uint32_t)roots.size();
    t.n_vertices = g.n_vertices();
    t.mask_words = mask_words(g, roots);
    const size_t rn = (size_t)t.n_roots * t.n_vertices;
    t.dist.resize(rn); t.hops.resize(rn); t.flags.resize(rn); t.mask.resize(rn * t.mask_words);
    if (run_flags & HSPF_RUN_POP_RANK) t.pop_rank.resize(rn);
    hspf_result out{t.dist.data(), t.hops.data(), t.flags.data(), t.mask.data(), t.mask_words,
                    t.pop_rank.empty() ? nullptr : t.pop_rank.data()};
    const int rc = hspf_run(ctx_, g.raw(), roots.data(), t.n_roots, run_flags, &out);
    if (rc != HSPF_OK) throw Error(rc, std::string("hspf_run (") + hspf_last_error(ctx_) + ")");
    hspf_get_stats(ctx_, &t.stats);
    return t;
  }

  // ABI 7: packed results — a quarter of hspf_run's bytes over the bus.  `reuse`: a PackedTables of an earlier run whose
  // buffer is taken over when it is large enough.  Throws Error with code HSPF_E_NO_PACKED when the run's results do not
  // fit packed words (more than 24 first-hop slots): the caller then uses run().
  PackedTables run_packed(const Graph &g, const std::vector<uint32_t> &roots, uint32_t run_flags = 0, PackedTables *reuse = nullptr) {
    PackedTables t;
    t.n_roots = (uint32_t)roots.size();
    t.n_vertices = g.n_vertices();
    const size_t need = (size_t)8 * t.n_roots * t.n_vertices;
    if (reuse && reuse->words.size() >= need) t.words = std::move(reuse->words);
    else t.words = PinnedBuffer(ctx_, need);
    t.root_status.assign(t.n_roots, 0);
    const int rc = hspf_run_packed(ctx_, g.raw(), roots.data(), t.n_roots, run_flags, t.words.data(), t.words.size(), &t.layout, t.root_status.data());
    if (rc != HSPF_OK) throw Error(rc, std::string("hspf_run_packed (") + hspf_last_error(ctx_) + ")");
    hspf_get_stats(ctx_, &t.stats);
    return t;
  }

  // ABI 6: several runs in flight from one caller thread (one per area / level / neighbour set).  `out_device` holds
  // DEVICE pointers sized for roots.size() rows; they must stay valid, and unshared with other runs in flight, until
  // wait(ticket) has returned.  Results are bit-identical to hspf_run_device's.
  uint64_t run_device_async(const Graph &g, const std::vector<uint32_t> &roots, uint32_t run_flags, const hspf_result &out_device) {
    uint64_t ticket = 0;
    const int rc = hspf_run_device_async(ctx_, g.raw(), roots.data(), (uint32_t)roots.size(), run_flags, &out_device, &ticket);
    if (rc != HSPF_OK) throw Error(rc, std::string("hspf_run_device_async (") + hspf_last_error(ctx_) + ")");
    return ticket;
  }
  hspf_stats wait(uint64_t ticket) {
    hspf_stats st{};
    const int rc = hspf_wait(ctx_, ticket, &st);
    if (rc != HSPF_OK) throw Error(rc, std::string("hspf_wait (") + hspf_last_error(ctx_) + ")");
    return st;
  }
  void wait_all() { (void)hspf_wait_all(ctx_); }
  uint32_t async_lanes() const { return hspf_async_lanes(ctx_); }
  // true: these runs are too small to pay for a launch — the caller keeps its own CPU loop (hspf_recommend_cpu)
  static bool recommend_cpu(uint32_t n_vertices, uint32_t n_edges, uint32_t n_roots) {
    return hspf_recommend_cpu(n_vertices, n_edges, n_roots) != 0;
  }

 private:
  std::shared_ptr<CtxHolder> holder_;
  hspf_ctx *ctx_ = nullptr;
};

}  // namespace hspf
#endif
