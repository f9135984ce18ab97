// spf_multi.hip.h — several GPUs behind the C ABI (include/holo_spf_hip.h "several GPUs"): one engine context per rank,
// whole 64-root batches dealt to the ranks, in-place all-gather of per-root tables.  Included by spf_capi.hip (one TU).
//
// Gather back ends:
//   peer copies  single process: every local device pushes its rows to every other one with hipMemcpyPeerAsync on its
//                own stream (xGMI is point-to-point: N x (N-1) direct copies use every link once, no ring);
//   RCCL         one process per GPU: librccl.so is loaded with dlopen at hspf_multi_init (the engine has no link-time
//                dependency on it), ncclAllGather when the slices are equal, a group of ncclBroadcast otherwise.
#pragma once
#include <dlfcn.h>
#include <thread>

namespace {

// the few RCCL entry points used, bound at run time (rccl.h: ncclResult_t = int, ncclSuccess = 0, ncclUniqueId = 128 bytes)
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, /* ncclUniqueId by value */ struct RcclId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
struct RcclId { char internal[HSPF_COMM_ID_BYTES]; };
constexpr int RCCL_INT8 = 0;      // ncclInt8 / ncclChar

bool rccl_load(RcclApi &a, std::string &err) {
  if (a.lib) return true;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (a.lib) break;
  }
  if (!a.lib) { err = std::string("librccl.so not found: ") + dlerror(); return false; }
  auto sym = [&](const char *n) { void *p = dlsym(a.lib, n); if (!p && err.empty()) err = std::string("librccl.so lacks ") + n; return p; };
  a.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
  a.CommInitRank = (int (*)(void **, int, RcclId, int))sym("ncclCommInitRank");
  a.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
  a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))sym("ncclAllGather");
  a.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclBroadcast");
  a.GroupStart = (int (*)())sym("ncclGroupStart");
  a.GroupEnd = (int (*)())sym("ncclGroupEnd");
  a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
  return err.empty();
}

}  // namespace

struct hspf_multi {
  std::vector<hspf_ctx *> ctx;
  std::vector<int> dev;
  uint32_t world = 0, first_rank = 0;
  bool use_rccl = false;
  RcclApi rccl;
  std::vector<void *> comm;
  std::vector<hipStream_t> cstream;                  // per local device: the stream the gathers run on
  struct Pending { const void *key; hipEvent_t ev; uint32_t local; };
  std::vector<Pending> pending;                      // asynchronous gathers in flight, keyed by the dist table they fill
  bool force_bcast = false;
  struct Ticket { uint64_t id; std::vector<uint64_t> t; std::vector<uint8_t> has; uint32_t n_roots; size_t n_vertices; };
  std::vector<Ticket> tickets;                       // hspf_multi_run_async calls not yet waited for
  uint64_t next_ticket = 1;
  std::vector<std::vector<hipEvent_t>> free_events;  // per LOCAL device: an event is only ever recorded on the device it was created on
  std::string last_error;
};

struct hspf_multi_graph {
  std::vector<hspf_graph *> g;
};

static thread_local std::string g_multi_init_error;   // why the last hspf_multi_init of this thread failed (it has no handle to carry the text)

extern "C" {

void hspf_shard_bounds(uint32_t n_roots, uint32_t world, uint32_t rank, uint32_t *begin, uint32_t *end) {
  if (world == 0) world = 1;
  const uint32_t nb = (n_roots + 63u) / 64u, base = nb / world, extra = nb % world;
  const uint64_t b0 = (uint64_t)base * rank + std::min(rank, extra);
  const uint64_t b1 = b0 + base + (rank < extra ? 1u : 0u);
  if (begin) *begin = (uint32_t)std::min<uint64_t>(b0 * 64u, n_roots);
  if (end) *end = (uint32_t)std::min<uint64_t>(b1 * 64u, n_roots);
}

uint32_t hspf_plan_areas(uint32_t n_areas, const uint32_t *roots_per_area, uint32_t world, hspf_area_slice *out, uint32_t cap) {
  if (!roots_per_area || world == 0) return 0;
  uint64_t total = 0;
  for (uint32_t a = 0; a < n_areas; ++a) total += (roots_per_area[a] + 63u) / 64u;
  uint32_t n_out = 0;
  uint64_t unit = 0;                                  // index of the next (area, batch) unit in area order
  uint32_t area = 0, batch_in_area = 0;
  for (uint32_t r = 0; r < world; ++r) {
    const uint64_t hi = total / world * (r + 1) + std::min<uint64_t>(r + 1, total % world);   // units [.., hi) belong to ranks <= r
    while (unit < hi && area < n_areas) {
      const uint32_t nb = (roots_per_area[area] + 63u) / 64u;
      if (batch_in_area >= nb) { ++area; batch_in_area = 0; continue; }
      const uint32_t take = (uint32_t)std::min<uint64_t>(nb - batch_in_area, hi - unit);
      if (n_out < cap && out)
        out[n_out] = hspf_area_slice{r, area, batch_in_area * 64u, std::min((batch_in_area + take) * 64u, roots_per_area[area])};
      ++n_out;
      batch_in_area += take; unit += take;
    }
  }
  return n_out;
}

int hspf_multi_unique_id(uint8_t id[HSPF_COMM_ID_BYTES]) {
  if (!id) return HSPF_E_INVAL;
  static RcclApi api;
  std::string err;
  if (!rccl_load(api, err)) return HSPF_E_NODEV;
  RcclId u{};
  if (api.GetUniqueId(&u) != 0) return HSPF_E_HIP;
  memcpy(id, u.internal, HSPF_COMM_ID_BYTES);
  return HSPF_OK;
}

const char *hspf_multi_last_error(const hspf_multi *m) { return m ? m->last_error.c_str() : ""; }
hspf_ctx *hspf_multi_ctx(hspf_multi *m, uint32_t i) { return (m && i < m->ctx.size()) ? m->ctx[i] : nullptr; }
uint32_t hspf_multi_n_local(const hspf_multi *m) { return m ? (uint32_t)m->ctx.size() : 0; }
hspf_graph *hspf_multi_graph_local(hspf_multi_graph *g, uint32_t i) { return (g && i < g->g.size()) ? g->g[i] : nullptr; }

void hspf_multi_shutdown(hspf_multi *m) {
  if (!m) return;
  for (size_t i = 0; i < m->cstream.size(); ++i)
    if (m->cstream[i]) { (void)hipSetDevice(m->dev[i]); (void)hipStreamSynchronize(m->cstream[i]); }
  for (auto &p : m->pending) (void)hipEventDestroy(p.ev);
  for (auto &v : m->free_events) for (auto &e : v) (void)hipEventDestroy(e);
  for (size_t i = 0; i < m->comm.size(); ++i)
    if (m->comm[i]) { (void)hipSetDevice(m->dev[i]); (void)m->rccl.CommDestroy(m->comm[i]); }
  for (size_t i = 0; i < m->cstream.size(); ++i)
    if (m->cstream[i]) { (void)hipSetDevice(m->dev[i]); (void)hipStreamDestroy(m->cstream[i]); }
  for (hspf_ctx *c : m->ctx) hspf_shutdown(c);
  delete m;
}

const char *hspf_multi_init_error(void) { return g_multi_init_error.c_str(); }

int hspf_multi_init(const hspf_multi_config *cfg, hspf_multi **out) {
  if (!cfg || !out || cfg->n_local == 0 || !cfg->device_ordinals || cfg->world < cfg->n_local ||
      (uint64_t)cfg->first_rank + cfg->n_local > cfg->world || (!cfg->unique_id && cfg->world != cfg->n_local))
    return HSPF_E_INVAL;
  *out = nullptr;
  hspf_multi *m = nullptr;
  try { m = new hspf_multi(); } catch (...) { return HSPF_E_NOMEM; }
  m->world = cfg->world; m->first_rank = cfg->first_rank;
  m->force_bcast = getenv("HSPF_GATHER_FORCE_BCAST") != nullptr;
  try {
    for (uint32_t i = 0; i < cfg->n_local; ++i) {
      hspf_ctx *c = nullptr;
      const int rc = hspf_init(cfg->device_ordinals[i], &c);
      if (rc) { hspf_multi_shutdown(m); return rc; }
      m->ctx.push_back(c); m->dev.push_back(cfg->device_ordinals[i]);
      hipStream_t cs = nullptr;
      (void)hipSetDevice(cfg->device_ordinals[i]);
      if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { hspf_multi_shutdown(m); return HSPF_E_HIP; }
      m->cstream.push_back(cs);
      m->free_events.emplace_back();
    }
    // direct device-to-device copies between the local devices (errors ignored: already enabled / same device)
    for (uint32_t i = 0; i < cfg->n_local; ++i)
      for (uint32_t j = 0; j < cfg->n_local; ++j)
        if (m->dev[i] != m->dev[j]) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, m->dev[i], m->dev[j]) == hipSuccess && can) {
            (void)hipSetDevice(m->dev[i]);
            (void)hipDeviceEnablePeerAccess(m->dev[j], 0);
            (void)hipGetLastError();
          }
        }
    if (cfg->unique_id) {
      std::string err;
      if (!rccl_load(m->rccl, err)) { try { g_multi_init_error = err; } catch (...) {} hspf_multi_shutdown(m); return HSPF_E_NODEV; }
      RcclId id{};
      memcpy(id.internal, cfg->unique_id, HSPF_COMM_ID_BYTES);
      m->comm.assign(cfg->n_local, nullptr);
      int rc = m->rccl.GroupStart();
      for (uint32_t i = 0; i < cfg->n_local && rc == 0; ++i) {
        (void)hipSetDevice(m->dev[i]);
        rc = m->rccl.CommInitRank(&m->comm[i], (int)cfg->world, id, (int)(cfg->first_rank + i));
      }
      const int rc2 = m->rccl.GroupEnd();
      if (rc == 0) rc = rc2;
      if (rc != 0) {
        try { g_multi_init_error = std::string("rccl communicator: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "error") + " (code " + std::to_string(rc) + ")"; } catch (...) {}
        hspf_multi_shutdown(m); return HSPF_E_HIP;
      }
      m->use_rccl = true;
    }
  } catch (...) { hspf_multi_shutdown(m); return HSPF_E_NOMEM; }
  *out = m;
  return HSPF_OK;
}

int hspf_multi_graph_upload(hspf_multi *m, const hspf_csr *csr, hspf_multi_graph **out) {
  if (!m || !csr || !out) return HSPF_E_INVAL;
  hspf_multi_graph *mg = nullptr;
  try { mg = new hspf_multi_graph(); mg->g.assign(m->ctx.size(), nullptr); } catch (...) { delete mg; return HSPF_E_NOMEM; }
  for (size_t i = 0; i < m->ctx.size(); ++i) {
    const int rc = hspf_graph_upload(m->ctx[i], csr, &mg->g[i]);
    if (rc) { m->last_error = m->ctx[i]->last_error; hspf_multi_graph_free(m, mg); return rc; }
  }
  *out = mg;
  return HSPF_OK;
}

int hspf_multi_graph_patch(hspf_multi *m, hspf_multi_graph *g, const hspf_rows *rows) {
  if (!m || !g || g->g.size() != m->ctx.size()) return HSPF_E_INVAL;
  for (size_t i = 0; i < m->ctx.size(); ++i) {
    const int rc = hspf_graph_patch(m->ctx[i], g->g[i], rows);
    if (rc) { m->last_error = m->ctx[i]->last_error; return rc; }     // replica 0 rejects an invalid patch before any is changed
  }
  return HSPF_OK;
}

void hspf_multi_graph_free(hspf_multi *m, hspf_multi_graph *g) {
  if (!g) return;
  for (size_t i = 0; i < g->g.size(); ++i)
    if (g->g[i] && m && i < m->ctx.size()) hspf_graph_free(m->ctx[i], g->g[i]);
  delete g;
}

int hspf_multi_mask_words(hspf_multi *m, const hspf_multi_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t *out_words) {
  if (!m || !g || g->g.empty()) return HSPF_E_INVAL;
  return hspf_mask_words(m->ctx[0], g->g[0], roots, n_roots, out_words);
}

static int allgather_rows_impl(hspf_multi *m, void *const *tables, size_t row_bytes, uint32_t n_roots, bool sync) {
  if (!m || !tables || row_bytes == 0) return HSPF_E_INVAL;
  const uint32_t nl = (uint32_t)m->ctx.size();
  for (uint32_t i = 0; i < nl; ++i) if (!tables[i]) return HSPF_E_INVAL;
  int rc = HSPF_OK;
  if (m->use_rccl) {
    // equal slices: one ncclAllGather, in place (send = own slice of recv); ragged: one broadcast per rank, grouped
    bool equal = !m->force_bcast;                // HSPF_GATHER_FORCE_BCAST (tests): the ragged branch even when the slices are equal
    uint32_t b0, e0;
    hspf_shard_bounds(n_roots, m->world, 0, &b0, &e0);
    for (uint32_t r = 1; r < m->world; ++r) { uint32_t b, e; hspf_shard_bounds(n_roots, m->world, r, &b, &e); equal = equal && (e - b == e0 - b0); }
    int nrc = m->rccl.GroupStart();
    for (uint32_t i = 0; i < nl && nrc == 0; ++i) {
      (void)hipSetDevice(m->dev[i]);
      char *base = (char *)tables[i];
      if (equal) {
        uint32_t b, e; hspf_shard_bounds(n_roots, m->world, m->first_rank + i, &b, &e);
        nrc = m->rccl.AllGather(base + (size_t)b * row_bytes, base, (size_t)(e - b) * row_bytes, RCCL_INT8, m->comm[i], m->cstream[i]);
      } else {
        for (uint32_t r = 0; r < m->world && nrc == 0; ++r) {
          uint32_t b, e; hspf_shard_bounds(n_roots, m->world, r, &b, &e);
          if (e > b) nrc = m->rccl.Broadcast(base + (size_t)b * row_bytes, base + (size_t)b * row_bytes, (size_t)(e - b) * row_bytes, RCCL_INT8, (int)r, m->comm[i], m->cstream[i]);
        }
      }
    }
    const int nrc2 = m->rccl.GroupEnd();
    if (nrc == 0) nrc = nrc2;
    if (nrc != 0) { m->last_error = std::string("rccl: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(nrc) : "error"); rc = HSPF_E_HIP; }
  } else {
    for (uint32_t i = 0; i < nl && rc == HSPF_OK; ++i) {
      uint32_t b, e; hspf_shard_bounds(n_roots, m->world, m->first_rank + i, &b, &e);
      if (e == b) continue;
      const size_t off = (size_t)b * row_bytes, bytes = (size_t)(e - b) * row_bytes;
      (void)hipSetDevice(m->dev[i]);
      for (uint32_t j = 0; j < nl; ++j) {
        if (j == i || tables[j] == tables[i]) continue;
        const hipError_t er = m->dev[i] == m->dev[j]
            ? hipMemcpyAsync((char *)tables[j] + off, (char *)tables[i] + off, bytes, hipMemcpyDeviceToDevice, m->cstream[i])
            : hipMemcpyPeerAsync((char *)tables[j] + off, m->dev[j], (char *)tables[i] + off, m->dev[i], bytes, m->cstream[i]);
        if (er != hipSuccess) { m->last_error = std::string("peer copy: ") + hipGetErrorString(er); rc = HSPF_E_HIP; break; }
      }
    }
  }
  for (uint32_t i = 0; sync && i < nl; ++i) {
    (void)hipSetDevice(m->dev[i]);
    const hipError_t er = hipStreamSynchronize(m->cstream[i]);
    if (er != hipSuccess && rc == HSPF_OK) { m->last_error = std::string("gather sync: ") + hipGetErrorString(er); rc = HSPF_E_HIP; }
  }
  return rc;
}

int hspf_multi_allgather_rows(hspf_multi *m, void *const *tables, size_t row_bytes, uint32_t n_roots) {
  return allgather_rows_impl(m, tables, row_bytes, n_roots, true);
}

int hspf_multi_wait(hspf_multi *m) {
  if (!m) return HSPF_E_INVAL;
  int rc = HSPF_OK;
  for (size_t i = 0; i < m->cstream.size(); ++i) {
    (void)hipSetDevice(m->dev[i]);
    if (hipStreamSynchronize(m->cstream[i]) != hipSuccess) rc = HSPF_E_HIP;
  }
  for (auto &p : m->pending) m->free_events[p.local].push_back(p.ev);
  m->pending.clear();
  return rc;
}

// rows of local device i for this job: its slice of the roots, its tables moved to the slice's first row
static bool multi_slice(const hspf_multi *m, const hspf_multi_graph *g, uint32_t i, uint32_t n_roots, const hspf_result *all,
                        uint32_t &b, uint32_t &e, hspf_result &part) {
  hspf_shard_bounds(n_roots, m->world, m->first_rank + i, &b, &e);
  part = all[i];
  if (e == b) return false;
  const size_t n = g->g[i]->n, off = (size_t)b * n;
  part.dist = all[i].dist + off;
  if (all[i].hops) part.hops = all[i].hops + off;
  if (all[i].vflags_out) part.vflags_out = all[i].vflags_out + off;
  if (all[i].first_hop_mask) part.first_hop_mask = all[i].first_hop_mask + off * all[i].n_mask_words;
  if (all[i].pop_rank) part.pop_rank = all[i].pop_rank + off;
  return true;
}

// an asynchronous gather of an earlier call may still be filling these very tables: wait for THAT one only
static void multi_wait_pending_into(hspf_multi *m, const hspf_result *all) {
  for (size_t k = 0; k < m->pending.size();) {
    auto &p = m->pending[k];
    if (p.key == (const void *)all[p.local].dist) {
      (void)hipEventSynchronize(p.ev);
      m->free_events[p.local].push_back(p.ev);
      m->pending.erase(m->pending.begin() + (long)k);
    } else ++k;
  }
}

// the exchange behind a run (hspf_multi_run / hspf_multi_run_wait): the tables selected in `gather`, all-gathered in place
static int multi_gather(hspf_multi *m, size_t n, uint32_t n_roots, hspf_result *all, uint32_t gather) {
  const uint32_t nl = (uint32_t)m->ctx.size();
  if (!(gather & 0xFu) || m->world == 1) return HSPF_OK;
  const bool async = (gather & HSPF_GATHER_ASYNC) != 0;
  std::vector<void *> tabs(nl);
  struct T { uint32_t bit; size_t row_bytes; int which; };
  for (const T &t : {T{HSPF_GATHER_DIST, n * 4, 0}, T{HSPF_GATHER_HOPS, n * 2, 1}, T{HSPF_GATHER_FLAGS, n * 2, 2}, T{HSPF_GATHER_MASK, 0, 3}}) {
    if (!(gather & t.bit)) continue;
    bool have = true;
    for (uint32_t i = 0; i < nl; ++i) {
      void *p = t.which == 0 ? (void *)all[i].dist : t.which == 1 ? (void *)all[i].hops : t.which == 2 ? (void *)all[i].vflags_out : (void *)all[i].first_hop_mask;
      have = have && p != nullptr; tabs[i] = p;
    }
    if (!have) continue;
    const size_t rb = t.which == 3 ? n * 8 * all[0].n_mask_words : t.row_bytes;
    const int rc = allgather_rows_impl(m, tabs.data(), rb, n_roots, !async);
    if (rc) return rc;
  }
  if (async) {
    for (uint32_t i = 0; i < nl; ++i) {
      hipEvent_t ev = nullptr;
      (void)hipSetDevice(m->dev[i]);
      auto &pool = m->free_events[i];
      if (!pool.empty()) { ev = pool.back(); pool.pop_back(); }
      else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { m->last_error = "gather event"; return HSPF_E_HIP; }
      if (hipEventRecord(ev, m->cstream[i]) != hipSuccess) {
        // no marker for this gather: finish it here, so that a later run into the same tables cannot overtake it
        pool.push_back(ev);
        if (hipStreamSynchronize(m->cstream[i]) != hipSuccess) { m->last_error = "gather event record / stream sync"; return HSPF_E_HIP; }
        continue;
      }
      m->pending.push_back(hspf_multi::Pending{(const void *)all[i].dist, ev, i});
    }
  }
  return HSPF_OK;
}

int hspf_multi_run_async(hspf_multi *m, const hspf_multi_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                         const hspf_result *all, uint64_t *ticket) {
  if (!m || !g || !roots || !all || !ticket || n_roots == 0 || g->g.size() != m->ctx.size()) return HSPF_E_INVAL;
  const uint32_t nl = (uint32_t)m->ctx.size();
  for (uint32_t i = 0; i < nl; ++i) if (!all[i].dist) return HSPF_E_INVAL;       // every table set is looked at BEFORE anything is handed to a lane
  multi_wait_pending_into(m, all);
  try {
    hspf_multi::Ticket tk{m->next_ticket, std::vector<uint64_t>(nl, 0), std::vector<uint8_t>(nl, 0), n_roots, (size_t)g->g[0]->n};
    for (uint32_t i = 0; i < nl; ++i) {
      uint32_t b, e; hspf_result part;
      m->ctx[i]->stats = hspf_stats{};
      if (!multi_slice(m, g, i, n_roots, all, b, e, part)) continue;
      const int rc = hspf_run_device_async(m->ctx[i], g->g[i], roots + b, e - b, run_flags, &part, &tk.t[i]);
      if (rc) {                                       // what was handed out already is waited for: no run may outlive its tables
        for (uint32_t j = 0; j < i; ++j) if (tk.has[j]) (void)hspf_wait(m->ctx[j], tk.t[j], nullptr);
        m->last_error = m->ctx[i]->last_error; return rc;
      }
      tk.has[i] = 1;
    }
    m->tickets.push_back(std::move(tk));
  } catch (...) { m->last_error = "hspf_multi_run_async: host allocation failed"; return HSPF_E_NOMEM; }
  *ticket = m->next_ticket++;
  return HSPF_OK;
}

int hspf_multi_run_wait(hspf_multi *m, uint64_t ticket, hspf_result *all, uint32_t gather) {
  if (!m || !all) return HSPF_E_INVAL;
  size_t k = 0;
  while (k < m->tickets.size() && m->tickets[k].id != ticket) ++k;
  if (k == m->tickets.size()) return HSPF_E_INVAL;
  const hspf_multi::Ticket tk = m->tickets[k];
  m->tickets.erase(m->tickets.begin() + (long)k);
  int rc = HSPF_OK;
  for (uint32_t i = 0; i < (uint32_t)m->ctx.size(); ++i) {
    if (!tk.has[i]) continue;
    const int r = hspf_wait(m->ctx[i], tk.t[i], nullptr);         // (leaves the run's statistics in ctx[i]->stats)
    if (r && rc == HSPF_OK) { rc = r; m->last_error = m->ctx[i]->last_error; }
  }
  if (rc) return rc;
  return multi_gather(m, tk.n_vertices, tk.n_roots, all, gather);
}

int hspf_multi_run(hspf_multi *m, const hspf_multi_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                   hspf_result *all, uint32_t gather) {
  if (!m || !g || !roots || !all || n_roots == 0 || g->g.size() != m->ctx.size()) return HSPF_E_INVAL;
  const uint32_t nl = (uint32_t)m->ctx.size();
  if (nl == 1) {                                       // one local device: on the caller's thread, on the context's own stream
    multi_wait_pending_into(m, all);
    uint32_t b, e; hspf_result part;
    m->ctx[0]->stats = hspf_stats{};
    if (!all[0].dist) return HSPF_E_INVAL;
    if (multi_slice(m, g, 0, n_roots, all, b, e, part)) {
      const int rc = hspf_run_device(m->ctx[0], g->g[0], roots + b, e - b, run_flags, &part);
      if (rc) { m->last_error = m->ctx[0]->last_error; return rc; }
    }
  } else {                                             // several: every device's slice on a lane of its context, all at once
    uint64_t t = 0;
    int rc = hspf_multi_run_async(m, g, roots, n_roots, run_flags, all, &t);
    if (rc == HSPF_OK) rc = hspf_multi_run_wait(m, t, all, 0u);
    if (rc) return rc;
  }
  return multi_gather(m, g->g[0]->n, n_roots, all, gather);
}

int hspf_multi_get_stats(const hspf_multi *m, uint32_t i, hspf_stats *out) {
  if (!m || i >= m->ctx.size() || !out) return HSPF_E_INVAL;
  *out = m->ctx[i]->stats;
  return HSPF_OK;
}

}  // extern "C"
