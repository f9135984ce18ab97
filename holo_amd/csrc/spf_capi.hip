// spf_capi.hip — host side of libholo_spf_hip.so: the C ABI of include/holo_spf_hip.h.
//
// Replaces, at function granularity, the SPT loops of
//   holo-ospf/src/spf.rs:587-729 (run_area) and holo-isis/src/spf.rs:527-709 (compute_spt);
// the batched caller is holo-isis/src/flooding/manet.rs:47-69.  No CPU fallback lives here: if
// there is no HIP device, hspf_init fails and the caller keeps its own loop.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared spf_capi.hip -o libholo_spf_hip.so
#include "../../include/holo_spf_hip.h"
#include "spf_kernels.hip.h"
#include "spf_repair.hip.h"
#include "graph_build.hip.h"
#include "graph_patch.hip.h"
#include "hub_sort.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>

using namespace hspf;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct hspf_graph {
  uint32_t n = 0, e = 0, e_kept = 0;
  uint32_t cap_e = 0;                // link capacity of the device arrays (>= e; patches grow into it)
  uint32_t max_path_metric = 0;
  uint32_t wmax = 0;                 // largest cost among the kept links
  // (atomic: the lanes of an asynchronous context run on the same graph handle from their own host threads)
  mutable std::atomic<bool> narrow_bad{false};   // a run overflowed the 4-byte fused state: use the 8-byte one
  mutable std::atomic<bool> wide24_bad{false};   // a run overflowed the hop field of the 8-byte state with > 16 mask bits
  mutable std::atomic<bool> lean_bad{false};     // a run overflowed the fields of the lean 4-byte state (k_fused_lean): k_fused from now on
  bool hopcount_like = false;        // every kept link into a network costs 0 (from a router), into a router 1
  bool heavy_rows = false;           // a quarter or more of the links sit in rows of more than 32 (fat-tree switches, big LANs)
  uint32_t xcd_start[9] = {};        // work-balanced chunk ranges of the 8 XCDs (GraphDev::xcd_start)
  uint32_t xcd_heavy(int x) const { return (uint32_t)((uint64_t)n_heavy_chunks * 4u * (uint32_t)x / 8ull); }   // even shares of the heavy units
  uint32_t xcd_blocks() const {
    uint32_t m = 0;
    for (int x = 0; x < 8; ++x) m = std::max(m, xcd_start[x + 1] - xcd_start[x] + xcd_heavy(x + 1) - xcd_heavy(x));
    return 8u * std::max(m, 1u);
  }
  // host mirrors: slot tables walk the root's neighbourhood on the host.  Row v of the caller's CSR = the pool entries
  // [rstart[v], rstart[v] + rlen[v]) of `col` / `twoway`.  An upload lays the rows out back to back; a structural patch
  // rewrites a replaced row where it is when it does not grow and appends it to the pool otherwise — O(row), where the
  // compact mirror of rounds 3-5 moved on average half of its 5 MB per one-row patch (0.02-0.17 ms at a million links: more
  // than the device side of the patch takes since round 6).  The pool is compacted when half of it is dead.
  std::vector<uint32_t> rstart, rlen, col;
  std::vector<uint8_t> twoway;     // per pool entry: 1 = the target's row lists the source (computed on device, copied back; kept by patches)
  std::vector<uint8_t> vflags;
  size_t pool_dead = 0;            // pool entries that belong to no row
  bool pool_packed = true;         // the rows lie back to back in vertex order (as after an upload)
  uint32_t rb(uint32_t v) const { return rstart[v]; }
  uint32_t re(uint32_t v) const { return rstart[v] + rlen[v]; }
  bool pool_compact() const { return pool_packed; }
  void mirror_bounds(const uint32_t *row_ptr) {                  // rows back to back, as the caller's CSR has them
    rstart.assign(row_ptr, row_ptr + n); rlen.resize(n);
    for (uint32_t v = 0; v < n; ++v) rlen[v] = row_ptr[v + 1] - row_ptr[v];
    pool_dead = 0; pool_packed = true;
  }
  void mirror_compact() {                                       // O(links): rows back to back again, in vertex order
    if (pool_compact()) return;
    std::vector<uint32_t> c2((size_t)e);
    std::vector<uint8_t> t2(twoway.size() == col.size() ? (size_t)e : 0);
    size_t o = 0;
    for (uint32_t v = 0; v < n; ++v) {
      if (rlen[v]) {
        memcpy(c2.data() + o, col.data() + rstart[v], (size_t)rlen[v] * 4);
        if (!t2.empty()) memcpy(t2.data() + o, twoway.data() + rstart[v], rlen[v]);
      }
      rstart[v] = (uint32_t)o; o += rlen[v];
    }
    const bool had_tw = twoway.size() == col.size();
    col.swap(c2);
    if (had_tw) twoway.swap(t2); else twoway.clear();
    pool_dead = 0; pool_packed = true;
  }
  // device: one allocation, carved by layout()
  char *arena = nullptr;
  size_t arena_bytes = 0;
  int cur = 0;                                                    // which raw set is live
  uint32_t *d_row_ptr[2] = {}, *d_col[2] = {}, *d_metric[2] = {};  // caller's CSR (ping-pong for patches)
  uint32_t *d_in_ptr = nullptr, *d_in_src = nullptr, *d_in_w = nullptr, *d_in_fpos = nullptr;
  uint32_t *d_out_ptr = nullptr, *d_out_dst = nullptr, *d_out_w = nullptr, *d_out_fpos = nullptr;
  uint8_t *d_vflags = nullptr, *d_rowflags = nullptr, *d_leaf = nullptr;   // d_leaf: GraphDev::leaf
  uint8_t *d_rowaux = nullptr;                                     // RA_* per row (graph_build.hip.h)
  uint32_t *d_in_src2 = nullptr, *d_in_w2 = nullptr, *d_in_fpos2 = nullptr, *d_out_dst2 = nullptr, *d_out_w2 = nullptr, *d_out_fpos2 = nullptr;
  uint8_t *d_zcyc = nullptr, *d_zcyc_tmp = nullptr;                // GraphDev::zcyc (+ the other buffer of the trimming rounds)
  bool zcyc_valid = false;                                        // d_zcyc describes the current links (zcyc_update)
  uint32_t n_leaf = 0;
  uint32_t *d_unit_first = nullptr;                               // work units (GraphDev::unit_first), n / 4 + 4 entries
  uint32_t *d_giant = nullptr;                                    // giant rows: vertex list [n_giant] | first slices [n_giant + 1]
  uint32_t *d_ell_so = nullptr, *d_ell_w = nullptr, *d_ell_od = nullptr;   // fixed-stride link records of k_fused_lean (kb_ell): 16 per vertex, rows 0 .. n
  uint32_t n_giant = 0, n_giant_slices = 0;                       // rows of more than GIANT_DEG in-links (GraphDev::giant_vtx)
  uint32_t n_heavy_chunks = 0;                                    // > 0: the kernels go through unit_first
  uint64_t build_id = 0;                                          // changes with every device build (upload, patch)
  uint32_t max_in_deg = 0;                                        // largest kept in-degree
  uint32_t n_zero_rows = 0, n_bad_rows = 0, any_rowflags = 0;    // BuildInfo's counts, kept current by cost patches
  bool hc_net = false, any_net = false;                           // a network row has a kept in-link / a network vertex exists
  // of the caller's rows (host mirrors), kept current by hspf_graph_patch without walking all rows:
  uint32_t max_out = 0, n_net = 0;                                // longest row; network vertices
  uint64_t heavy_links = 0;                                       // links in rows of more than 32
  void host_summary_full() {
    max_out = 0; n_net = 0; heavy_links = 0;
    for (uint32_t v = 0; v < n; ++v) {
      const uint32_t d = rlen[v];
      max_out = std::max(max_out, d);
      if (d > 32u) heavy_links += d;
      if (vflags[v] & HSPF_VF_NETWORK) ++n_net;
    }
  }
  bool hub_built = false;                                         // the last build ran in hub mode (sorted keys)
  bool costs_only = false;                                        // the last patch changed costs only: nothing was rebuilt
  bool patched_in_place = false;                                  // the last patch was structural and took the incremental path (graph_patch.hip.h)
  bool lean = false;                                              // no network vertex, no static row flag, in-degrees <= 8: k_single_lean
  // A patch that failed after it had started to rewrite the device arrays or the host mirrors (HIP error, allocation failure)
  // leaves the two out of step: the graph is marked and every later call on it returns HSPF_E_INVAL — the caller frees it
  // and uploads again (include/holo_spf_hip.h: "after HSPF_E_HIP / HSPF_E_NOMEM the graph must be freed").
  bool invalid = false;
  // Carves the arrays out of `base` for n vertices / cap links; returns the bytes needed.
  size_t layout(char *base, uint32_t nv, uint32_t cap) {
    size_t off = 0;
    auto carve = [&](size_t bytes) { char *p = base ? base + off : nullptr; off = (off + bytes + 255) & ~size_t(255); return p; };
    const size_t vb = (size_t(nv) + 17) * 4, lb = (size_t(cap) + 16) * 4;
    for (int i = 0; i < 2; ++i) {
      d_row_ptr[i] = (uint32_t *)carve(vb); d_col[i] = (uint32_t *)carve(lb); d_metric[i] = (uint32_t *)carve(lb);
    }
    d_in_ptr = (uint32_t *)carve(vb); d_out_ptr = (uint32_t *)carve(vb);
    d_in_src = (uint32_t *)carve(lb); d_in_w = (uint32_t *)carve(lb); d_in_fpos = (uint32_t *)carve(lb);
    d_out_dst = (uint32_t *)carve(lb); d_out_w = (uint32_t *)carve(lb); d_out_fpos = (uint32_t *)carve(lb);
    d_vflags = (uint8_t *)carve(nv); d_rowflags = (uint8_t *)carve(nv); d_leaf = (uint8_t *)carve(nv); d_rowaux = (uint8_t *)carve(nv);
    // the other set of the six link arrays: a structural patch writes the shifted arrays there and the sets swap (graph_patch.hip.h)
    d_in_src2 = (uint32_t *)carve(lb); d_in_w2 = (uint32_t *)carve(lb); d_in_fpos2 = (uint32_t *)carve(lb);
    d_out_dst2 = (uint32_t *)carve(lb); d_out_w2 = (uint32_t *)carve(lb); d_out_fpos2 = (uint32_t *)carve(lb);
    d_unit_first = (uint32_t *)carve((size_t(nv) / 4 + 8) * 4);
    d_giant = (uint32_t *)carve((size_t(cap) / (GIANT_DEG / 2) + 8) * 4);
    d_ell_so = (uint32_t *)carve((size_t(nv) + 1) * 64); d_ell_w = (uint32_t *)carve((size_t(nv) + 1) * 64); d_ell_od = (uint32_t *)carve((size_t(nv) + 1) * 64);
    d_zcyc = (uint8_t *)carve(nv); d_zcyc_tmp = (uint8_t *)carve(nv);
    if (links_alt) swap_link_sets_ptrs();
    return off;
  }
  bool links_alt = false;            // the live link arrays are the second set (an odd number of incremental structural patches)
  void swap_link_sets_ptrs() {
    std::swap(d_in_src, d_in_src2); std::swap(d_in_w, d_in_w2); std::swap(d_in_fpos, d_in_fpos2);
    std::swap(d_out_dst, d_out_dst2); std::swap(d_out_w, d_out_w2); std::swap(d_out_fpos, d_out_fpos2);
  }
  void swap_link_sets() { swap_link_sets_ptrs(); links_alt = !links_alt; }
  GraphDev dev() const {
    GraphDev g;
    g.n = n; g.e_in = e_kept;
    g.in_ptr = d_in_ptr; g.in_src = d_in_src; g.in_w = d_in_w; g.in_fpos = d_in_fpos;
    g.vflags = d_vflags; g.rowflags = d_rowflags; g.leaf = d_leaf;
    g.zcyc = (zcyc_valid && n_zero_rows) ? d_zcyc : nullptr;
    g.out_ptr = d_out_ptr; g.out_dst = d_out_dst; g.out_w = d_out_w; g.out_fpos = d_out_fpos;
    for (int x = 0; x < 9; ++x) g.xcd_start[x] = xcd_start[x];
    g.unit_first = n_heavy_chunks ? d_unit_first : nullptr;
    g.n_heavy_units = n_heavy_chunks * 4u;
    for (int x = 0; x < 9; ++x) g.xcd_heavy[x] = xcd_heavy(x);
    g.giant_vtx = d_giant; g.giant_slice0 = d_giant + n_giant; g.n_giant = n_giant;
    return g;
  }
};

struct hspf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  std::string last_error;
  hipEvent_t ev[7] = {};             // [6]: behind the flag read-back of a chunk of sweeps (run_phase)
  // scratch (grown on demand, reused across runs)
  DevBuf dist, hv, mask, lane_flags, changed, st64, stamp, hnb, kcnt, swcnt;
  // k_fused_lean's plan: head sweeps, dense passes, one all-due sweep, tail sweeps.  The launches decide on the device
  // whether they still have a job (LEAN_CTL_*, spf_kernels.hip.h); the host only sizes the plan, from what the previous
  // run of this context used (any graph, any roots: a wrong guess costs a few skipped launches or passes, nothing else).
  uint32_t lean_head = 4, lean_passes = 16;          // a first run: four head sweeps, a dense stretch of up to 16 passes
  uint32_t lean_dense_pct = 30;                      // HSPF_DENSE_PCT: a head sweep with this share of the rows due is the last one
  uint32_t lean_stay_pct = 10;                       // HSPF_DENSE_STAY_PCT: the stretch ends when a pass changes a smaller share of the rows
  uint32_t lean_dense_passes = 16;                   // HSPF_DENSE_PASSES: passes per dense launch (1: a launch per dense sweep)
  uint32_t lean_max_passes = 48;                     // longest dense stretch a plan may hold
  uint32_t lean_multi_min_wgs = 4096;                // HSPF_DENSE_MIN_WGS: workgroups per pass from which a dense launch carries several passes
  DevBuf o_dist, o_hops, o_flags, o_mask, o_rank;   // device staging of row-major outputs
  DevBuf o_pack, pk_flag;                            // hspf_run_packed: device staging of the packed words (host destinations); misfit flag of k_pack_full
  char *h_stage = nullptr;                           // pinned: two blocks through which a copy to pageable host memory is staged (copy_to_host)
  size_t h_stage_cap = 0;                            // bytes (both blocks)
  hipEvent_t ev_stage[2] = {};
  DevBuf ex_list, ex_heap, ex_pos;
  DevBuf rp_rank;                                    // pop_rank of repaired roots: keys / items of the two sorts, rocPRIM's temporary storage
  DevBuf dyn_part;                                   // FusedGraph::dyn_part: DYN_PARTS partial LF_DYN arrays of the lane = root sweeps
  DevBuf rp_trace;                                   // HSPF_REPAIR_TRACE (debugging)
  DevBuf rp_z, rp_ord, rp_work, rp_status;          // k_repair (spf_repair.hip.h): zero-cost marks + list, (R, pos), stamps + worklists, status
  uint32_t *h_rp = nullptr; size_t h_rp_cap = 0;    // pinned: the repair's control block (RpCtl)
  uint32_t est_rp_rounds = 6;                        // relaxation rounds launched ahead
  uint32_t est_rp = 10;                              // worklist sweeps launched ahead (the last repair's count + 2)
  DevBuf pf_ptr, pf_vtx, pf_met, pf_org;            // prefix table of hspf_routes_device
  bool pf_shadow_ok = false;                        // the device holds a plain table; what it was uploaded from (HSPF_PFX_RESIDENT):
  uint32_t pf_shadow_nv = 0, pf_res_np = 0, pf_res_ne = 0;
  const void *pf_res_ptr = nullptr, *pf_res_vtx = nullptr, *pf_res_met = nullptr;
  DevBuf pack;                                      // record stream of hspf_routes_pack
  uint32_t last_diff_count = 0;                     // changed pairs of the last hspf_routes_diff_device (hspf_routes_diff_count)
  DevBuf gb_kx;                                                 // hspf_graph_upload_keyed: keys, ranks, resolved targets
  DevBuf gb, gb_delta, gb_hub, giant_part;                      // graph build scratch, patch delta, hub-mode sort buffers
  DevBuf leaf_jobs;                                             // run_classes: the rows derived for leaf roots (LeafRootJob)
  uint32_t hub_deg = HUB_DEG;                       // HSPF_HUB_DEG env: rows with more links than this -> graph build from sorted keys
  uint64_t tw_host_max = UINT64_MAX;                // HSPF_TW_HOST_MAX env: row entries a structural patch may scan to keep the two-way mirror itself (unset: max(4096, links / 32))
  BuildInfo *h_info = nullptr;     // pinned
  int *h_changed = nullptr;        // pinned, h_changed_cap ints: per-sweep "something changed" flags of a phase
  size_t h_changed_cap = 0;
  // host copies of what a run uploads: owned by the ctx so that the asynchronous H2D copies need no stream
  // synchronisation of their own (they are consumed long before the next run overwrites them)
  std::vector<uint32_t> hb_tab_ptr, hb_tab_vtx, hb_tab_base;
  // everything a run uploads (roots, slot tables, the fused kernel's descriptor) goes through ONE pinned staging
  // buffer and ONE asynchronous H2D copy
  DevBuf up;
  uint32_t *h_up = nullptr;        // pinned: two halves — the block of the last upload and the one being built
  size_t h_up_cap = 0;             // bytes (both halves)
  int h_up_sel = 0;                // which half the next block is built in
  bool up_valid = false;           // the device block holds the other half's content (up_len bytes)
  size_t up_len = 0;
  std::vector<uint32_t> mark;      // visited stamps of build_slot_table, kept across calls (no O(n) fill per run)
  uint32_t mark_epoch = 0;
  uint32_t *h_patch = nullptr;      // pinned staging block of the cost-only patch
  std::vector<uint32_t> patch_targets;
  std::vector<uint32_t> patch_aff;  // affected rows of an incremental structural patch (graph_patch.hip.h)
  DevBuf gb_pa;                     // ... its BuildInfo, chunk flags, row metadata and staging area
  bool patch_full = false;          // HSPF_PATCH_FULL env (tests, A/B): every structural patch rebuilds the layout, as before round 6
  size_t h_patch_cap = 0;           // words
  uint32_t *h_lane_flags = nullptr; // pinned: per-root status bits, then the 256 rows_done words of the fused kernel
  size_t h_lane_cap = 0;
  uint32_t est_relax = 12, est_dag = 12, est_fused = 12, est_fw = 12;   // launch-ahead estimates (adapted run to run)
  bool est_seen = false;                   // est_fused comes from a run of this context (not the initial guess)
  uint32_t variant = 0;                    // HSPF_VARIANT env: kernel A/B switches (tuning only)
  uint32_t single_attr = 0;                // per k_single instantiation: its dynamic-LDS attribute has been set
  uint32_t single_max_n = 1024;            // HSPF_SINGLE_MAX_N env: largest graph that takes the one-workgroup-per-root kernel
  uint32_t lv_max_roots = 2;               // HSPF_LV_MAX_ROOTS env: runs of at most this many roots take the lane = vertex kernel (0: never)
  uint32_t lv_min_n = 32768;               // HSPF_LV_MIN_N env: ... on graphs of at least this many vertices
  uint32_t est_lv = 24;
  // k_xcd (one XCD per root, the state replicated in every CU's LDS): runs of at most xcd_max_roots roots (HSPF_XCD_MAX_ROOTS
  // env, at most 8, 0: never) on graphs too large for the one-workgroup kernel and of at most XCD_MAX_N vertices.  A run
  // that gave up on one of its barriers (see the kernel) is redone on the launch-per-sweep path and switches the kernel off
  // for this context.
  uint32_t xcd_max_roots = 8;
  bool xcd_always = false;                 // HSPF_XCD_ALWAYS env (tests): every eligible run takes k_xcd, whatever was measured
  bool xcd_off = false;
  uint32_t xcd_epoch = 0;                  // numbers the barrier flags of a run: nothing is cleared between runs
  uint32_t xcd_attr = 0;
  uint32_t xcd_timeout_ms = 2;             // HSPF_XCD_TIMEOUT_MS env (a barrier normally completes in ~1 us, a sweep in ~4: 2 ms is 500 sweeps; was 20)
  bool xcd_skew = false;                   // HSPF_XCD_SKEW env (tests): put a root's workgroups on DIFFERENT XCDs — the run must be redone
  DevBuf xcd_ctl;
  // A fused run's scratch, filled for the NEXT run behind this one's results (k_init_fill costs 14 us at the head of a
  // run, and the GPU idles for longer than that while the host turns a run around): valid for exactly these parameters
  struct Prefill { bool valid = false; uint64_t build_id = 0; uint32_t n = 0, B = 0, esz = 0, n_changed = 0, L = 0; bool kcnt = false;
                   uint32_t ns = 0, fillw = 0; } prefill;      // ns: state rows per batch slab; fillw: the "not reached" word
  uint32_t unit_heavy_deg = UNIT_HEAVY_DEG; // HSPF_UNIT_HEAVY_DEG env: a chunk with a row of more in-links than this runs one row per wave
  uint32_t xcd_row_cost = 8;               // HSPF_XCD_ROW_COST env: fixed cost of a row, in links, when the XCD ranges are cut
  hspf_stats stats = {};
  // ---- asynchronous runs (hspf_run_device_async / hspf_wait): LANES = private engine contexts on this device, each with
  // its own stream, scratch and host thread.  A caller that keeps N runs in flight starts them together, and they stay
  // together: all in their sparse first sweeps at once, all dense at once, all in the tail at once (kernel trace,
  // profiles/r04_notes.md r04d).  That lockstep is what pays: the chains of small dependent launches of the N runs
  // interleave (N sparse sweeps in the time of ~1.3), the dense stretches share the chip at no loss.  What does NOT pay on
  // this hardware is a sparse sweep next to ANOTHER run's dense launch: its 6 250 workgroups wait for slots behind the
  // 100 000 of the dense grid (65-80 us instead of 6), whatever the stream priorities, and with a wave slot per SIMD kept
  // free by an LDS reservation the kernel boundaries of the sparse chain stretch to ~60 us instead — a dense stretch
  // handed from lane to lane (events, or one low-priority stream for all dense launches) measured 118-141 k runs/s
  // against 150 k for lanes left alone (r04d-r04f; removed).  HSPF_ASYNC_LANES (default 3).
  std::vector<struct hspf_lane *> lanes;
  uint64_t next_ticket = 1;
  uint32_t lanes_cfg = 3;
  hipStream_t copy_stream = nullptr;                // device -> host copies of packed tickets (hspf_run_packed_async), next to the lanes' runs
  hspf_ctx *parent = nullptr;                       // in a lane's context: the context the caller holds
};

// hspf_run_packed* (ABI 7): where the packed words of a run go, and what the run reports back.
struct PackedReq {
  void *dst = nullptr;              // host or device destination, rows of the whole call
  size_t cap = 0;                   // bytes
  bool host = true;
  uint8_t *root_status = nullptr;   // host, [n_roots of the whole call] or null
  hspf_packed_layout layout{};      // out
  // set by the group loop of run_impl (calls of more roots than one pass holds):
  size_t row_off = 0;               // first row of this group
  uint32_t min_slots = 0;           // first-hop slots of the whole call: every group uses the same field split
  bool force_wide = false;          // 8-byte words for every group (a group's layout differed from the first one's)
  // set by a lane for an asynchronous ticket with a page-locked destination: the words stay in `dev_stage` (device) and the
  // lane sends them to the host on the context's COPY stream — the lane's own stream goes straight on to its next run
  void *dev_stage = nullptr;
  size_t copy_bytes = 0;            // out: bytes of this run's words in dev_stage
};

// One lane of an asynchronous context: a short queue of jobs, the thread that runs them one after the other, the last
// few results.  (A queue, not a slot: with one slot a lane idles from the end of its run until the caller has woken up,
// collected the result and handed it the next one — ~10 % of a 0.4 ms run from Python.)
struct hspf_lane {
  hspf_ctx *sub = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  struct Job {
    const hspf_graph *g; std::vector<uint32_t> roots; uint32_t flags; hspf_result out; uint64_t ticket;
    std::vector<uint32_t> dest; uint32_t n_total = 0;   // a class of a larger run (run_classes): output row of each root, rows of the whole run
    bool packed = false; PackedReq pk;                  // hspf_run_packed_async: the run and its copy to the host
    bool pk_pinned = false;                             // ... into page-locked memory: the copy goes to the context's copy stream
  };
  std::deque<Job> jobs;                // waiting, in ticket order
  bool running = false, quit = false;
  struct Done { uint64_t ticket = 0; int rc = 0; hspf_stats st{}; std::string err; hspf_packed_layout layout{}; hipEvent_t copy_ev = nullptr; } done[8];
  // packed tickets: two device staging blocks per lane, each with the event behind the last copy out of it
  DevBuf stage[2];
  hipEvent_t stage_ev[2] = {};
  bool stage_busy[2] = {false, false};
  int stage_sel = 0;
  uint64_t last_done = 0;
  bool idle() const { return jobs.empty() && !running; }
};
constexpr size_t LANE_QUEUE_MAX = 3;   // jobs waiting per lane (+ the one running): results of a ticket survive 8 later ones of its lane

namespace {

constexpr uint32_t CHANGED_CAP = 1u << 20;   // max launches per phase
constexpr uint32_t HSPF_MAX_LINKS = (1u << 30) - 16u;

#define HIPCHK(ctx, call)                                                          \
  do {                                                                             \
    hipError_t _e = (call);                                                        \
    if (_e != hipSuccess) {                                                        \
      (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(_e);       \
      return _e == hipErrorOutOfMemory ? HSPF_E_NOMEM : HSPF_E_HIP;                \
    }                                                                              \
  } while (0)

// Nothing may unwind through the C boundary (the caller is Rust / ctypes): every entry point that can allocate host
// memory runs its body through this.
template <typename F>
int guarded(hspf_ctx *ctx, F &&body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    if (ctx) { try { ctx->last_error = "host allocation failed"; } catch (...) {} }
    return HSPF_E_NOMEM;
  } catch (...) {
    if (ctx) { try { ctx->last_error = "unexpected C++ exception"; } catch (...) {} }
    return HSPF_E_INTERNAL;
  }
}

// run_scratch = false: a buffer that neither the prefilled scratch nor the run's upload block refers to (graph-patch
// staging, sweep counters): growing it leaves both valid
int ensure(hspf_ctx *ctx, DevBuf &b, size_t bytes, bool run_scratch = true) {
  if (bytes <= b.cap) return HSPF_OK;
  if (run_scratch) { ctx->prefill.valid = false; ctx->up_valid = false; }
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + bytes / 8;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    want = bytes;
    e = hipMalloc(&b.p, want);
  }
  if (e != hipSuccess) {
    ctx->last_error = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
    b.p = nullptr; b.cap = 0;
    return HSPF_E_NOMEM;
  }
  b.cap = want;
  return HSPF_OK;
}

void release(DevBuf &b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

// Slot table of one root (include/holo_spf_hip.h): H = [root] ++ BFS over network vertices
// reachable through network vertices only, links in row order, two-way links only.
void build_slot_table(const hspf_graph *g, uint32_t root, std::vector<uint32_t> &hv,
                      std::vector<uint32_t> &hb, uint32_t &total, std::vector<uint32_t> &mark,
                      uint32_t stamp) {
  hv.clear(); hb.clear();
  hv.push_back(root); hb.push_back(0);
  total = g->rlen[root];
  mark[root] = stamp;
  for (size_t qi = 0; qi < hv.size(); ++qi) {
    const uint32_t p = hv[qi];
    for (uint32_t k = g->rb(p); k < g->re(p); ++k) {
      const uint32_t t = g->col[k];
      if (!g->twoway[k] || !(g->vflags[t] & HSPF_VF_NETWORK) || mark[t] == stamp) continue;
      mark[t] = stamp;
      hv.push_back(t); hb.push_back(total);
      total += g->rlen[t];
    }
  }
}

// A fresh stamp for build_slot_table's visited marks (ctx->mark has at least n entries, none equal to the stamp).
uint32_t next_mark(hspf_ctx *ctx, uint32_t n) {
  if (ctx->mark.size() < n || ctx->mark_epoch >= 0xFFFFFFF0u) { ctx->mark.assign(std::max<size_t>(n, ctx->mark.size()), 0u); ctx->mark_epoch = 0; }
  return ++ctx->mark_epoch;
}

uint32_t round_words(uint32_t w) {   // template instantiations of k_dag / k_emit
  if (w <= 1) return 1;
  if (w <= 2) return 2;
  if (w <= 4) return 4;
  if (w <= 8) return 8;
  return 16;
}

template <int W, bool GS = false>
void launch_dag(dim3 grid, hipStream_t s, GraphDev g, const uint32_t *dist, uint32_t *hv, uint64_t *mask,
                const uint32_t *roots, SlotTabs tabs, uint32_t nn, uint32_t io, int *changed, int sweep,
                uint32_t epoch, uint32_t *lf, uint32_t hc, uint32_t *act) {
  hipLaunchKernelGGL((k_dag<W, GS>), grid, dim3(256), 0, s, g, dist, hv, mask, roots, tabs, nn, io, changed,
                     sweep, epoch, lf, hc, act);
}
template <int W>
void launch_emit(dim3 grid, hipStream_t s, uint32_t n, uint32_t nr, const uint32_t *dist, const uint32_t *hv,
                 const uint64_t *mask, OutDev o, bool leaves, EmitLeaf el) {
  if (leaves) hipLaunchKernelGGL((k_emit<W, true>), grid, dim3(256), 0, s, n, nr, dist, hv, mask, o, el);
  else hipLaunchKernelGGL((k_emit<W, false>), grid, dim3(256), 0, s, n, nr, dist, hv, mask, o, el);
}

// Exclusive prefix sums of in[0..m) into out[0..m], out[m] = total (graph_build.hip.h).
template <typename T>
void gb_scan(hipStream_t s, const T *in, uint32_t m, uint32_t *out, uint32_t *sums) {
  const uint32_t nb = (uint32_t)(((uint64_t)m + 1 + GB_TILE - 1) / GB_TILE);
  hipLaunchKernelGGL((kb_scan_sums<T>), dim3(nb), dim3(GB_BLOCK), 0, s, in, m, sums);
  hipLaunchKernelGGL(kb_scan_mid, dim3(1), dim3(GB_BLOCK), 0, s, sums, nb, (uint32_t *)nullptr);
  hipLaunchKernelGGL((kb_scan_apply<T>), dim3(nb), dim3(GB_BLOCK), 0, s, in, m, (const uint32_t *)sums, out);
}

// Builds everything the SPF kernels read from the live raw CSR of g (d_row_ptr/d_col/d_metric[g->cur],
// d_vflags), on the ctx stream, and refreshes the host-side summary (e_kept, wmax, hop-count shape, two-way
// flags for the slot tables).  Synchronises the stream once at the end.
constexpr int HSPF_RETRY_HUB = 1000;    // build_pass -> build_on_device only

// What a build or a cost patch leaves to derive from the counts it brought back.
void finish_summary(hspf_graph *g) {
  g->hopcount_like = g->n_bad_rows == 0 && g->hc_net;
  g->lean = !g->any_net && g->any_rowflags == 0 && g->max_in_deg <= SINGLE_RL && !g->hopcount_like;
  static std::atomic<uint64_t> next_build{1};
  g->build_id = next_build.fetch_add(1);
  g->narrow_bad = false;
  g->wide24_bad = false;
  g->lean_bad = false;
}

// GraphDev::zcyc after a build or a patch: only graphs with a zero-cost link from a higher- or equal-numbered source need it
// (n_zero_rows), hop-count graphs have their own rule.  A handful of small launches, then the stream is drained: runs on other
// streams (lanes, other contexts) must see the array.
int zcyc_update(hspf_ctx *ctx, hspf_graph *g) {
  g->zcyc_valid = false;
  if (g->n_zero_rows == 0 || g->n == 0) return HSPF_OK;
  hipStream_t s = ctx->stream;
  const uint32_t n = g->n;
  const dim3 gn((n + GB_BLOCK - 1) / GB_BLOCK);
  static_assert(GB_ZC_ROUNDS % 2 == 0, "the last round writes d_zcyc");
  hipLaunchKernelGGL(kb_zc_init, gn, dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr, (const uint32_t *)g->d_in_w, (const uint32_t *)g->d_out_ptr,
                     (const uint32_t *)g->d_out_w, g->d_zcyc);
  for (int r = 0; r < GB_ZC_ROUNDS; ++r) {
    const uint8_t *a = (r & 1) ? g->d_zcyc_tmp : g->d_zcyc;
    uint8_t *b = (r & 1) ? g->d_zcyc : g->d_zcyc_tmp;
    hipLaunchKernelGGL(kb_zc_round, gn, dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr, (const uint32_t *)g->d_in_src, (const uint32_t *)g->d_in_w,
                       (const uint32_t *)g->d_out_ptr, (const uint32_t *)g->d_out_dst, (const uint32_t *)g->d_out_w, a, b);
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s));
  g->zcyc_valid = true;
  return HSPF_OK;
}

// build_launch enqueues the whole build (and the read-back of its counts) on the ctx stream and returns; build_finish
// waits for it and derives the host-side summary.  hspf_graph_patch does its host work (mirrors, two-way flags) between
// the two, behind the kernels.
struct BuildScratch { BuildInfo *info = nullptr; uint8_t *twoway = nullptr; };

int build_launch(hspf_ctx *ctx, hspf_graph *g, bool hub, BuildScratch &bs) {
  const uint32_t n = g->n, e = g->e;
  hipStream_t s = ctx->stream;
  const size_t le = (size_t)e + 16;
  const size_t nsums = (size_t)std::max(e, n) / GB_TILE + 4;
  // scratch: tmp (uint4 x le: the in-row records before ranking), src_of, kpre, lslot (u32 x le each) | in_cnt (n+1) | sums | info | keep, twoway (u8)
  static_assert(sizeof(BuildInfo) <= 32 * 4, "BuildInfo scratch slot");
  const size_t words = 7 * le + (size_t)n + 17 + nsums + 32 + GB_SC_WORDS;
  const size_t bytes = words * 4 + 2 * le + 64;
  int rc = ensure(ctx, ctx->gb, bytes);
  if (rc != HSPF_OK) return rc;
  uint32_t *w = (uint32_t *)ctx->gb.p;
  uint4 *tmp = (uint4 *)w; w += 4 * le;                  // first: 16-byte aligned
  uint32_t *src_of = w; w += le;
  uint32_t *kpre = w; w += le;
  uint32_t *lslot = w; w += le;
  uint32_t *in_cnt = w; w += (size_t)n + 17;
  uint32_t *sums = w; w += nsums;
  BuildInfo *info = (BuildInfo *)w; w += 32 + GB_SC_WORDS;     // + the spread counters (gb_spread), zeroed with it
  uint8_t *keep = (uint8_t *)w;
  uint8_t *twoway = keep + le;
  const uint32_t *row_ptr = g->d_row_ptr[g->cur], *col = g->d_col[g->cur], *metric = g->d_metric[g->cur];
  // hub mode: key | sorted (u64 x le each) | val | perm | srcflag (u32 x le each) | temporary storage of the sorts
  uint64_t *hkey = nullptr, *hsorted = nullptr;
  uint32_t *hval = nullptr, *hperm = nullptr, *hsrcflag = nullptr;
  void *htmp = nullptr;
  size_t htmp_bytes = 0;
  unsigned nbits = 1;
  while (nbits < 32 && (1ull << nbits) < (uint64_t)n) ++nbits;
  if (hub && e) {
    size_t t1 = 0, t2 = 0;
    HIPCHK(ctx, (hipError_t)hub_sort_keys(nullptr, &t1, nullptr, nullptr, e, 32 + nbits, s));
    HIPCHK(ctx, (hipError_t)hub_sort_pairs(nullptr, &t2, nullptr, nullptr, nullptr, nullptr, e, 33 + nbits, s));
    htmp_bytes = std::max(t1, t2);
    rc = ensure(ctx, ctx->gb_hub, 16 * le + 12 * le + htmp_bytes + 256);
    if (rc != HSPF_OK) return rc;
    hkey = (uint64_t *)ctx->gb_hub.p; hsorted = hkey + le;
    hval = (uint32_t *)(hsorted + le); hperm = hval + le; hsrcflag = hperm + le;
    htmp = (void *)(((uintptr_t)(hsrcflag + le) + 255) & ~(uintptr_t)255);
  }

  {
    const uint32_t n_info = 32u + GB_SC_WORDS, n_clear = std::max(n + 1u, n_info);
    hipLaunchKernelGGL(kb_clear, dim3((n_clear + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, s, in_cnt, n + 1u, (uint32_t *)info, n_info);
  }
  const dim3 ge((e + GB_BLOCK - 1) / GB_BLOCK), gn((n + 1 + GB_BLOCK - 1) / GB_BLOCK);
  if (e && hub) {
    hipLaunchKernelGGL(kb_hub_keys, ge, dim3(GB_BLOCK), 0, s, n, e, row_ptr, col, src_of, hkey);
    size_t tb = htmp_bytes;
    HIPCHK(ctx, (hipError_t)hub_sort_keys(htmp, &tb, hkey, hsorted, e, 32 + nbits, s));
    hipLaunchKernelGGL(kb_hub_links, ge, dim3(GB_BLOCK), 0, s, n, e, row_ptr, col, metric, (const uint8_t *)g->d_vflags,
                       (const uint32_t *)src_of, (const uint64_t *)hsorted, twoway, keep, info);
  } else if (e) {
    hipLaunchKernelGGL(kb_links, ge, dim3(GB_BLOCK), 0, s, n, e, row_ptr, col, metric, (const uint8_t *)g->d_vflags,
                       src_of, twoway, keep, in_cnt, lslot, info);
  }
  gb_scan<uint8_t>(s, keep, e, kpre, sums);
  hipLaunchKernelGGL(kb_out_ptr, gn, dim3(GB_BLOCK), 0, s, n, row_ptr, (const uint32_t *)kpre, g->d_out_ptr, info, e);
  if (!(e && hub)) gb_scan<uint32_t>(s, in_cnt, n, g->d_in_ptr, sums);
  if (e && hub) {
    const uint64_t dropped = 1ull << (32 + nbits);                  // behind every (target << 32 | ~cost)
    hipLaunchKernelGGL(kb_hub_scatter, ge, dim3(GB_BLOCK), 0, s, e, row_ptr, col, metric, (const uint8_t *)g->d_vflags,
                       (const uint32_t *)src_of, (const uint8_t *)keep, (const uint32_t *)kpre, g->d_out_dst, g->d_out_w,
                       g->d_out_fpos, hkey, hval, hsrcflag, dropped, info);
    size_t tb = htmp_bytes;
    HIPCHK(ctx, (hipError_t)hub_sort_pairs(htmp, &tb, hkey, hsorted, hval, hperm, e, 33 + nbits, s));
    hipLaunchKernelGGL(kb_hub_in_ptr, gn, dim3(GB_BLOCK), 0, s, n, e, (const uint64_t *)hsorted, g->d_in_ptr);
    hipLaunchKernelGGL(kb_hub_gather, ge, dim3(GB_BLOCK), 0, s, e, (const BuildInfo *)info, (const uint64_t *)hsorted,
                       (const uint32_t *)hperm, (const uint32_t *)hsrcflag, (const uint32_t *)g->d_out_fpos, g->d_in_src,
                       g->d_in_w, g->d_in_fpos);
  } else if (e) {
    hipLaunchKernelGGL(kb_scatter, ge, dim3(GB_BLOCK), 0, s, e, row_ptr, col, metric, (const uint8_t *)g->d_vflags,
                       (const uint32_t *)src_of, (const uint8_t *)keep, (const uint32_t *)kpre,
                       (const uint32_t *)g->d_in_ptr, (const uint32_t *)lslot, g->d_out_dst, g->d_out_w, g->d_out_fpos,
                       tmp, info);
    hipLaunchKernelGGL(kb_rank, ge, dim3(GB_BLOCK), 0, s, e, (const BuildInfo *)info, (const uint32_t *)g->d_in_ptr,
                       (const uint4 *)tmp, g->d_in_src, g->d_in_w, g->d_in_fpos, ctx->hub_deg);
  }
  hipLaunchKernelGGL(kb_rowflags, dim3((uint32_t)(((size_t)n * 16 + GB_BLOCK - 1) / GB_BLOCK)), dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr, (const uint32_t *)g->d_in_src,
                     (const uint32_t *)g->d_in_w, (const uint8_t *)g->d_vflags, g->d_rowflags, g->d_rowaux, info, GIANT_DEG);
  hipLaunchKernelGGL(kb_leaf_mark, gn, dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr, (const uint32_t *)g->d_in_src,
                     (const uint32_t *)g->d_out_ptr, (const uint32_t *)g->d_out_dst, g->d_leaf, info);
  if (e) hipLaunchKernelGGL(kb_leaf_links, ge, dim3(GB_BLOCK), 0, s, e, (const BuildInfo *)info, g->d_in_src, (const uint8_t *)g->d_leaf);
  hipLaunchKernelGGL(kb_ell, dim3((uint32_t)((((size_t)n + 1) * 16 + GB_BLOCK - 1) / GB_BLOCK)), dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr,
                     (const uint32_t *)g->d_in_src, (const uint32_t *)g->d_in_w, (const uint32_t *)g->d_out_ptr, (const uint32_t *)g->d_out_dst,
                     (const uint8_t *)g->d_vflags, g->d_ell_so, g->d_ell_w, g->d_ell_od);
  if ((n + 15u) / 16u <= GB_UNITS_MAX_CHUNKS) {
    // work units, XCD ranges and the pads behind the arrays: heavy flag per chunk (in_cnt is free again), the rest in one workgroup
    const uint32_t nb = (n + 15u) / 16u;
    hipLaunchKernelGGL(kb_unit_count, dim3((nb + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr, in_cnt, ctx->unit_heavy_deg);
    hipLaunchKernelGGL(kb_units_small, dim3(1), dim3(GB_UNITS_THREADS), 0, s, n, (const uint32_t *)g->d_in_ptr, (const uint32_t *)in_cnt, g->d_unit_first, info,
                       ctx->xcd_row_cost, g->d_in_ptr, g->d_out_ptr, g->d_in_src, g->d_in_w, g->d_in_fpos, g->d_out_dst, g->d_out_w, g->d_out_fpos, (BuildInfo *)nullptr);
  } else {
    // work units: heavy flag per chunk -> heavy chunks before each chunk -> unit_first (scratch: in_cnt, n + 17 words, is
    // free again and holds both: nb flags, then nb + 1 positions)
    const uint32_t nb = (n + 15u) / 16u;
    uint32_t *hpos = in_cnt + nb + 1;
    hipLaunchKernelGGL(kb_unit_count, dim3((nb + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr, in_cnt, ctx->unit_heavy_deg);
    gb_scan<uint32_t>(s, in_cnt, nb, hpos, sums);
    hipLaunchKernelGGL(kb_unit_fill, dim3((nb + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, s, n, (const uint32_t *)in_cnt,
                       (const uint32_t *)hpos, g->d_unit_first, info);
    hipLaunchKernelGGL(kb_xcd, dim3(1), dim3(64), 0, s, n, (const uint32_t *)g->d_in_ptr, info, ctx->xcd_row_cost);
    hipLaunchKernelGGL(kb_pads, dim3(1), dim3(64), 0, s, n, (const BuildInfo *)info, g->d_in_ptr, g->d_out_ptr,
                       g->d_in_src, g->d_in_w, g->d_in_fpos, g->d_out_dst, g->d_out_w, g->d_out_fpos);
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_info, info, sizeof(BuildInfo), hipMemcpyDeviceToHost, s));
  bs.info = info; bs.twoway = twoway;
  return HSPF_OK;
}

// fetch_twoway: the per-link two-way flags come back from the device (upload); a patch keeps the host mirror current
// itself (1 MB less over the bus per patch at a million links).
int build_finish(hspf_ctx *ctx, hspf_graph *g, bool hub, const BuildScratch &bs, bool fetch_twoway) {
  const uint32_t n = g->n, e = g->e;
  hipStream_t s = ctx->stream;
  const bool tdbg = getenv("HSPF_PATCH_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tdbg) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[hspf build] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  if (fetch_twoway) {
    g->mirror_compact();             // the device's flags are in row order: the pool must be too (an upload's is; a patch beyond the host's own flag keeping)
    g->twoway.resize(e);
    if (e) HIPCHK(ctx, hipMemcpyAsync(g->twoway.data(), bs.twoway, e, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ctx, hipStreamSynchronize(s));
  lap("wait for the build");
  const BuildInfo bi = *ctx->h_info;
  if (bi.err) {
    ctx->last_error = (bi.err & GB_ERR_COL) ? "col out of range" : "metric 0xFFFFFFFF is reserved";
    return HSPF_E_INVAL;
  }
  g->e_kept = bi.kept;
  g->wmax = bi.wmax;
  g->hc_net = bi.hc_net != 0; g->n_bad_rows = bi.n_bad_rows; g->n_zero_rows = bi.n_zero_rows; g->any_rowflags = bi.any_rowflags;
  g->hopcount_like = !bi.hc_bad && bi.hc_net;
  g->heavy_rows = g->heavy_links * 4u >= (uint64_t)std::max<uint32_t>(e, 1u);   // from the caller's rows (out-degrees)
  if (!hub && bi.max_in_deg > ctx->hub_deg) return HSPF_RETRY_HUB;     // kb_rank left those rows unsorted
  g->hub_built = hub;
  g->costs_only = false;
  g->patched_in_place = false;
  for (int x = 0; x < 9; ++x) g->xcd_start[x] = bi.xcd_start[x];
  g->n_heavy_chunks = bi.n_heavy;
  g->max_in_deg = bi.max_in_deg;
  g->n_leaf = bi.n_leaf;
  g->n_giant = 0; g->n_giant_slices = 0;
  if (bi.max_in_deg > GIANT_DEG) {
    // giant rows (GraphDev::giant_vtx): a rare shape, so the in-row bounds simply come back once and the two small
    // tables are made here
    std::vector<uint32_t> ip((size_t)n + 1), tab, s0;
    HIPCHK(ctx, hipMemcpy(ip.data(), g->d_in_ptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    uint32_t total = 0;
    for (uint32_t v = 0; v < n; ++v) {
      const uint32_t d = ip[v + 1] - ip[v];
      if (d > GIANT_DEG) { tab.push_back(v); s0.push_back(total); total += (d + GIANT_SLICE - 1) / GIANT_SLICE; }
    }
    s0.push_back(total);
    g->n_giant = (uint32_t)tab.size(); g->n_giant_slices = total;
    tab.insert(tab.end(), s0.begin(), s0.end());
    HIPCHK(ctx, hipMemcpy(g->d_giant, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  }
  g->any_net = g->n_net != 0;
  finish_summary(g);
  lap("host summary");
  { const int zr = zcyc_update(ctx, g); if (zr != HSPF_OK) return zr; }
  lap("zero-cost cycles");
  return HSPF_OK;
}

int build_pass(hspf_ctx *ctx, hspf_graph *g, bool hub, bool fetch_twoway) {
  BuildScratch bs;
  int rc = build_launch(ctx, g, hub, bs);
  if (rc != HSPF_OK) return rc;
  return build_finish(ctx, g, hub, bs, fetch_twoway);
}

// Hub mode when some row of the caller's CSR lists more than hub_deg links (then a target row scan of kb_links could be
// that long), or when the plain pass found a row with more kept in-links than that (parallel links piled onto one row:
// only then can the in-degree exceed every out-degree).
int build_on_device(hspf_ctx *ctx, hspf_graph *g) {
  g->host_summary_full();
  const bool hub = g->max_out > ctx->hub_deg;
  int rc = build_pass(ctx, g, hub, true);
  if (rc == HSPF_RETRY_HUB) rc = build_pass(ctx, g, true, true);
  return rc;
}

// ---- incremental structural patch (graph_patch.hip.h).  The raw CSR of g (g->cur) and the vertex flags are already the new
// ones (kb_patch_raw is enqueued); d_aff = the affected rows, ascending, on the device.
constexpr int HSPF_RETRY_REBUILD = 1001;    // patch_finish -> graph_patch_impl only

size_t patch_scratch_words(const hspf_graph *g, uint32_t na) {
  const size_t nb = (g->n + 15u) / 16u;
  return (32u + GB_SC_WORDS) + (nb + 16) + (size_t)PA_META * (na + 1u) + (size_t)na * 3u * (PA_IN_STRIDE + PA_OUT_STRIDE) + 64;
}
int patch_prepare(hspf_ctx *ctx, hspf_graph *g, uint32_t na) { return ensure(ctx, ctx->gb_pa, patch_scratch_words(g, na) * 4, false); }

int patch_launch(hspf_ctx *ctx, hspf_graph *g, uint32_t na, const uint32_t *d_aff) {
  const uint32_t n = g->n, nb = (n + 15u) / 16u;
  hipStream_t s = ctx->stream;
  const size_t n_info = 32u + GB_SC_WORDS;
  uint32_t *w = (uint32_t *)ctx->gb_pa.p;                        // (patch_prepare; BuildInfo + spread counters zeroed by kb_patch_raw)
  BuildInfo *info = (BuildInfo *)w; w += n_info;
  uint32_t *done = (uint32_t *)info + 24;                        // a free word of BuildInfo's 32-word slot: kb_pa_rows' count of finished workgroups
  static_assert(sizeof(BuildInfo) <= 24 * 4, "BuildInfo reaches the patch's counter");
  uint32_t *hf = w; w += (size_t)nb + 16;
  uint32_t *meta = w; w += (size_t)PA_META * (na + 1u);
  uint32_t *st_in = w; w += (size_t)na * 3u * PA_IN_STRIDE;
  uint32_t *st_out = w;
  const uint32_t *row_ptr = g->d_row_ptr[g->cur], *col = g->d_col[g->cur], *metric = g->d_metric[g->cur];
  hipLaunchKernelGGL(kb_pa_rows, dim3(na), dim3(256), 0, s, n, na, d_aff, row_ptr, col, metric, (const uint8_t *)g->d_vflags, (const uint32_t *)g->d_in_ptr,
                     (const uint32_t *)g->d_out_ptr, meta, st_in, st_out, g->d_rowflags, g->d_rowaux, g->d_leaf, g->d_ell_so, g->d_ell_w, g->d_ell_od, GIANT_DEG, g->e_kept, done, info);
  // the new number of kept links is the device's to know; the grid covers the most it can be
  const uint64_t bound = std::max<uint64_t>((uint64_t)n + 1u, std::min<uint64_t>((uint64_t)g->e_kept + (uint64_t)na * PA_IN_STRIDE, (uint64_t)g->e));
  hipLaunchKernelGGL(kb_pa_shift, dim3((uint32_t)((bound + GB_BLOCK - 1) / GB_BLOCK)), dim3(GB_BLOCK), 0, s, n, na, d_aff, (const uint32_t *)meta, (const uint32_t *)st_in,
                     (const uint32_t *)st_out, (const uint32_t *)g->d_in_src, (const uint32_t *)g->d_in_w, (const uint32_t *)g->d_in_fpos, (const uint32_t *)g->d_out_dst,
                     (const uint32_t *)g->d_out_w, (const uint32_t *)g->d_out_fpos, g->d_in_src2, g->d_in_w2, g->d_in_fpos2, g->d_out_dst2, g->d_out_w2, g->d_out_fpos2,
                     g->d_in_ptr, g->d_out_ptr, (const uint8_t *)g->d_leaf, info);
  g->swap_link_sets();                                           // (host pointers only: the launches above hold the old ones)
  hipLaunchKernelGGL(kb_pa_summary, dim3((n + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, s, n, (const uint32_t *)g->d_in_ptr, (const uint8_t *)g->d_rowflags,
                     (const uint8_t *)g->d_rowaux, (const uint8_t *)g->d_leaf, hf, ctx->unit_heavy_deg, info);
  hipLaunchKernelGGL(kb_units_small, dim3(1), dim3(GB_UNITS_THREADS), 0, s, n, (const uint32_t *)g->d_in_ptr, (const uint32_t *)hf, g->d_unit_first, info,
                     ctx->xcd_row_cost, g->d_in_ptr, g->d_out_ptr, g->d_in_src, g->d_in_w, g->d_in_fpos, g->d_out_dst, g->d_out_w, g->d_out_fpos, ctx->h_info);
  HIPCHK(ctx, hipGetLastError());
  return HSPF_OK;
}

int patch_finish(hspf_ctx *ctx, hspf_graph *g) {
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const BuildInfo bi = *ctx->h_info;
  if (bi.err & GB_ERR_PATCH) return HSPF_RETRY_REBUILD;
  if (bi.err) { ctx->last_error = "hspf_graph_patch: internal: link error on the incremental path"; return HSPF_E_INTERNAL; }
  g->e_kept = bi.kept;
  g->wmax = bi.wmax;
  g->hc_net = bi.hc_net != 0; g->n_bad_rows = bi.n_bad_rows; g->n_zero_rows = bi.n_zero_rows; g->any_rowflags = bi.any_rowflags;
  g->heavy_rows = g->heavy_links * 4u >= (uint64_t)std::max<uint32_t>(g->e, 1u);
  g->hub_built = false;
  g->costs_only = false;
  g->patched_in_place = true;
  for (int x = 0; x < 9; ++x) g->xcd_start[x] = bi.xcd_start[x];
  g->n_heavy_chunks = bi.n_heavy;
  g->max_in_deg = bi.max_in_deg;
  g->n_leaf = bi.n_leaf;
  g->n_giant = 0; g->n_giant_slices = 0;
  g->any_net = g->n_net != 0;
  finish_summary(g);
  return zcyc_update(ctx, g);
}

int alloc_arena(hspf_ctx *ctx, hspf_graph *g, uint32_t n, uint32_t cap) {
  const size_t bytes = g->layout(nullptr, n, cap);
  char *base = nullptr;
  hipError_t er = hipMalloc((void **)&base, bytes);
  if (er != hipSuccess) {
    ctx->last_error = std::string("graph arena hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(er);
    g->layout(nullptr, n, cap);
    return er == hipErrorOutOfMemory ? HSPF_E_NOMEM : HSPF_E_HIP;
  }
  g->arena = base; g->arena_bytes = bytes; g->cap_e = cap;
  g->layout(base, n, cap);
  return HSPF_OK;
}

}  // namespace

// The lanes of an asynchronous context are HIP streams of their own; the HIP runtime maps a process's streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue run one after the other.  Context stream +
// three lanes + the caller's own stream(s) are more than four: measured 137 k runs/s with the default against 149 k with
// 6 or 8 queues (bench.py, three steps in flight).  The variable is read when the HIP runtime initialises and belongs to
// the PROCESS, not to this library: since ABI 7 nothing is set from in here (round 4 planted a default from a library
// constructor — a process-global side effect inside someone else's daemon).  A host that keeps runs in flight exports
// GPU_MAX_HW_QUEUES=8 itself before its first HIP call (bench.py and holo_amd/_lib.py do).

extern "C" {

static void lanes_quiesce(hspf_ctx *ctx);
static void lanes_shutdown(hspf_ctx *ctx);
static int lanes_ensure(hspf_ctx *ctx);
static uint64_t lane_submit(hspf_ctx *ctx, hspf_lane::Job &&job);
static int lane_collect(hspf_ctx *ctx, uint64_t ticket, hspf_stats *stats, hspf_packed_layout *layout = nullptr);

uint32_t hspf_abi_version(void) { return HSPF_ABI_VERSION; }

int hspf_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return e == hipErrorNoDevice ? 0 : HSPF_E_HIP;
  return n;
}

int hspf_recommend_cpu(uint32_t n_vertices, uint32_t n_edges, uint32_t n_roots) {
  if (n_vertices == 0 || n_roots == 0) return 1;
  const double n = (double)n_vertices;
  const double dens = std::max(1.0, (double)n_edges / (8.0 * n));
  const double cpu_ms = (double)n_roots * dens * (2.4e-4 * n + 7e-7 * n * n);   // reference-shaped loop, one core
  const double gpu_ms = 0.032 + 3.5e-5 * n;                                     // one launch + one synchronisation + the sweeps (k_single_lean; tools/gpu_r04_probe.py tiny)
  return cpu_ms < gpu_ms ? 1 : 0;
}

const char *hspf_strerror(int code) {
  switch (code) {
    case HSPF_OK: return "ok";
    case HSPF_E_INVAL: return "invalid argument";
    case HSPF_E_NODEV: return "no usable HIP device";
    case HSPF_E_HIP: return "HIP runtime error";
    case HSPF_E_NOMEM: return "out of memory";
    case HSPF_E_TOO_MANY_SLOTS: return "too many first-hop slots for n_mask_words";
    case HSPF_E_INTERNAL: return "internal invariant violated";
    case HSPF_E_NO_PACKED: return "results do not fit packed words";
    default: return "unknown error";
  }
}

const char *hspf_last_error(const hspf_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int hspf_init(int device_ordinal, hspf_ctx **out) {
  if (!out) return HSPF_E_INVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return HSPF_E_NODEV;
  if (device_ordinal < 0 || device_ordinal >= n) return HSPF_E_NODEV;
  hspf_ctx *ctx = new (std::nothrow) hspf_ctx();
  if (!ctx) return HSPF_E_NOMEM;
  ctx->device = device_ordinal;
  if (const char *v = getenv("HSPF_VARIANT")) ctx->variant = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_SINGLE_MAX_N")) ctx->single_max_n = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_DENSE_PASSES")) ctx->lean_dense_passes = std::min<uint32_t>(std::max<uint32_t>((uint32_t)strtoul(v, nullptr, 0), 1u), 32u);
  if (const char *v = getenv("HSPF_DENSE_PCT")) ctx->lean_dense_pct = (uint32_t)strtoul(v, nullptr, 0);   // rows due (% of all) in a head sweep of k_fused_lean from which the dense stretch starts
  if (const char *v = getenv("HSPF_DENSE_STAY_PCT")) ctx->lean_stay_pct = (uint32_t)strtoul(v, nullptr, 0);   // rows changed (% of all) by a dense pass below which the stretch ends
  if (const char *v = getenv("HSPF_DENSE_MIN_WGS")) ctx->lean_multi_min_wgs = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_LEAN_HEAD")) ctx->lean_head = std::min<uint32_t>((uint32_t)strtoul(v, nullptr, 0), 60u);   // a first run's head sweeps
  if (const char *v = getenv("HSPF_LV_MAX_ROOTS")) ctx->lv_max_roots = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_LV_MIN_N")) ctx->lv_min_n = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_XCD_MAX_ROOTS")) ctx->xcd_max_roots = std::min<uint32_t>((uint32_t)strtoul(v, nullptr, 0), XCD_MAX_ROOTS);
  if (const char *v = getenv("HSPF_XCD_ALWAYS")) ctx->xcd_always = strtoul(v, nullptr, 0) != 0;
  if (const char *v = getenv("HSPF_XCD_SKEW")) ctx->xcd_skew = atoi(v) != 0;
  if (const char *v = getenv("HSPF_XCD_TIMEOUT_MS")) ctx->xcd_timeout_ms = (uint32_t)strtoul(v, nullptr, 0);   // (0: every barrier gives up at once — the test of the fallback)
  if (const char *v = getenv("HSPF_XCD_ROW_COST")) ctx->xcd_row_cost = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_UNIT_HEAVY_DEG")) ctx->unit_heavy_deg = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_HUB_DEG")) ctx->hub_deg = (uint32_t)strtoul(v, nullptr, 0);
  if (const char *v = getenv("HSPF_TW_HOST_MAX")) ctx->tw_host_max = strtoull(v, nullptr, 0);
  if (const char *v = getenv("HSPF_PATCH_FULL")) ctx->patch_full = atoi(v) != 0;
  if (const char *v = getenv("HSPF_ASYNC_LANES")) ctx->lanes_cfg = std::min<uint32_t>(std::max<uint32_t>((uint32_t)strtoul(v, nullptr, 0), 1u), 8u);
  if (hipSetDevice(device_ordinal) != hipSuccess) { delete ctx; return HSPF_E_NODEV; }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return HSPF_E_HIP; }
  for (auto &e : ctx->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete ctx; return HSPF_E_HIP; }
  if (hipHostMalloc((void **)&ctx->h_changed, sizeof(int) * 256, hipHostMallocDefault) != hipSuccess) { delete ctx; return HSPF_E_NOMEM; }
  ctx->h_changed_cap = 256;
  if (hipHostMalloc((void **)&ctx->h_info, sizeof(BuildInfo), hipHostMallocDefault) != hipSuccess) { hspf_shutdown(ctx); return HSPF_E_NOMEM; }
  *out = ctx;
  return HSPF_OK;
}

void hspf_shutdown(hspf_ctx *ctx) {
  if (!ctx) return;
  lanes_shutdown(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (DevBuf *b : {&ctx->dist, &ctx->hv, &ctx->mask, &ctx->lane_flags, &ctx->changed,
                    &ctx->st64, &ctx->stamp, &ctx->hnb, &ctx->o_dist, &ctx->o_hops, &ctx->o_flags,
                    &ctx->o_mask, &ctx->o_rank, &ctx->ex_list, &ctx->ex_heap, &ctx->ex_pos, &ctx->rp_rank, &ctx->dyn_part, &ctx->rp_trace, &ctx->rp_z, &ctx->rp_ord, &ctx->rp_work, &ctx->rp_status, &ctx->pf_ptr, &ctx->pf_vtx, &ctx->pf_met, &ctx->pf_org, &ctx->gb_kx, &ctx->gb, &ctx->gb_pa, &ctx->gb_delta, &ctx->gb_hub, &ctx->giant_part, &ctx->leaf_jobs, &ctx->kcnt, &ctx->pack, &ctx->swcnt, &ctx->o_pack, &ctx->pk_flag, &ctx->xcd_ctl})
    release(*b);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (auto &e : ctx->ev_stage) if (e) (void)hipEventDestroy(e);
  if (ctx->h_changed) (void)hipHostFree(ctx->h_changed);
  if (ctx->h_lane_flags) (void)hipHostFree(ctx->h_lane_flags);
  if (ctx->h_rp) (void)hipHostFree(ctx->h_rp);
  if (ctx->h_patch) (void)hipHostFree(ctx->h_patch);
  if (ctx->h_up) (void)hipHostFree(ctx->h_up);
  release(ctx->up);
  if (ctx->h_info) (void)hipHostFree(ctx->h_info);
  for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int hspf_set_stream(hspf_ctx *ctx, void *hip_stream) {
  if (!ctx) return HSPF_E_INVAL;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);        // (a prefill of the next run may still be in flight)
  ctx->prefill.valid = false;
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  return HSPF_OK;
}

void *hspf_get_stream(const hspf_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---- graph ------------------------------------------------------------------------------------

int hspf_graph_upload(hspf_ctx *ctx, const hspf_csr *csr, hspf_graph **out) {
  if (!ctx || !csr || !out) return HSPF_E_INVAL;
  *out = nullptr;
  const uint32_t n = csr->n_vertices, e = csr->n_edges;
  // link arrays are addressed through 32-bit byte offsets (buffer resources of k_fused): at most 2^30 - 16 links
  if (n == 0 || n > (1u << 24) || e > HSPF_MAX_LINKS || !csr->row_ptr || !csr->vflags || (e && (!csr->col || !csr->metric))) {
    ctx->last_error = "hspf_graph_upload: malformed hspf_csr";
    return HSPF_E_INVAL;
  }
  if (csr->row_ptr[0] != 0 || csr->row_ptr[n] != e) { ctx->last_error = "row_ptr[0]!=0 or row_ptr[n]!=n_edges"; return HSPF_E_INVAL; }
  for (uint32_t u = 0; u < n; ++u)
    if (csr->row_ptr[u + 1] < csr->row_ptr[u]) { ctx->last_error = "row_ptr not monotone"; return HSPF_E_INVAL; }
  (void)hipSetDevice(ctx->device);

  const bool tdbg = getenv("HSPF_UPLOAD_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tdbg) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[hspf upload] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  hspf_graph *g = new (std::nothrow) hspf_graph();
  if (!g) return HSPF_E_NOMEM;
  g->n = n; g->e = e; g->max_path_metric = csr->max_path_metric;
  int rc = alloc_arena(ctx, g, n, e + std::max(e / 8, 1024u));
  if (rc != HSPF_OK) { delete g; return rc; }
  lap("device alloc");
  auto fail = [&](int code) { hspf_graph_free(ctx, g); return code; };
  hipStream_t s = ctx->stream;
  hipError_t er = hipMemcpyAsync(g->d_row_ptr[0], csr->row_ptr, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s);
  if (er == hipSuccess && e) er = hipMemcpyAsync(g->d_col[0], csr->col, (size_t)e * 4, hipMemcpyHostToDevice, s);
  if (er == hipSuccess && e) er = hipMemcpyAsync(g->d_metric[0], csr->metric, (size_t)e * 4, hipMemcpyHostToDevice, s);
  if (er == hipSuccess) er = hipMemcpyAsync(g->d_vflags, csr->vflags, n, hipMemcpyHostToDevice, s);
  if (er != hipSuccess) { ctx->last_error = std::string("graph upload H2D: ") + hipGetErrorString(er); return fail(HSPF_E_HIP); }
  lap("H2D");
  try {
    // host mirrors for the slot tables (hspf_slot_table walks the root's neighbourhood on the host)
    g->mirror_bounds(csr->row_ptr);
    g->col.assign(csr->col, csr->col + e);
    g->vflags.assign(csr->vflags, csr->vflags + n);
    g->twoway.resize(e);
  } catch (const std::bad_alloc &) {
    (void)hipStreamSynchronize(s);
    return fail(HSPF_E_NOMEM);
  }
  lap("host mirrors");
  rc = build_on_device(ctx, g);
  lap("device build + sync");
  if (rc != HSPF_OK) return fail(rc);
  *out = g;
  return HSPF_OK;
}

// LSDB records -> CSR on the device -> the usual build (include/holo_spf_hip.h; kernels: graph_build.hip.h kb_kx_*).
int hspf_graph_upload_keyed(hspf_ctx *ctx, const hspf_keyed_lsdb *k, hspf_graph **out, uint32_t *rank_out) {
  if (!ctx || !k || !out) return HSPF_E_INVAL;
  *out = nullptr;
  const uint32_t n = k->n_vertices, m = k->n_links;
  if (n == 0 || n > (1u << 24) || m > HSPF_MAX_LINKS || !k->vertex_key || !k->row_ptr || !k->vflags || (m && (!k->target_key || !k->metric))) {
    ctx->last_error = "hspf_graph_upload_keyed: malformed hspf_keyed_lsdb";
    return HSPF_E_INVAL;
  }
  if (k->row_ptr[0] != 0 || k->row_ptr[n] != m) { ctx->last_error = "row_ptr[0]!=0 or row_ptr[n]!=n_links"; return HSPF_E_INVAL; }
  for (uint32_t u = 0; u < n; ++u)
    if (k->row_ptr[u + 1] < k->row_ptr[u]) { ctx->last_error = "row_ptr not monotone"; return HSPF_E_INVAL; }
  (void)hipSetDevice(ctx->device);
  return guarded(ctx, [&]() -> int {
    hspf_graph *g = new (std::nothrow) hspf_graph();
    if (!g) return HSPF_E_NOMEM;
    g->n = n; g->e = m; g->max_path_metric = k->max_path_metric;
    int rc = alloc_arena(ctx, g, n, m + std::max(m / 8, 1024u));
    if (rc != HSPF_OK) { delete g; return rc; }
    auto fail = [&](int code) { hspf_graph_free(ctx, g); return code; };
    hipStream_t s = ctx->stream;
    // scratch: keys in / sorted, identity / permutation, rank, input rows, target keys, costs, flags, keep, prefix sums, targets,
    // degrees, scan sums, the sort's temporary storage
    size_t tsort = 0;
    if (hub_sort_pairs(nullptr, &tsort, nullptr, nullptr, nullptr, nullptr, n, 64, s) != 0) { ctx->last_error = "hspf_graph_upload_keyed: sort size query"; return fail(HSPF_E_HIP); }
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_kin = carve((size_t)n * 8), o_ks = carve((size_t)n * 8), o_id = carve((size_t)n * 4), o_perm = carve((size_t)n * 4), o_rank = carve((size_t)n * 4),
                 o_vrow = carve(((size_t)n + 1) * 4), o_tk = carve((size_t)m * 8 + 8), o_tm = carve((size_t)m * 4 + 4), o_vf = carve(n), o_keep = carve((size_t)m * 4 + 4),
                 o_kpre = carve(((size_t)m + 2) * 4), o_tidx = carve((size_t)m * 4 + 4), o_deg = carve(((size_t)n + 1) * 4),
                 o_sums = carve((((size_t)std::max(m, n) + 2) / GB_TILE + 4) * 4), o_err = carve(256), o_tmp = carve(tsort + 256);
    if ((rc = ensure(ctx, ctx->gb_kx, off, false))) return fail(rc);
    char *b = (char *)ctx->gb_kx.p;
    uint64_t *kin = (uint64_t *)(b + o_kin), *ks = (uint64_t *)(b + o_ks), *tk = (uint64_t *)(b + o_tk);
    uint32_t *idn = (uint32_t *)(b + o_id), *perm = (uint32_t *)(b + o_perm), *rank = (uint32_t *)(b + o_rank), *vrow = (uint32_t *)(b + o_vrow), *tm = (uint32_t *)(b + o_tm),
             *keep = (uint32_t *)(b + o_keep), *kpre = (uint32_t *)(b + o_kpre), *tidx = (uint32_t *)(b + o_tidx), *deg = (uint32_t *)(b + o_deg), *sums = (uint32_t *)(b + o_sums),
             *err = (uint32_t *)(b + o_err);
    uint8_t *vf = (uint8_t *)(b + o_vf);
    HIPCHK(ctx, hipMemcpyAsync(kin, k->vertex_key, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(vrow, k->row_ptr, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(vf, k->vflags, n, hipMemcpyHostToDevice, s));
    if (m) {
      HIPCHK(ctx, hipMemcpyAsync(tk, k->target_key, (size_t)m * 8, hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(tm, k->metric, (size_t)m * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(ctx, hipMemsetAsync(err, 0, 4, s));
    const dim3 gn((n + GB_BLOCK - 1) / GB_BLOCK), gm((std::max(m, 1u) + GB_BLOCK - 1) / GB_BLOCK);
    hipLaunchKernelGGL(kb_kx_iota, gn, dim3(GB_BLOCK), 0, s, n, idn);
    size_t tb = tsort;
    HIPCHK(ctx, (hipError_t)hub_sort_pairs(b + o_tmp, &tb, kin, ks, idn, perm, n, 64, s));
    hipLaunchKernelGGL(kb_kx_rank, gn, dim3(GB_BLOCK), 0, s, n, (const uint32_t *)perm, (const uint64_t *)ks, (const uint8_t *)vf, rank, g->d_vflags, err);
    if (m) hipLaunchKernelGGL(kb_kx_resolve, gm, dim3(GB_BLOCK), 0, s, m, (const uint64_t *)tk, (const uint64_t *)ks, n, keep, tidx);
    gb_scan<uint32_t>(s, keep, m, kpre, sums);
    hipLaunchKernelGGL(kb_kx_deg, gn, dim3(GB_BLOCK), 0, s, n, (const uint32_t *)perm, (const uint32_t *)vrow, (const uint32_t *)kpre, deg);
    gb_scan<uint32_t>(s, deg, n, g->d_row_ptr[0], sums);
    hipLaunchKernelGGL(kb_kx_scatter, dim3((uint32_t)(((size_t)n * 16 + GB_BLOCK - 1) / GB_BLOCK)), dim3(GB_BLOCK), 0, s, n, (const uint32_t *)rank, (const uint32_t *)vrow,
                       (const uint32_t *)keep, (const uint32_t *)kpre, (const uint32_t *)tidx, (const uint32_t *)tm, (const uint32_t *)g->d_row_ptr[0], g->d_col[0], g->d_metric[0]);
    // the host mirrors (slot tables walk the root's neighbourhood on the host): row bounds, targets, flags come back
    std::vector<uint32_t> h_rp((size_t)n + 1);
    g->col.resize(m); g->vflags.resize(n);
    uint32_t h_err = 0;
    HIPCHK(ctx, hipMemcpyAsync(h_rp.data(), g->d_row_ptr[0], ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, s));
    if (m) HIPCHK(ctx, hipMemcpyAsync(g->col.data(), g->d_col[0], (size_t)m * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(g->vflags.data(), g->d_vflags, n, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, s));
    if (rank_out) HIPCHK(ctx, hipMemcpyAsync(rank_out, rank, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    if (h_err) { ctx->last_error = "hspf_graph_upload_keyed: a vertex key occurs twice"; return fail(HSPF_E_INVAL); }
    g->e = h_rp[n];
    g->mirror_bounds(h_rp.data());
    g->col.resize(g->e);
    g->twoway.resize(g->e);
    rc = build_on_device(ctx, g);
    if (rc != HSPF_OK) return fail(rc);
    *out = g;
    return HSPF_OK;
  });
}

static int graph_patch_impl(hspf_ctx *ctx, hspf_graph *g, const hspf_rows *rows, bool &committed);

// Nothing unwinds through the boundary and nothing half-done is left usable: the whole patch — the in-place cost
// path AND the structural path with its host-side splices, sorts and resizes — runs inside the exception guard, and a
// failure after the first write to the device arrays or the mirrors (`committed`) marks the graph invalid.
int hspf_graph_patch(hspf_ctx *ctx, hspf_graph *g, const hspf_rows *rows) {
  if (!ctx || !g || !rows) return HSPF_E_INVAL;
  if (g->invalid) { try { ctx->last_error = "hspf_graph_patch: the graph is invalid after a failed patch (free it and upload again)"; } catch (...) {} return HSPF_E_INVAL; }
  bool committed = false;
  const int rc = guarded(ctx, [&]() { return graph_patch_impl(ctx, g, rows, committed); });
  if (rc != HSPF_OK && committed) g->invalid = true;
  return rc;
}

static int graph_patch_impl(hspf_ctx *ctx, hspf_graph *g, const hspf_rows *rows, bool &committed) {
  lanes_quiesce(ctx);                 // asynchronous runs of this context may still read the arrays a patch rewrites
  const uint32_t m = rows->n_changed, n = g->n;
  if (m == 0) return HSPF_OK;
  if (!rows->vertex || !rows->row_ptr || !rows->vflags) { ctx->last_error = "hspf_graph_patch: NULL array"; return HSPF_E_INVAL; }
  if (rows->row_ptr[0] != 0) { ctx->last_error = "hspf_graph_patch: row_ptr[0] != 0"; return HSPF_E_INVAL; }
  for (uint32_t j = 0; j < m; ++j) {
    if (rows->vertex[j] >= n || (j && rows->vertex[j] <= rows->vertex[j - 1])) {
      ctx->last_error = "hspf_graph_patch: vertex list must be strictly ascending and < n_vertices";
      return HSPF_E_INVAL;
    }
    if (rows->row_ptr[j + 1] < rows->row_ptr[j]) { ctx->last_error = "hspf_graph_patch: row_ptr not monotone"; return HSPF_E_INVAL; }
  }
  const uint32_t de = rows->row_ptr[m];
  if (de && (!rows->col || !rows->metric)) { ctx->last_error = "hspf_graph_patch: NULL col/metric"; return HSPF_E_INVAL; }
  for (uint32_t k = 0; k < de; ++k) {
    if (rows->col[k] >= n) { ctx->last_error = "col out of range"; return HSPF_E_INVAL; }
    if (rows->metric[k] == 0xFFFFFFFFu) { ctx->last_error = "metric 0xFFFFFFFF is reserved"; return HSPF_E_INVAL; }
  }
  (void)hipSetDevice(ctx->device);
  hipStream_t s = ctx->stream;
  // ---- fast path: every replaced row lists the same targets in the same order with the same flags — only costs differ
  // (kb_pc_apply / kb_pc_resort, graph_build.hip.h): nothing is rebuilt, no per-link array crosses the bus, the host
  // mirrors stay as they are.  HSPF_VARIANT bit 16 switches it off (A/B, tests of the rebuild path).
  if (g->max_in_deg <= 256u && g->n_giant == 0) {
    bool same = true;
    for (uint32_t j = 0; j < m && same; ++j) {
      const uint32_t v = rows->vertex[j], a = g->rb(v), len = g->rlen[v];
      same = rows->row_ptr[j + 1] - rows->row_ptr[j] == len && rows->vflags[j] == g->vflags[v] &&
             (len == 0 || memcmp(rows->col + rows->row_ptr[j], &g->col[a], (size_t)len * 4) == 0);
    }
    if (same) {
      return [&]() -> int {
        // affected targets: those of the replaced rows' links, each once (dropped links name targets whose rows do not
        // hold them; such a row is re-ranked into the order it already has)
        std::vector<uint32_t> &tg = ctx->patch_targets;
        tg.assign(rows->col, rows->col + de);
        std::sort(tg.begin(), tg.end());
        tg.erase(std::unique(tg.begin(), tg.end()), tg.end());
        const uint32_t nt = (uint32_t)tg.size();
        // one staging block, one copy: PatchInfo | changed[m] | dptr[m + 1] | dmet[de] | targets[nt]
        const size_t words = 4 + (size_t)m + (m + 1) + de + nt;
        int rc = ensure(ctx, ctx->gb_delta, words * 4, false);
        if (rc != HSPF_OK) return rc;
        if (ctx->h_patch_cap < words) {
          (void)hipStreamSynchronize(s);
          if (ctx->h_patch) (void)hipHostFree(ctx->h_patch);
          ctx->h_patch = nullptr; ctx->h_patch_cap = 0;
          HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_patch, (words + 1024) * 4, hipHostMallocDefault));
          ctx->h_patch_cap = words + 1024;
        }
        uint32_t *h = ctx->h_patch;
        h[0] = h[1] = h[2] = h[3] = 0;
        memcpy(h + 4, rows->vertex, (size_t)m * 4);
        memcpy(h + 4 + m, rows->row_ptr, ((size_t)m + 1) * 4);
        if (de) memcpy(h + 4 + m + m + 1, rows->metric, (size_t)de * 4);
        if (nt) memcpy(h + 4 + m + m + 1 + de, tg.data(), (size_t)nt * 4);
        uint32_t *d = (uint32_t *)ctx->gb_delta.p;
        PatchInfo *d_pi = (PatchInfo *)d;
        const uint32_t *d_changed = d + 4, *d_dptr = d_changed + m, *d_dmet = d_dptr + m + 1, *d_tg = d_dmet + de;
        committed = true;                                      // from here on the device arrays change in place
        HIPCHK(ctx, hipMemcpyAsync(d, h, words * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kb_pc_apply, dim3(m), dim3(256), 0, s, d_changed, d_dptr, d_dmet, (const uint32_t *)g->d_row_ptr[g->cur],
                           g->d_metric[g->cur], (const uint32_t *)g->d_out_ptr, g->d_out_w, (const uint32_t *)g->d_out_fpos);
        if (nt)
          hipLaunchKernelGGL(kb_pc_resort, dim3(nt), dim3(256), 0, s, m, d_changed, d_dptr, d_dmet, d_tg, (const uint32_t *)g->d_in_ptr,
                             g->d_in_src, g->d_in_w, g->d_in_fpos, (const uint8_t *)g->d_vflags, g->d_rowflags, g->d_rowaux, g->d_ell_so, g->d_ell_w,
                             GIANT_DEG, g->wmax, d_pi);
        HIPCHK(ctx, hipMemcpyAsync(h, d_pi, sizeof(PatchInfo), hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        PatchInfo pi = *(const PatchInfo *)h;
        if (pi.stale) {                                        // a link that carried the largest cost got cheaper: look again
          HIPCHK(ctx, hipMemsetAsync(d_pi, 0, 4, s));
          if (g->e_kept) hipLaunchKernelGGL(kb_pc_wmax, dim3(std::min<uint32_t>((g->e_kept + GB_BLOCK - 1) / GB_BLOCK, 1024u)), dim3(GB_BLOCK), 0, s,
                                            g->e_kept, (const uint32_t *)g->d_out_w, d_pi);
          HIPCHK(ctx, hipMemcpyAsync(h, d_pi, 4, hipMemcpyDeviceToHost, s));
          HIPCHK(ctx, hipStreamSynchronize(s));
          g->wmax = h[0];
        } else {
          g->wmax = std::max(g->wmax, pi.wmax);
        }
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { ctx->last_error = std::string("hspf_graph_patch (costs): ") + hipGetErrorString(le); return HSPF_E_HIP; }
        g->n_zero_rows = (uint32_t)((int64_t)g->n_zero_rows + pi.d_zero);
        g->n_bad_rows = (uint32_t)((int64_t)g->n_bad_rows + pi.d_bad);
        g->any_rowflags = (g->any_rowflags & ~RF_ZERO) | (g->n_zero_rows ? RF_ZERO : 0u);
        finish_summary(g);
        { const int zr = zcyc_update(ctx, g); if (zr != HSPF_OK) return zr; }     // (a no-op unless the graph has zero-cost links from higher-numbered sources)
        g->costs_only = true;
        g->patched_in_place = false;
        ctx->prefill.valid = false;
        return HSPF_OK;
      }();
    }
  }
  const bool tdbg = getenv("HSPF_PATCH_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tdbg) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[hspf patch] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  // ---- structural patch.  The device derives everything from the old arrays and the delta (new row bounds and
  // raw CSR: kb_patch_raw, then the build or the incremental path), all enqueued first; the host mirrors (rows, two-way flags, the summary of the
  // caller's rows) are brought up to date BEHIND the kernels, in O(replaced rows x their targets' rows) + one block move of
  // the mirrors.  (Round 3: mirrors first, 400 KB of row bounds up and 1 MB of two-way flags down the bus, three walks over
  // all rows: 0.59-0.68 ms at 100 000 rows / 1 000 000 links, now hidden or gone: profiles/r04_notes.md.)
  uint64_t e_new64 = g->e;
  uint32_t new_max_len = 0;
  bool max_row_shrinks = false;
  for (uint32_t j = 0; j < m; ++j) {
    const uint32_t v = rows->vertex[j];
    const uint32_t nl = rows->row_ptr[j + 1] - rows->row_ptr[j], ol = g->rlen[v];
    e_new64 += (uint64_t)nl;
    e_new64 -= (uint64_t)ol;
    new_max_len = std::max(new_max_len, nl);
    if (ol == g->max_out && nl < ol) max_row_shrinks = true;
  }
  if (e_new64 > HSPF_MAX_LINKS) { ctx->last_error = "hspf_graph_patch: too many links"; return HSPF_E_INVAL; }
  const uint32_t e_new = (uint32_t)e_new64;
  // grow the arena when the patched graph does not fit (raw CSR and flags move device-to-device)
  bool grown = false;
  if (e_new > g->cap_e) {
    grown = true;
    char *old_arena = g->arena;
    const size_t old_bytes = g->arena_bytes;
    const uint32_t old_cap = g->cap_e;
    const uint32_t *o_rp = g->d_row_ptr[g->cur], *o_col = g->d_col[g->cur], *o_met = g->d_metric[g->cur];
    const uint8_t *o_vf = g->d_vflags;
    const uint32_t cap = e_new + std::max(e_new / 4, 1024u);
    int rc = alloc_arena(ctx, g, n, cap);
    if (rc != HSPF_OK) {
      g->arena = old_arena; g->arena_bytes = old_bytes; g->cap_e = old_cap; g->layout(old_arena, n, old_cap);
      return rc;
    }
    g->cur = 0;
    committed = true;                                          // the graph lives in the new arena from here on
    hipError_t er = hipMemcpyAsync(g->d_row_ptr[0], o_rp, ((size_t)n + 1) * 4, hipMemcpyDeviceToDevice, s);
    if (er == hipSuccess && g->e) er = hipMemcpyAsync(g->d_col[0], o_col, (size_t)g->e * 4, hipMemcpyDeviceToDevice, s);
    if (er == hipSuccess && g->e) er = hipMemcpyAsync(g->d_metric[0], o_met, (size_t)g->e * 4, hipMemcpyDeviceToDevice, s);
    if (er == hipSuccess) er = hipMemcpyAsync(g->d_vflags, o_vf, n, hipMemcpyDeviceToDevice, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    (void)hipFree(old_arena);
    if (er != hipSuccess) { ctx->last_error = std::string("hspf_graph_patch: grow: ") + hipGetErrorString(er); return HSPF_E_HIP; }
  }
  // work of keeping the mirror's two-way flags on the host (see "behind the kernels" below); decided here because the
  // incremental path needs the flags kept by the host (it has no per-link flags to fetch)
  uint64_t tw_work = 0;
  for (uint32_t j = 0; j < m; ++j) {
    const uint32_t v = rows->vertex[j];
    for (uint32_t k = rows->row_ptr[j]; k < rows->row_ptr[j + 1]; ++k) { const uint32_t t = rows->col[k]; tw_work += g->rlen[t] + 1u; }
    for (uint32_t k = g->rb(v); k < g->re(v); ++k) { const uint32_t t = g->col[k]; tw_work += g->rlen[t] + 1u; }
  }
  tw_work += 2ull * de;                                       // rows of replaced targets are read at their new length
  // (a scanned entry costs ~5 ns, a byte of flags over the bus ~0.1 ns + a fixed ~20 us: the patch keeps the flags itself
  // while that is the cheaper side — every ordinary LSP; a replaced hub row of 100 000 links is not)
  const uint64_t tw_bound = ctx->tw_host_max == UINT64_MAX ? std::max<uint64_t>(4096u, e_new / 32u) : ctx->tw_host_max;
  const bool tw_host = tw_work <= tw_bound && g->twoway.size() == g->col.size();
  // ---- the incremental path (graph_patch.hip.h): affected rows = replaced rows + their old and new targets
  std::vector<uint32_t> &aff = ctx->patch_aff;
  aff.clear();
  bool incremental = !ctx->patch_full && tw_host && e_new <= g->cap_e && !grown && g->max_in_deg <= PA_IN_STRIDE && g->n_giant == 0 && !g->hub_built &&
                     std::max(g->max_out, new_max_len) <= std::min(ctx->hub_deg, PA_OUT_STRIDE) && (n + 15u) / 16u <= GB_UNITS_MAX_CHUNKS && g->e_kept != 0;
  if (incremental) {
    aff.assign(rows->vertex, rows->vertex + m);
    aff.insert(aff.end(), rows->col, rows->col + de);
    for (uint32_t j = 0; j < m && aff.size() <= 8u * PA_MAX_ROWS; ++j) {
      const uint32_t v = rows->vertex[j];
      aff.insert(aff.end(), g->col.begin() + g->rb(v), g->col.begin() + g->re(v));
    }
    std::sort(aff.begin(), aff.end());
    aff.erase(std::unique(aff.begin(), aff.end()), aff.end());
    incremental = aff.size() <= PA_MAX_ROWS && patch_prepare(ctx, g, (uint32_t)aff.size()) == HSPF_OK;   // (no room for the scratch: the rebuild)
  }
  const uint32_t na = incremental ? (uint32_t)aff.size() : 0u;
  // the delta, one pinned staging block, one copy: changed[m] | delta_ptr[m+1] | shift[m+1] | delta_col[de] | delta_metric[de] | flags[m] (bytes) | affected[na]
  const size_t dwords = (size_t)m + 2 * ((size_t)m + 1) + 2 * (size_t)de + ((size_t)m + 3) / 4 + na;
  int rc = ensure(ctx, ctx->gb_delta, dwords * 4 + 64, false);
  if (rc != HSPF_OK) return rc;
  if (ctx->h_patch_cap < dwords) {
    (void)hipStreamSynchronize(s);
    if (ctx->h_patch) (void)hipHostFree(ctx->h_patch);
    ctx->h_patch = nullptr; ctx->h_patch_cap = 0;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_patch, (dwords + 1024) * 4, hipHostMallocDefault));
    ctx->h_patch_cap = dwords + 1024;
  }
  {
    uint32_t *h = ctx->h_patch;
    memcpy(h, rows->vertex, (size_t)m * 4);
    memcpy(h + m, rows->row_ptr, ((size_t)m + 1) * 4);
    uint32_t *sh = h + m + (m + 1);
    uint32_t acc = 0;                                       // modulo 2^32 (kb_patch_raw)
    for (uint32_t j = 0; j < m; ++j) {
      sh[j] = acc;
      const uint32_t v = rows->vertex[j];
      acc += (rows->row_ptr[j + 1] - rows->row_ptr[j]) - g->rlen[v];
    }
    sh[m] = acc;
    if (de) {
      memcpy(sh + m + 1, rows->col, (size_t)de * 4);
      memcpy(sh + m + 1 + de, rows->metric, (size_t)de * 4);
    }
    memcpy(sh + m + 1 + 2 * (size_t)de, rows->vflags, m);
    if (na) memcpy(sh + m + 1 + 2 * (size_t)de + ((size_t)m + 3) / 4, aff.data(), (size_t)na * 4);
  }
  uint32_t *d_changed = (uint32_t *)ctx->gb_delta.p;
  uint32_t *d_dptr = d_changed + m, *d_shift = d_dptr + m + 1, *d_dcol = d_shift + m + 1, *d_dmet = d_dcol + de;
  uint8_t *d_nf = (uint8_t *)(d_dmet + de);
  const uint32_t *d_aff = d_dmet + de + ((size_t)m + 3) / 4;
  const int nxt = g->cur ^ 1;
  committed = true;                                            // kb_patch_raw rewrites the vertex flags in place; the mirrors follow
  HIPCHK(ctx, hipMemcpyAsync(d_changed, ctx->h_patch, dwords * 4, hipMemcpyHostToDevice, s));
  {
    // the incremental path's BuildInfo block is zeroed by the same launch (patch_prepare made room for it)
    uint32_t *clr = nullptr; uint32_t n_clr = 0;
    if (incremental) { clr = (uint32_t *)ctx->gb_pa.p; n_clr = 32u + GB_SC_WORDS; }
    const uint32_t span = std::max(std::max(n + 1u, e_new), n_clr);
    hipLaunchKernelGGL(kb_patch_raw, dim3((span + GB_BLOCK - 1) / GB_BLOCK), dim3(GB_BLOCK), 0, s, n, e_new, (const uint32_t *)g->d_row_ptr[g->cur],
                       (const uint32_t *)g->d_col[g->cur], (const uint32_t *)g->d_metric[g->cur], m, (const uint32_t *)d_changed, (const uint32_t *)d_shift,
                       (const uint32_t *)d_dptr, (const uint32_t *)d_dcol, (const uint32_t *)d_dmet, g->d_row_ptr[nxt], g->d_col[nxt], g->d_metric[nxt],
                       (const uint8_t *)d_nf, g->d_vflags, clr, n_clr);
  }
  // the summary of the caller's rows, from the replaced rows alone (the longest row is looked for again only when it
  // was one of them and got shorter)
  const uint32_t e_old = g->e;
  uint32_t max_out_new = std::max(g->max_out, new_max_len);
  if (max_row_shrinks) {
    max_out_new = new_max_len;
    uint32_t j = 0;
    for (uint32_t v = 0; v < n; ++v) {
      if (j < m && rows->vertex[j] == v) { ++j; continue; }
      max_out_new = std::max(max_out_new, g->rlen[v]);
    }
  }
  g->cur = nxt;
  g->e = e_new;
  const bool hub = max_out_new > ctx->hub_deg;
  BuildScratch bs;
  rc = incremental ? patch_launch(ctx, g, na, d_aff) : build_launch(ctx, g, hub, bs);
  if (rc != HSPF_OK) {                                       // nothing of the host side was touched: the graph is unusable only if the device failed
    (void)hipStreamSynchronize(s);
    g->cur = nxt ^ 1; g->e = e_old;
    return rc;
  }
  lap("staging + launches");
  // ---- behind the kernels: the host mirrors.  A replaced row is rewritten where it is when it does not grow, else appended
  // to the pool (hspf_graph::rstart / rlen): O(replaced rows), no block move.
  // Two-way flags of the mirror: a link u -> t is two-way when row t lists u.  Replacing row u changes the flags of u's
  // own links and of the links t -> u in the rows of its old and new targets, nothing else.  Long rows would make the
  // membership scans quadratic, so beyond a work bound the flags come back from the device as after an upload.
  std::vector<uint32_t> old_targets, old_len;
  uint64_t heavy_new = g->heavy_links;
  uint32_t n_net_new = g->n_net;
  try {
    size_t grow = 0;
    old_len.resize(m);
    for (uint32_t j = 0; j < m; ++j) {
      const uint32_t u = rows->vertex[j];
      const uint32_t len = rows->row_ptr[j + 1] - rows->row_ptr[j], ol = g->rlen[u];
      old_len[j] = ol;
      if (len > ol) grow += len;
      if (ol > 32u) heavy_new -= ol;
      if (len > 32u) heavy_new += len;
      if (g->vflags[u] & HSPF_VF_NETWORK) --n_net_new;
      if (rows->vflags[j] & HSPF_VF_NETWORK) ++n_net_new;
      if (tw_host) old_targets.insert(old_targets.end(), g->col.begin() + g->rb(u), g->col.begin() + g->re(u));
    }
    const bool had_tw = g->twoway.size() == g->col.size();
    if (g->col.size() + grow > g->col.capacity()) g->col.reserve(g->col.size() + grow + g->col.size() / 8);
    if (had_tw && g->col.size() + grow > g->twoway.capacity()) g->twoway.reserve(g->col.size() + grow + g->col.size() / 8);
    if (g->col.size() + grow >= (1ull << 32)) throw std::bad_alloc();
  } catch (const std::bad_alloc &) {
    (void)hipStreamSynchronize(s);
    g->cur = nxt ^ 1; g->e = e_old;
    if (incremental) g->swap_link_sets();                     // (patch_launch had swapped the sets; the kernels' output is dropped)
    return HSPF_E_NOMEM;
  }
  {
    // (nothing below allocates: the pool's growth was reserved above)
    const bool keep_tw = g->twoway.size() == g->col.size();
    for (uint32_t j = 0; j < m; ++j) {
      const uint32_t u = rows->vertex[j];
      const uint32_t len = rows->row_ptr[j + 1] - rows->row_ptr[j], ol = g->rlen[u];
      if (len != ol) g->pool_packed = false;
      if (len > ol) {
        g->pool_dead += ol;
        g->rstart[u] = (uint32_t)g->col.size();
        g->col.insert(g->col.end(), rows->col + rows->row_ptr[j], rows->col + rows->row_ptr[j] + len);
        if (keep_tw) g->twoway.resize(g->col.size());
      } else {
        g->pool_dead += ol - len;
        if (len) memcpy(g->col.data() + g->rstart[u], rows->col + rows->row_ptr[j], (size_t)len * 4);
      }
      g->rlen[u] = len;
    }
    if (!keep_tw) g->twoway.clear();
    if (tw_host) {
      // per replaced row u: its own links, then the links t -> u of its old and new targets (old_targets holds the old
      // rows back to back, in the order of rows->vertex)
      const uint32_t *ncol = g->col.data();
      uint8_t *ntw = g->twoway.data();
      size_t ot = 0;
      std::vector<uint32_t> &mine = ctx->patch_targets;         // (scratch of the context: no allocation in the steady state)
      for (uint32_t j2 = 0; j2 < m; ++j2) {
        const uint32_t u = rows->vertex[j2];
        const uint32_t ua = g->rb(u), ub = g->re(u);
        const size_t ol = old_len[j2];
        mine.assign(ncol + ua, ncol + ub);
        std::sort(mine.begin(), mine.end());
        auto lists = [&](uint32_t t) { return std::binary_search(mine.begin(), mine.end(), t); };
        auto back_links = [&](uint32_t t) {                              // links t -> u in the (new) row of t
          const bool two = lists(t);
          bool t_lists_u = false;
          for (uint32_t k = g->rb(t); k < g->re(t); ++k)
            if (ncol[k] == u) { ntw[k] = two ? 1 : 0; t_lists_u = true; }
          return t_lists_u;
        };
        for (uint32_t k = ua; k < ub; ++k) ntw[k] = back_links(ncol[k]) ? 1 : 0;
        for (size_t k = 0; k < ol; ++k) (void)back_links(old_targets[ot + k]);
        ot += ol;
      }
    }
    // half of the pool dead: rows back to back again (O(links), amortised over the patches that made the holes)
    if (g->pool_dead > (size_t)e_new / 2 + 65536 || !tw_host) g->mirror_compact();
  }
  for (uint32_t j = 0; j < m; ++j) g->vflags[rows->vertex[j]] = rows->vflags[j];
  g->max_out = max_out_new; g->heavy_links = heavy_new; g->n_net = n_net_new;
  lap("host mirrors");
  if (incremental) {
    rc = patch_finish(ctx, g);
    if (rc == HSPF_RETRY_REBUILD) rc = build_pass(ctx, g, hub, false);      // an affected row outgrew the staging area: the raw CSR is complete, rebuild from it
  } else {
    rc = build_finish(ctx, g, hub, bs, !tw_host);
  }
  if (rc == HSPF_RETRY_HUB) rc = build_pass(ctx, g, true, !tw_host);
  if (rc != HSPF_OK) (void)hipStreamSynchronize(s);
  lap("wait + summary");
  return rc;
}

int hspf_graph_export(hspf_ctx *ctx, const hspf_graph *g, uint32_t which, void *dst, size_t cap_bytes, size_t *out_bytes) {
  if (!ctx || !g) return HSPF_E_INVAL;
  const void *src = nullptr;
  size_t bytes = 0;
  const size_t nb = ((size_t)g->n + 1) * 4, eb = (size_t)g->e * 4, kb = (size_t)g->e_kept * 4;
  switch (which) {
    case HSPF_GX_ROW_PTR: src = g->d_row_ptr[g->cur]; bytes = nb; break;
    case HSPF_GX_COL: src = g->d_col[g->cur]; bytes = eb; break;
    case HSPF_GX_METRIC: src = g->d_metric[g->cur]; bytes = eb; break;
    case HSPF_GX_VFLAGS: src = g->d_vflags; bytes = g->n; break;
    case HSPF_GX_IN_PTR: src = g->d_in_ptr; bytes = nb; break;
    case HSPF_GX_IN_SRC: src = g->d_in_src; bytes = kb; break;
    case HSPF_GX_IN_COST: src = g->d_in_w; bytes = kb; break;
    case HSPF_GX_IN_POS: src = g->d_in_fpos; bytes = kb; break;
    case HSPF_GX_OUT_PTR: src = g->d_out_ptr; bytes = nb; break;
    case HSPF_GX_OUT_DST: src = g->d_out_dst; bytes = kb; break;
    case HSPF_GX_OUT_COST: src = g->d_out_w; bytes = kb; break;
    case HSPF_GX_OUT_POS: src = g->d_out_fpos; bytes = kb; break;
    case HSPF_GX_ROWFLAGS: src = g->d_rowflags; bytes = g->n; break;
    case HSPF_GX_TWOWAY: bytes = g->e; break;                       // host mirror
    case HSPF_GX_BUILD_MODE: bytes = 4; break;                       // host value
    case HSPF_GX_HOST_ROW_PTR: bytes = nb; break;                    // host mirrors
    case HSPF_GX_HOST_COL: bytes = eb; break;
    case HSPF_GX_ELL_SRC: src = g->d_ell_so; bytes = ((size_t)g->n + 1) * 64; break;
    case HSPF_GX_ELL_COST: src = g->d_ell_w; bytes = ((size_t)g->n + 1) * 64; break;
    case HSPF_GX_ELL_OUT: src = g->d_ell_od; bytes = ((size_t)g->n + 1) * 64; break;
    case HSPF_GX_SUMMARY: bytes = 48; break;                         // host values
    case HSPF_GX_LEAF: src = g->d_leaf; bytes = g->n; break;
    case HSPF_GX_ZCYC: src = g->d_zcyc; bytes = (g->zcyc_valid && g->n_zero_rows) ? g->n : 0; break;
    case HSPF_GX_UNITS: src = g->d_unit_first; bytes = g->n_heavy_chunks ? ((size_t)g->n_heavy_chunks * 3 + (g->n + 15u) / 16u) * 4 : 0; break;
    default: ctx->last_error = "hspf_graph_export: unknown array"; return HSPF_E_INVAL;
  }
  if (out_bytes) *out_bytes = bytes;
  if (!dst) return HSPF_OK;
  if (cap_bytes < bytes) { ctx->last_error = "hspf_graph_export: buffer too small"; return HSPF_E_INVAL; }
  if (which == HSPF_GX_TWOWAY || which == HSPF_GX_HOST_ROW_PTR || which == HSPF_GX_HOST_COL) {
    // the host mirror in the caller's (compact) form: rows back to back in vertex order (the pool holds them wherever patches put them)
    size_t o = 0;
    for (uint32_t v = 0; v < g->n; ++v) {
      if (which == HSPF_GX_HOST_ROW_PTR) ((uint32_t *)dst)[v] = (uint32_t)o;
      else if (which == HSPF_GX_HOST_COL) { if (g->rlen[v]) memcpy((uint32_t *)dst + o, g->col.data() + g->rstart[v], (size_t)g->rlen[v] * 4); }
      else if (g->rlen[v]) {
        if (g->twoway.size() == g->col.size()) memcpy((uint8_t *)dst + o, g->twoway.data() + g->rstart[v], g->rlen[v]);
        else memset((uint8_t *)dst + o, 0, g->rlen[v]);            // (no flags on the host: only on a graph whose last build failed)
      }
      o += g->rlen[v];
    }
    if (which == HSPF_GX_HOST_ROW_PTR) ((uint32_t *)dst)[g->n] = (uint32_t)o;
    return HSPF_OK;
  }
  if (which == HSPF_GX_BUILD_MODE) { const uint32_t m = g->costs_only ? 2u : g->patched_in_place ? 3u : g->hub_built ? 1u : 0u; memcpy(dst, &m, 4); return HSPF_OK; }
  if (which == HSPF_GX_SUMMARY) {
    const uint32_t v[12] = {g->wmax, g->hopcount_like ? 1u : 0u, g->lean ? 1u : 0u, g->any_rowflags, g->n_zero_rows, g->n_bad_rows, g->max_in_deg, g->e_kept,
                            g->max_out, g->n_net, (uint32_t)g->heavy_links, g->heavy_rows ? 1u : 0u};
    memcpy(dst, v, 48);
    return HSPF_OK;
  }
  (void)hipSetDevice(ctx->device);
  if (bytes) {
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return HSPF_OK;
}

void hspf_graph_free(hspf_ctx *ctx, hspf_graph *g) {
  if (!g) return;
  if (ctx) lanes_quiesce(ctx);
  if (ctx) { (void)hipSetDevice(ctx->device); if (ctx->stream) (void)hipStreamSynchronize(ctx->stream); }
  if (g->arena) (void)hipFree(g->arena);
  delete g;
}

uint32_t hspf_graph_n_vertices(const hspf_graph *g) { return g ? g->n : 0; }
uint32_t hspf_graph_n_edges_kept(const hspf_graph *g) { return g ? g->e_kept : 0; }
uint32_t hspf_graph_n_edges(const hspf_graph *g) { return g ? g->e : 0; }

// ---- slots ------------------------------------------------------------------------------------

int hspf_mask_words(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t *out_words) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !g || !roots || !out_words || g->invalid) return HSPF_E_INVAL;
  std::vector<uint32_t> hv, hb;
  uint32_t w = 1;
  for (uint32_t r = 0; r < n_roots; ++r) {
    if (roots[r] == HSPF_NO_ROOT) continue;
    if (roots[r] >= g->n) return HSPF_E_INVAL;
    uint32_t total = 0;
    build_slot_table(g, roots[r], hv, hb, total, ctx->mark, next_mark(ctx, g->n));
    w = std::max(w, (total + 63) / 64);
  }
  *out_words = w;
  return HSPF_OK;
  });
}

int hspf_slot_table(hspf_ctx *ctx, const hspf_graph *g, uint32_t root, uint32_t *h_vertex, uint32_t *h_base,
                    uint32_t cap, uint32_t *out_total_slots) {
  return guarded(ctx, [&]() -> int {
  if (!ctx || !g || root >= g->n || g->invalid) return HSPF_E_INVAL;
  std::vector<uint32_t> hv, hb;
  uint32_t total = 0;
  build_slot_table(g, root, hv, hb, total, ctx->mark, next_mark(ctx, g->n));
  for (uint32_t i = 0; i < hv.size() && i < cap; ++i) {
    if (h_vertex) h_vertex[i] = hv[i];
    if (h_base) h_base[i] = hb[i];
  }
  if (out_total_slots) *out_total_slots = total;
  return (int)hv.size();
  });
}

// ---- run --------------------------------------------------------------------------------------

// row_map (host array, may be null): output row of roots[r] inside `out` (device buffers of total_rows rows; only with
// host_out == false) — used by run_classes to let every class write straight into the caller's row order.
// Device -> host on the ctx stream.  Page-locked destinations (hspf_host_alloc, hipHostMalloc, hipHostRegister) take ONE
// asynchronous copy at bus speed, left in flight (the caller synchronises).  Anything else is staged through two
// page-locked blocks of the context: block k + 1 crosses the bus while the host copies block k out — bounded by one host
// memcpy (~10 GB/s), which is why the wrappers hand out page-locked result buffers.  Returns with the data in place then.
constexpr size_t STAGE_BLOCK = 8u << 20;
static int copy_to_host(hspf_ctx *ctx, void *dst, const void *src_dev, size_t bytes, hipStream_t s) {
  if (bytes == 0) return HSPF_OK;
  hipPointerAttribute_t at{};
  const bool locked = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!locked) (void)hipGetLastError();                             // (an unregistered pointer is reported as an error)
  if (locked) { HIPCHK(ctx, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, s)); return HSPF_OK; }
  if (!ctx->h_stage) {
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_stage, 2 * STAGE_BLOCK, hipHostMallocDefault));
    ctx->h_stage_cap = 2 * STAGE_BLOCK;
    for (auto &e : ctx->ev_stage) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const size_t nb = (bytes + STAGE_BLOCK - 1) / STAGE_BLOCK;
  auto issue = [&](size_t k) -> hipError_t {
    const size_t o = k * STAGE_BLOCK, len = std::min(STAGE_BLOCK, bytes - o);
    hipError_t e = hipMemcpyAsync(ctx->h_stage + (k & 1) * STAGE_BLOCK, (const char *)src_dev + o, len, hipMemcpyDeviceToHost, s);
    return e != hipSuccess ? e : hipEventRecord(ctx->ev_stage[k & 1], s);
  };
  HIPCHK(ctx, issue(0));
  for (size_t k = 0; k < nb; ++k) {
    if (k + 1 < nb) HIPCHK(ctx, issue(k + 1));                       // (its block was copied out at the end of the previous turn)
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_stage[k & 1]));
    const size_t o = k * STAGE_BLOCK;
    memcpy((char *)dst + o, ctx->h_stage + (k & 1) * STAGE_BLOCK, std::min(STAGE_BLOCK, bytes - o));
  }
  return HSPF_OK;
}

static int run_impl(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                    hspf_result *out, bool host_out, const uint32_t *row_map = nullptr, uint32_t total_rows = 0,
                    bool no_fused = false, PackedReq *pk = nullptr);

}  // extern "C"  (the pass below has a member template)

// ---- one pass of run_impl (at most one group of roots): the state of the call, one member function per step / engine path ----
// (round 5: run_impl was one function of ~900 lines; the steps and the engine paths are separate functions now, the
// dispatcher is Run::go.  Behaviour unchanged: the GPU suite and the fuzz campaign ran on this form.)
namespace {
struct Run {
  // the call
  hspf_ctx *ctx; const hspf_graph *g; const uint32_t *roots; uint32_t n_roots, run_flags; hspf_result *out; bool host_out;
  const uint32_t *row_map; uint32_t total_rows; bool no_fused; PackedReq *pk; uint32_t n;
  hspf_stats &st;                                                    // ctx->stats
  std::vector<uint32_t> &tab_ptr, &tab_vtx, &tab_base;               // slot tables of the roots (host)
  Run(hspf_ctx *c, const hspf_graph *gr, const uint32_t *r, uint32_t nr, uint32_t fl, hspf_result *o, bool ho, const uint32_t *rm, uint32_t tr, bool nf, PackedReq *p)
      : ctx(c), g(gr), roots(r), n_roots(nr), run_flags(fl), out(o), host_out(ho), row_map(rm), total_rows(tr), no_fused(nf), pk(p), n(gr->n), st(c->stats),
        tab_ptr(c->hb_tab_ptr), tab_vtx(c->hb_tab_vtx), tab_base(c->hb_tab_base) {}
  // shape of the run
  uint32_t B = 0, L = 0, need_words = 1, max_slots = 0, W = 1, out_words = 1;
  hipStream_t s = nullptr;
  std::chrono::steady_clock::time_point t_entry;
  hspf_ctx::Prefill pf;
  bool want_mask = false, fused = false, narrow = false, lean = false, giant = false, count_rows = false, single = false, lv = false, xcdp = false, xcd_ok = false, use_fw = false, defer = false;
  FusedParams fp_wide{}, fp_narrow{}, fp_lean{};
  size_t rows = 0, rn = 0, giant_tags = 0, up_bytes = 0, w_roots = 0, w_ptr = 0, w_vtx = 0, w_base = 0, w_map = 0, w_fg = 0;
  int rc = 0;
  // device pointers
  OutDev od{};
  uint32_t *d_kcnt = nullptr, *d_rank = nullptr, *d_misfit = nullptr, *d_up = nullptr, *d_dist = nullptr, *d_hv = nullptr, *d_roots = nullptr, *d_lf = nullptr, *d_stamp = nullptr;
  uint64_t *d_mask = nullptr, *d_st = nullptr;
  int *d_changed = nullptr;
  const FusedGraph *d_fg = nullptr;
  GraphDev gd{};
  SlotTabs tabs{};
  uint32_t ignore_ovl = 0, net_nh = 0;
  dim3 grid, fgrid;
  // phases
  std::function<void()> on_retry;      // set by a path whose `post` leaves something behind that a non-final chunk must undo
  std::function<void(uint32_t)> spec_fill;   // fused path: the NEXT run's scratch fill, enqueued behind a chunk's flag read-back
  uint32_t *rb_ctl = nullptr;          // lean path: its plan counters come back with every chunk's flags
  uint32_t chunk_last = 0;             // index of the last sweep of the chunk `post` is enqueued behind
  bool two_events = false, tail_done = false, finished = false, delegated = false;
  uint32_t last_esz = 0, last_ns = 0, last_fillw = 0xFFFFFFFFu, spec_nz = 0;
  bool spec_done = false, emit_reset = false;
  std::vector<uint32_t> ex;            // roots of the sequential kernel
  std::vector<uint32_t> dy;            // roots with a dynamic pop order whose rows k_repair put right (spf_repair.hip.h)
  std::vector<uint32_t> oob;           // dy ++ ex: the rows that did not come out of the emit as they are (device copy: ctx->ex_list)
  // packed results
  bool pk_full = false;
  int pk_mode = -1;                    // which fused_run produced the words: 2 lean, 1 narrow, 0 wide; -1: k_pack_full (wide layout)
  size_t pk_esz = 0;
  // device address of this run's first row of packed words of `esz` bytes: staging for a host destination, else the
  // caller's buffer at the group's row offset
  char *pk_dev(size_t esz) const { return pk->dev_stage ? (char *)pk->dev_stage : pk->host ? (char *)ctx->o_pack.p : (char *)pk->dst + pk->row_off * (size_t)n * esz; }
  int pk_staging() {                         // row-major staging tables of a packed run (k_single / k_lv / k_exact write them)
    int r2;
    if ((r2 = ensure(ctx, ctx->o_dist, rn * 4, false))) return r2;
    if ((r2 = ensure(ctx, ctx->o_hops, rn * 2, false))) return r2;
    if ((r2 = ensure(ctx, ctx->o_flags, rn * 2, false))) return r2;
    if ((r2 = ensure(ctx, ctx->o_mask, rn * 8, false))) return r2;
    return HSPF_OK;
  }
  int prepare();
  int choose_state();
  int prepare_scratch();
  int prepare_outputs();
  int upload_block();
  int init_state();
  template <class Launch, class Post> int run_phase(uint32_t est, uint32_t pre_zeroed, Launch &&launch, uint32_t &n_launch, Post &&post);
  int fused_run(int mode);
  int lv_run();
  int xcd_run();
  int single_run();
  int path_fused();
  int path_wide();
  int repair_roots();
  int exact_roots();
  int packed_finish();
  int deliver();
  int go() {
    if ((rc = prepare()) || (rc = choose_state()) || (rc = prepare_scratch()) || (rc = prepare_outputs()) || (rc = upload_block()) || (rc = init_state())) return rc;
    rc = fused ? path_fused() : path_wide();
    if (rc || delegated) return rc;
    if ((rc = exact_roots())) return rc;
    if ((rc = packed_finish())) return rc;
    return deliver();
  }
};

// Validation of the roots, the first-hop slot tables (host, O(degree) per root), the mask width.
int Run::prepare() {
  for (uint32_t r = 0; r < n_roots; ++r)
    if (roots[r] != HSPF_NO_ROOT && roots[r] >= n) { ctx->last_error = "root out of range"; return HSPF_E_INVAL; }
  if ((run_flags & HSPF_RUN_POP_RANK) && !out->pop_rank) { ctx->last_error = "HSPF_RUN_POP_RANK without pop_rank buffer"; return HSPF_E_INVAL; }
  B = (n_roots + 63) / 64; L = B * 64;
  s = ctx->stream;
  st = hspf_stats{};
  st.n_roots = n_roots; st.n_batches = B;
  t_entry = std::chrono::steady_clock::now();      // hspf_stats::dbg[2..3]: host time of the call (us)
  pf = ctx->prefill;      // what the previous run left for this one; whoever does not take it loses it
  ctx->prefill.valid = false;

  // ---- slot tables (host, O(deg) per root) and mask width
  tab_ptr.assign(L + 1, 0); tab_vtx.clear(); tab_base.clear();
  need_words = 1; max_slots = 0;
  {
    std::vector<uint32_t> hv, hb;
    for (uint32_t r = 0; r < L; ++r) {
      if (r < n_roots && roots[r] != HSPF_NO_ROOT) {
        uint32_t total = 0;
        build_slot_table(g, roots[r], hv, hb, total, ctx->mark, next_mark(ctx, n));
        max_slots = std::max(max_slots, total);
        need_words = std::max(need_words, (total + 63) / 64);
        // entry 0 (the root itself, base 0) is implicit on device
        tab_vtx.insert(tab_vtx.end(), hv.begin() + 1, hv.end());
        tab_base.insert(tab_base.end(), hb.begin() + 1, hb.end());
      }
      tab_ptr[r + 1] = (uint32_t)tab_vtx.size();
    }
  }
  if (pk) max_slots = std::max(max_slots, pk->min_slots);
  want_mask = pk || out->first_hop_mask != nullptr;
  if (pk) out->n_mask_words = 1;                              // (`out` is the wrapper's placeholder in packed mode)
  if (pk && (need_words > 1 || max_slots > (g->wide24_bad ? 16u : 24u) || n >= (1u << 23) || (ctx->variant & 1u))) {
    ctx->last_error = "packed results: a root of the run has more than " + std::to_string(g->wide24_bad ? 16 : 24) + " first-hop slots (or the fused path is off)";
    return HSPF_E_NO_PACKED;
  }
  if (want_mask && out->n_mask_words < need_words) {
    ctx->last_error = "n_mask_words too small: need " + std::to_string(need_words);
    return HSPF_E_TOO_MANY_SLOTS;
  }
  if (need_words > 16) { ctx->last_error = "more than 1024 first-hop slots"; return HSPF_E_TOO_MANY_SLOTS; }
  W = round_words(need_words);
  out_words = want_mask ? out->n_mask_words : W;
  st.n_mask_words = need_words;
  return HSPF_OK;
}

// The state of the run: one fused fixed point over a packed word (4 bytes when the fields fit, else 8) or the wide-mask path.
int Run::choose_state() {
  // Fast path: every root has <= 24 first-hop slots -> one fused fixed point over a packed state
  // (k_fused), 4 bytes per (vertex, root) when the slots, hop counts and distances fit (checked on
  // device, LF_OVERFLOW -> the run is redone with the 8-byte state and the graph remembers), else 8;
  // otherwise distances first, then the SPT-DAG phase with W mask words.
  // HSPF_VARIANT bit0 forces the two-phase path, bit1 forbids the narrow state (A/B measurements).
  // 17-24 slots: the 8-byte state with a wider mask field and correspondingly fewer hop bits (32 - M); a root farther
  // than that many router hops from something raises LF_OVERFLOW and the run is redone on the two-phase path.
  const uint32_t fused_max_slots = g->wide24_bad ? 16u : 24u;
  fused = !no_fused && max_slots <= fused_max_slots && n < (1u << 23) && !(ctx->variant & 1u);
  const uint32_t Mw = std::max(16u, max_slots);
  fp_wide = FusedParams{0u, Mw, Mw == 16u ? 0xFFFFu : (1u << (32u - Mw)) - 1u, 0xFFFFFFFFu, g->max_path_metric, 0xFFFFFFFFu,
                      g->hopcount_like ? 1u : 0u, 0xFFFFFFFFu};
  fp_narrow = fp_wide; fp_lean = fp_wide;
  narrow = false; lean = false;
  const bool pk_wide = pk && pk->force_wide;
  if (fused && !g->narrow_bad && !(ctx->variant & 2u) && !pk_wide) {
    // field split of the 4-byte state: M mask bits = slots of this run, 7 hop bits (6 when that
    // leaves fewer than 13 distance bits), the rest distance; used when a link cost is at most
    // 1/8 of the distance range and there are at least 12 distance bits
    const uint32_t M = std::max(max_slots, 1u);
    const uint32_t H = (32u - M - 7u >= 13u) ? 7u : 6u;
    const uint32_t sh = M + H, D = 32u - sh;
    const uint32_t dmax = (1u << D) - 1u;
    if (D >= 12u && g->wmax < (1u << (D - 3))) {
      narrow = true;
      fp_narrow = FusedParams{sh, M, (1u << H) - 1u, dmax << sh,
                              g->max_path_metric >= dmax ? 0xFFFFFFFFu : (g->max_path_metric << sh),
                              (dmax - g->wmax) << sh, g->hopcount_like ? 1u : 0u, 0xFFFFFFFFu};
    }
  }
  // The lean sweep (k_fused_lean): [dist | 3 tag bits (zero in stored words) | hops | mask].  "Not reached" = inf_t =
  // (dmax - wmax) << sh, so that a DPP add (no saturation) of any cost stays inside 32 bits; a FINAL word at or above
  // ovf_t = (dmax - 2 wmax) << sh, or with the hop field at its maximum, raises LF_OVERFLOW in k_emit_fused (-> lean_bad:
  // the run is redone by k_fused and the graph remembers).  The tag costs 3 of the 32 bits: 7 hop bits when that leaves a
  // distance field of at least 8 x wmax, else 6 and at least 4 x wmax.  Needs max_path_metric beyond the field (nothing to
  // prune inside it), no heavy work units, no giant rows.  HSPF_VARIANT bit 15: k_fused everywhere (A/B).
  if (fused && !g->lean_bad && !(ctx->variant & (2u | 32768u)) && g->n_heavy_chunks == 0 && g->n_giant == 0 && !pk_wide) {
    const uint32_t M = std::max(max_slots, 1u);
    uint32_t H = 7u;
    if (M + 3u + H + 4u > 32u || (1u << (32u - M - 3u - H)) < 8u * (g->wmax + 1u)) H = 6u;
    if (M + 3u + H + 4u <= 32u) {
      const uint32_t sh = M + H + 3u, D = 32u - sh, dmax = (1u << D) - 1u;
      if (dmax >= 4u * (g->wmax + 1u) && g->max_path_metric >= dmax) {
        lean = true;
        fp_lean = FusedParams{sh, M, (1u << H) - 1u, (dmax - g->wmax) << sh, 0xFFFFFFFFu, (dmax - 2u * g->wmax) << sh,
                              g->hopcount_like ? 1u : 0u, (dmax - g->wmax) << sh};
      }
    }
  }
  return HSPF_OK;
}

// Scratch of the pass, the layout of the upload block, the pinned blocks.
int Run::prepare_scratch() {
  // ---- scratch
  rows = (size_t)B * n * 64;
  if (fused) {
    if ((rc = ensure(ctx, ctx->st64, rows * 8))) return rc;
    if ((rc = ensure(ctx, ctx->stamp, (size_t)B * n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->hnb, (size_t)B * n + 8))) return rc;       // (+8: k_init_fused ORs marks into whole words)
  } else {
    if ((rc = ensure(ctx, ctx->dist, rows * 4))) return rc;
    if ((rc = ensure(ctx, ctx->hv, rows * 4))) return rc;
    if ((rc = ensure(ctx, ctx->mask, rows * 8 * W))) return rc;
    if ((rc = ensure(ctx, ctx->stamp, (size_t)B * n * 4))) return rc;
  }
  if ((rc = ensure(ctx, ctx->lane_flags, (size_t)L * 4))) return rc;
  if ((rc = ensure(ctx, ctx->dyn_part, (size_t)DYN_PARTS * L * 4))) return rc;
  if ((rc = ensure(ctx, ctx->changed, (size_t)CHANGED_CAP * 4))) return rc;
  // upload block (u32 words): roots[L] | tab_ptr[L+1] | tab_vtx[nv] | tab_base[nv] | row_map[L] | pad to 16 B | FusedGraph
  const size_t nv = tab_vtx.size();
  w_roots = 0; w_ptr = L; w_vtx = w_ptr + L + 1; w_base = w_vtx + std::max<size_t>(nv, 1);
  w_map = w_base + std::max<size_t>(nv, 1);
  w_fg = (w_map + L + 3) & ~size_t(3);
  up_bytes = w_fg * 4 + sizeof(FusedGraph);
  if ((rc = ensure(ctx, ctx->up, up_bytes))) return rc;
  if (ctx->h_up_cap / 2 < up_bytes) {
    (void)hipStreamSynchronize(s);
    if (ctx->h_up) (void)hipHostFree(ctx->h_up);
    ctx->h_up = nullptr; ctx->h_up_cap = 0; ctx->up_valid = false;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_up, up_bytes * 4, hipHostMallocDefault));
    ctx->h_up_cap = up_bytes * 4;
  }
  if (ctx->h_lane_cap < L) {
    if (ctx->h_lane_flags) (void)hipHostFree(ctx->h_lane_flags);
    ctx->h_lane_flags = nullptr; ctx->h_lane_cap = 0;
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_lane_flags, ((size_t)L + 256 + LEAN_CTL_WORDS + 4) * 4, hipHostMallocDefault));   // status bits | row counters | lean plan counters | misfit flag of a packed run
    ctx->h_lane_cap = L;
  }
  // slices of the giant rows (FusedGraph::giant_part): tags, then GIANT_WORDS x 64 words per (batch, slice)
  giant = g->n_giant != 0 && g->n_heavy_chunks != 0;
  giant_tags = ((size_t)B * g->n_giant + 63) & ~size_t(63);
  if (giant) {                                           // packed path: one accumulator per slice; k_fw: one per 64 links, W-word masks
    const size_t packed = (size_t)B * g->n_giant_slices * GIANT_WORDS * 64 * 4;
    const size_t wide = (size_t)B * g->n_giant_slices * 4 * (6 * 64 * 4 + 2 * (size_t)W * 64 * 8);
    if ((rc = ensure(ctx, ctx->giant_part, giant_tags * 4 + std::max(packed, wide)))) return rc;
  }
  // work counter of the fused kernel (HSPF_RUN_COUNT_ROWS): [256] rows recomputed
  if ((rc = ensure(ctx, ctx->kcnt, 256 * 4))) return rc;
  d_kcnt = (uint32_t *)ctx->kcnt.p;
  count_rows = (run_flags & HSPF_RUN_COUNT_ROWS) != 0;
  return HSPF_OK;
}

// Which kernel of the fused path takes the run; where the results go (the caller's device tables, staging for host
// output, packed words).
int Run::prepare_outputs() {
  // which of the fused path's kernels takes the run (the comments are at their use below)
  const uint32_t smax = std::min(n_roots <= 64 ? ctx->single_max_n * 2u : ctx->single_max_n, SINGLE_MAX_N);
  single = fused && n <= smax && g->e_kept <= SINGLE_MAX_E;
  lv = fused && !single && n_roots <= ctx->lv_max_roots && n >= ctx->lv_min_n && !giant;
  // One to eight roots on a graph of at most XCD_MAX_N vertices: k_xcd (one launch, a barrier inside one XCD per sweep) or
  // the launch-per-sweep engine.  k_xcd's sweeps are Jacobi across workgroups (ospf-10k 37 sweeps of ~4.5 us against ~36
  // launches of ~5.3 us; hop-count graphs and several roots 25-40 % faster; 18 000 vertices one root a tie; a 50 x 50 grid,
  // 111 sweeps against 63-98 launches, 13 % slower with two roots: profiles/r05_notes.md), and on k_single's graphs it is
  // 1.1-4.5 x faster from 300 to 4 000 vertices (with at most eight roots the one-workgroup kernel leaves 248 CUs idle,
  // r05p) — except where k_single_lean applies with ONE root: the reference's own 500-router case, 0.034 against 0.07 ms.
  // Round 5 let the graph remember which of the two was faster last time; the choice is a pure function of the run's shape
  // now (latency that depends on a graph's history is hard to reason about in a daemon).
  xcd_ok = fused && !lv && !ctx->xcd_off && n_roots >= 1 && n_roots <= ctx->xcd_max_roots && n <= XCD_MAX_N && g->n_giant == 0 &&
           !(run_flags & HSPF_RUN_COUNT_ROWS);
  xcdp = xcd_ok;
  if (xcd_ok && !ctx->xcd_always)
    xcdp = !(single && g->lean && n <= (uint32_t)SINGLE_THREADS && !(ctx->variant & 4096u) && n_roots == 1);
  if (xcdp) single = false;
  // row-major output targets (device): the caller's device buffers, or staging for host output
  od = OutDev{};
  rn = (size_t)total_rows * n;               // rows of the output arrays (== n_roots unless mapped)
  d_rank = nullptr;
  // packed results: the fused emit writes the words themselves (od.packed, set per fused_run: the word size is the run's);
  // k_single / k_lv write row-major tables into staging, which k_pack_full then packs (8-byte words)
  pk_full = pk && (single || lv || xcdp);
  d_misfit = nullptr;
  pk_mode = -1;                                        // which fused_run produced the words: 2 lean, 1 narrow, 0 wide; -1: k_pack_full (wide layout)
  if (pk) {
    if (pk->host && !pk->dev_stage && (rc = ensure(ctx, ctx->o_pack, rn * 8, false))) return rc;
    if ((rc = ensure(ctx, ctx->pk_flag, 256, false))) return rc;
    d_misfit = (uint32_t *)ctx->pk_flag.p;
    HIPCHK(ctx, hipMemsetAsync(d_misfit, 0, 4, ctx->stream));
  }
  if (pk) {
    od.out_words = 1;
    if (pk_full) {
      if ((rc = pk_staging())) return rc;
      od.dist = (uint32_t *)ctx->o_dist.p; od.hops = (uint16_t *)ctx->o_hops.p; od.flags = (uint16_t *)ctx->o_flags.p; od.mask = (uint64_t *)ctx->o_mask.p;
    }
  } else if (host_out) {
    if ((rc = ensure(ctx, ctx->o_dist, rn * 4))) return rc;
    if ((rc = ensure(ctx, ctx->o_hops, rn * 2))) return rc;
    if ((rc = ensure(ctx, ctx->o_flags, rn * 2))) return rc;
    if ((rc = ensure(ctx, ctx->o_mask, rn * 8 * out_words))) return rc;
    od.dist = (uint32_t *)ctx->o_dist.p; od.hops = (uint16_t *)ctx->o_hops.p; od.flags = (uint16_t *)ctx->o_flags.p;
    od.mask = (uint64_t *)ctx->o_mask.p; od.out_words = out_words;
    if (run_flags & HSPF_RUN_POP_RANK) { if ((rc = ensure(ctx, ctx->o_rank, rn * 4))) return rc; d_rank = (uint32_t *)ctx->o_rank.p; }
  } else {
    od.dist = out->dist; od.hops = out->hops; od.flags = out->vflags_out; od.mask = out->first_hop_mask; od.out_words = out_words;
    if (run_flags & HSPF_RUN_POP_RANK) d_rank = out->pop_rank;
  }
  od.row_map = nullptr;                                    // set below, once the upload block's address is known

  d_up = (uint32_t *)ctx->up.p;
  d_dist = (uint32_t *)ctx->dist.p; d_hv = (uint32_t *)ctx->hv.p; d_roots = d_up + w_roots;
  d_mask = (uint64_t *)ctx->mask.p;
  d_lf = (uint32_t *)ctx->lane_flags.p;
  d_changed = (int *)ctx->changed.p;
  gd = g->dev();
  tabs = SlotTabs{d_up + w_ptr, d_up + w_vtx, d_up + w_base};
  if (row_map) od.row_map = d_up + w_map;
  ignore_ovl = (run_flags & HSPF_RUN_IGNORE_OVERLOAD) ? 1u : 0u;
  net_nh = (run_flags & HSPF_RUN_NET_NEXTHOPS) ? 1u : 0u;
  return HSPF_OK;
}

// Roots, slot tables, the fused kernel's descriptor: one pinned block, one copy (skipped when the device holds it already).
int Run::upload_block() {
  // ---- upload roots / slot tables / descriptor (one pinned block, one copy), init state
  {
    uint32_t *h = ctx->h_up + (ctx->h_up_sel ? ctx->h_up_cap / 8 : 0);
    const uint32_t *prev = ctx->h_up + (ctx->h_up_sel ? 0 : ctx->h_up_cap / 8);
    std::fill(h, h + up_bytes / 4, 0u);                     // padding words take part in the comparison below
    std::fill(h + w_roots, h + w_roots + L, HSPF_NO_ROOT);
    std::copy(roots, roots + n_roots, h + w_roots);
    std::copy(tab_ptr.begin(), tab_ptr.end(), h + w_ptr);
    std::copy(tab_vtx.begin(), tab_vtx.end(), h + w_vtx);
    std::copy(tab_base.begin(), tab_base.end(), h + w_base);
    for (uint32_t r = 0; r < L; ++r) h[w_map + r] = (row_map && r < n_roots) ? row_map[r] : r;
    const FusedGraph fg{gd, tabs, d_kcnt, giant ? (uint32_t *)ctx->giant_part.p : (uint32_t *)nullptr,
                        (g->n_zero_rows && !g->hopcount_like) ? (uint32_t *)ctx->dyn_part.p : (uint32_t *)nullptr, L};
    memcpy(h + w_fg, &fg, sizeof(FusedGraph));
    // An SPF instance repeats its runs (same graph, same roots): when the block is byte for byte the one the device
    // already holds — same allocation, nothing in it is ever written by a kernel — the copy is skipped (HSPF_VARIANT bit
    // 14 keeps it).  Otherwise the halves swap: the block just built becomes the reference.
    st.dbg[2] = (uint32_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_entry).count();
    const bool same = ctx->up_valid && ctx->up_len == up_bytes && memcmp(h, prev, up_bytes) == 0;
    if (!same) {
      HIPCHK(ctx, hipMemcpyAsync(d_up, h, up_bytes, hipMemcpyHostToDevice, s));
      ctx->up_valid = true; ctx->up_len = up_bytes; ctx->h_up_sel ^= 1;
    }
  }
  return HSPF_OK;
}

// State of the wide-mask path (the fused path initialises per fused_run), grids, the events.
int Run::init_state() {
  d_st = (uint64_t *)ctx->st64.p;
  d_stamp = (uint32_t *)ctx->stamp.p;
  // Wide masks: k_fw unless the graph is hop-count-like with more than 4 mask words or HSPF_VARIANT bit6 asks for the
  // two-phase path (see below).  Leaves (GraphDev::leaf) stay out of k_fw's fixed point and are derived in the emit,
  // when there are any and neither the saturating-distance nor the hop-count instantiation is needed — and the graph has no
  // zero-cost link from a higher-numbered source (n_zero_rows): a leaf behind such a link makes its root's pop order dynamic
  // (the leaf is popped AFTER its parent, against the static (distance, index) order), the emit's derivation gives the right
  // triple but cannot raise LF_DYN any more (the status bits were read before it), and the caller would be told the order
  // is static (round 6: tests/test_gpu_fuzz.py, zero-metric instances, seed 5058).
  use_fw = !fused && !(ctx->variant & 64u) && (!g->hopcount_like || W <= 4);
  defer = use_fw && g->n_leaf != 0 && g->max_path_metric != HSPF_DIST_INF && !g->hopcount_like && g->n_zero_rows == 0;
  HIPCHK(ctx, hipEventRecord(ctx->ev[0], s));
  if (fused) {
    // state / stamps / status bits are initialised by fused_run (it may run twice: narrow, then wide), lv_run or the
    // one-workgroup path
  } else if (defer) {
    // k_fw with the leaves left to the emit: only the rows that take part are filled (distance, hops AND masks)
    HIPCHK(ctx, hipMemsetAsync(d_lf, 0, (size_t)L * 4, s));
    const dim3 ig((unsigned)std::min<size_t>((rows / 64 + 3) / 4, 8192));
    switch (W) {
      case 1: hipLaunchKernelGGL(k_init_fw_state<1>, ig, dim3(256), 0, s, n, B, (const uint8_t *)g->d_leaf, d_dist, d_hv, d_mask); break;
      case 2: hipLaunchKernelGGL(k_init_fw_state<2>, ig, dim3(256), 0, s, n, B, (const uint8_t *)g->d_leaf, d_dist, d_hv, d_mask); break;
      case 4: hipLaunchKernelGGL(k_init_fw_state<4>, ig, dim3(256), 0, s, n, B, (const uint8_t *)g->d_leaf, d_dist, d_hv, d_mask); break;
      case 8: hipLaunchKernelGGL(k_init_fw_state<8>, ig, dim3(256), 0, s, n, B, (const uint8_t *)g->d_leaf, d_dist, d_hv, d_mask); break;
      default: hipLaunchKernelGGL(k_init_fw_state<16>, ig, dim3(256), 0, s, n, B, (const uint8_t *)g->d_leaf, d_dist, d_hv, d_mask); break;
    }
  } else {
    HIPCHK(ctx, hipMemsetAsync(d_lf, 0, (size_t)L * 4, s));
    HIPCHK(ctx, hipMemsetAsync(d_dist, 0xFF, rows * 4, s));
    HIPCHK(ctx, hipMemsetAsync(d_hv, 0, rows * 4, s));
    hipLaunchKernelGGL(k_init_roots, dim3((L + 255) / 256), dim3(256), 0, s, n, d_dist, d_roots, L);
  }

  const uint32_t vblocks = (n + VPB - 1) / VPB;
  (void)vblocks;
  grid = dim3(g->xcd_blocks(), B);                     // 8 x the longest XCD range; shorter ranges leave idle blocks
  const uint32_t fblocks = (n + FVPB - 1) / FVPB;
  (void)fblocks;
  static_assert(FVPB == VPB && VPB == 16, "xcd_start is in 16-vertex chunks for every kernel");
  fgrid = dim3(g->xcd_blocks(), B);
  // An event record is a barrier packet: two of them between kernels cost ~10 us of stream time (kernel trace, r02n).  The
  // paths that do not have a second phase record three events per run (start, end of the sweeps, results in place)
  // instead of six and the missing ones alias their neighbours.
  two_events = !fused;                              // only k_relax + k_dag has a second timed phase
  tail_done = false;        // ev[4] sits behind the emit already and the phase's own synchronisation covered it
  if (!fused) HIPCHK(ctx, hipEventRecord(ctx->ev[1], s));
  return HSPF_OK;
}

  // ---- phase 1: distances.  Launch ahead `est` sweeps (each launch exits at once when the
  // previous one changed nothing), then read ONE flag back; repeat in small chunks if needed.
  // `post` is enqueued behind every chunk of sweeps, BEFORE the host knows whether the chunk reached the fixed point:
  // work that is only needed once (the emit of the fused path) then starts without waiting for the host's round
  // trip; after a chunk that did not converge it is simply enqueued again behind the next one.
  // pre_zeroed: how many leading sweep flags the caller's own init kernel has already cleared (0: cleared here)
  // set by the fused path: enqueue the NEXT run's scratch fill behind this chunk's flag read-back, guarded on the device by
  // "the chunk's last sweep changed nothing" (k_init_fill); run_phase then waits for the read-back only, and the fill
  // (15 us + a launch latency) runs while the host wakes up, returns and prepares the next run.  Argument: index of
  // the chunk's last sweep.
template <class Launch, class Post>
int Run::run_phase(uint32_t est, uint32_t pre_zeroed, Launch &&launch, uint32_t &n_launch, Post &&post) {
    hipError_t er = hipSuccess;
    uint32_t zeroed = std::min<uint32_t>(CHANGED_CAP, est + 4096);
    if (pre_zeroed < zeroed) er = hipMemsetAsync(d_changed + pre_zeroed, 0, (size_t)(zeroed - pre_zeroed) * 4, s);
    if (er != hipSuccess) { ctx->last_error = hipGetErrorString(er); return HSPF_E_HIP; }
    uint32_t sweep = 0, chunk = std::max(2u, est);
    for (;;) {
      if (sweep + chunk > zeroed) {
        if (sweep + chunk > CHANGED_CAP) { ctx->last_error = "phase did not converge within launch cap"; return HSPF_E_INTERNAL; }
        const uint32_t upto = std::min<uint32_t>(CHANGED_CAP, sweep + chunk + 4096);
        er = hipMemsetAsync(d_changed + zeroed, 0, (size_t)(upto - zeroed) * 4, s);
        if (er != hipSuccess) { ctx->last_error = hipGetErrorString(er); return HSPF_E_HIP; }
        zeroed = upto;
      }
      for (uint32_t i = 0; i < chunk; ++i) launch(sweep + i);
      sweep += chunk;
      chunk_last = sweep - 1u;
      post();
      // ONE read-back per chunk: the flags of every sweep launched so far and the per-root status bits
      if (ctx->h_changed_cap < sweep) {
        (void)hipStreamSynchronize(s);
        (void)hipHostFree(ctx->h_changed);
        ctx->h_changed = nullptr; ctx->h_changed_cap = 0;
        const size_t cap = (size_t)sweep * 2;
        if (hipHostMalloc((void **)&ctx->h_changed, cap * sizeof(int), hipHostMallocDefault) != hipSuccess) { ctx->last_error = "pinned flag buffer"; return HSPF_E_NOMEM; }
        ctx->h_changed_cap = cap;
      }
      er = hipMemcpyAsync(ctx->h_changed, d_changed, (size_t)sweep * sizeof(int), hipMemcpyDeviceToHost, s);
      if (er == hipSuccess) er = hipMemcpyAsync(ctx->h_lane_flags, d_lf, (size_t)L * 4, hipMemcpyDeviceToHost, s);
      if (er == hipSuccess && fused && count_rows) er = hipMemcpyAsync(ctx->h_lane_flags + L, d_kcnt, 256 * 4, hipMemcpyDeviceToHost, s);
      if (er == hipSuccess && rb_ctl) er = hipMemcpyAsync(ctx->h_lane_flags + L + 256, rb_ctl, LEAN_CTL_WORDS * 4, hipMemcpyDeviceToHost, s);
      if (er == hipSuccess && spec_fill) {
        er = hipEventRecord(ctx->ev[6], s);
        if (er == hipSuccess) { spec_fill(sweep - 1u); er = hipEventSynchronize(ctx->ev[6]); }
      } else if (er == hipSuccess) er = hipStreamSynchronize(s);
      if (er != hipSuccess) { ctx->last_error = std::string("phase: ") + hipGetErrorString(er); return HSPF_E_HIP; }
      if (ctx->h_changed[sweep - 1] == 0) break;
      if (on_retry) on_retry();
      chunk = 4;
    }
    // count the launches that did work (for stats and the next estimate)
    uint32_t active = 0;
    while (active < sweep && ctx->h_changed[active]) ++active;
    n_launch = active + 1;   // the launch that found the fixed point did a full pass too
    return HSPF_OK;
  }

// One fixed point over the packed state.  mode 0: 8-byte state, 1: 4-byte state (k_fused), 2: 4-byte state, lean sweep (k_fused_lean).
int Run::fused_run(int mode) {
      const bool nar = mode != 0, use_lean = mode == 2;
      const FusedParams P = use_lean ? fp_lean : (nar ? fp_narrow : fp_wide);
      const size_t esz = nar ? 4 : 8;
      OutDev ode = od;                                           // packed results: the emit writes the state words of THIS width
      if (pk) {
        // a DEVICE destination is written in place by the emit: its size is checked BEFORE anything is launched (ADVICE r05:
        // the check used to sit in deliver(), behind the kernels that had already overrun a short buffer)
        if (!pk->host && !pk->dev_stage && (pk->row_off + (size_t)n_roots) * n * esz > pk->cap) {
          ctx->last_error = "packed results: buffer too small, need " + std::to_string((pk->row_off + (size_t)n_roots) * n * esz) + " bytes";
          return HSPF_E_INVAL;
        }
        ode.packed = pk_dev(esz); pk_mode = mode;
      }
      const uint32_t ns = use_lean ? n + 1u : n;                 // rows per batch slab (the lean sweep's pad row)
      const size_t rows = (size_t)B * ns * 64;                   // (state rows of THIS width: shadows the member on purpose)
      const uint32_t fillw = use_lean ? P.infw : 0xFFFFFFFFu;
      // one fill launch (state, stamps, row flags, sweep flags, status bits, row counter), one launch for the roots
      uint32_t pre_zeroed = std::min<uint32_t>(CHANGED_CAP, ctx->est_fused + 4096);
      const bool have = pf.valid && pf.build_id == g->build_id && pf.n == n && pf.B == B && pf.esz == (uint32_t)esz &&
                        pf.ns == ns && pf.fillw == fillw &&
                        pf.L == L && (pf.kcnt || !count_rows) && pf.n_changed >= std::min<uint32_t>(CHANGED_CAP, 2u);
      pf.valid = false;
      if (have) pre_zeroed = pf.n_changed;      // run_phase clears whatever it needs beyond that
      else
        hipLaunchKernelGGL(k_init_fill, dim3(2048), dim3(256), 0, s, (uint4 *)d_st, rows * esz / 16, fillw, d_stamp, (size_t)B * n,
                           (const uint8_t *)g->d_rowflags, (uint8_t *)ctx->hnb.p, n, d_changed, pre_zeroed, d_lf, L,
                           count_rows ? d_kcnt : (uint32_t *)nullptr, -1, (uint32_t *)ctx->swcnt.p, ctx->swcnt.p ? LEAN_CTL_WORDS : 0u);
      last_esz = (uint32_t)esz; last_ns = ns; last_fillw = fillw;
      ctx->prefill.valid = false;                                // (a speculative fill of an earlier fused_run of this call is gone now)
      spec_done = false;
      spec_fill = nullptr;
      spec_fill = [&, esz, fillw, rows](uint32_t last_sweep) {
          const uint32_t nz = std::min<uint32_t>(CHANGED_CAP, ctx->est_fused + 4096);
          // (the emit in front of this launch has reset the state slab under the same guard: stamps, flags and counters are left)
          hipLaunchKernelGGL(k_init_fill, dim3(emit_reset ? 256 : 2048), dim3(256), 0, s, (uint4 *)d_st, emit_reset ? (size_t)0 : rows * esz / 16, fillw, d_stamp, (size_t)B * n,
                             (const uint8_t *)g->d_rowflags, (uint8_t *)ctx->hnb.p, n, d_changed, nz, d_lf, L, d_kcnt, (int)last_sweep, (uint32_t *)ctx->swcnt.p, ctx->swcnt.p ? LEAN_CTL_WORDS : 0u);
          spec_nz = nz;
          spec_done = true;                                      // (valid only if the host finds the chunk converged, see below)
        };
      if (use_lean) st.dbg[0] = 1;                               // hspf_stats::dbg[0]: 1 = the run took the lean sweep
      // k_emit_fused checks the lean state's fields on FINAL words; an emit behind a chunk that had not converged saw
      // transient ones: its LF_OVERFLOW bits are dropped before the next chunk (the final emit tests every word again)
      on_retry = nullptr;
      if (use_lean) on_retry = [&]() { hipLaunchKernelGGL(k_clear_lane_flag, dim3((L + 255) / 256), dim3(256), 0, s, d_lf, L, (uint32_t)LF_OVERFLOW); };
      if (giant && hipMemsetAsync(ctx->giant_part.p, 0, giant_tags * 4, s) != hipSuccess) { ctx->last_error = "giant tags"; return HSPF_E_HIP; }
      const bool dyn_parts = g->n_zero_rows && !g->hopcount_like;      // FusedGraph::dyn_part is in use
      if (dyn_parts && hipMemsetAsync(ctx->dyn_part.p, 0, (size_t)DYN_PARTS * L * 4, s) != hipSuccess) { ctx->last_error = "dyn partials"; return HSPF_E_HIP; }
      if (nar) hipLaunchKernelGGL((k_init_fused<uint32_t>), dim3((L + 3) / 4), dim3(256), 0, s, gd, (uint32_t *)d_st, d_stamp, (uint8_t *)ctx->hnb.p, d_roots, tabs, L, ns);
      else     hipLaunchKernelGGL((k_init_fused<uint64_t>), dim3((L + 3) / 4), dim3(256), 0, s, gd, d_st, d_stamp, (uint8_t *)ctx->hnb.p, d_roots, tabs, L, ns);
      uint32_t n_f = 0;
      const bool maxinf = g->max_path_metric == HSPF_DIST_INF;
      // The lean sweep's plan (k_fused_lean): launch index -> what the launch is.  [0, h) head sweeps (stamped, counting
      // their due rows; skipped on the device once one of them saw the frontier cover `lean_dense_pct` of the rows),
      // [h, h + nd) the dense stretch (launches of up to HSPF_DENSE_PASSES passes, `P` passes in all; a pass is skipped on
      // the device when the passes before it changed less than `lean_stay_pct` of the rows), h + nd the all-due stamped
      // sweep that makes the stamps valid again, then stamped tail sweeps until one changes nothing.  h and P are what
      // the previous run of this context used (+ 1 spare pass): a cost or structural patch, other roots, another graph
      // all start from there, and a fresh context from (4, 16).  HSPF_VARIANT bit19: stamped sweeps only.
      const bool plan_on = use_lean && !(ctx->variant & 524288u);
      // several passes per launch only where a pass — all batches of the call — is long against what the chip holds at
      // once (2 048 workgroups): a short pass would run NEXT TO its successor instead of ahead of it
      const bool multi = ctx->lean_dense_passes > 1u && (uint64_t)fgrid.x * B >= ctx->lean_multi_min_wgs;
      const uint32_t per_launch = multi ? (uint32_t)std::min<uint64_t>(ctx->lean_dense_passes, ((1ull << 31) - 1u) / ((uint64_t)fgrid.x * B)) : 1u;
      const uint32_t plan_h = plan_on ? std::min(ctx->lean_head, 60u) : 0u;
      const uint32_t plan_P = plan_on ? std::min(std::max(ctx->lean_passes, 2u), std::min(ctx->lean_max_passes, 62u)) : 0u;
      const uint32_t plan_nd = plan_on ? (plan_P + per_launch - 1u) / per_launch : 0u;
      const uint64_t all_rows = (uint64_t)n * B;
      // thresholds in sampled units (LEAN_SAMPLE): at least 1, so that "nothing counted" always reads as below
      const uint32_t thr_enter = (uint32_t)std::min<uint64_t>(0xFFFFFFF0u, std::max<uint64_t>(1u, all_rows * ctx->lean_dense_pct / (100u * LEAN_SAMPLE)));
      const uint32_t thr_stay = ctx->lean_stay_pct == 0u ? 0u : (uint32_t)std::min<uint64_t>(0xFFFFFFF0u, std::max<uint64_t>(1u, all_rows * ctx->lean_stay_pct / (100u * LEAN_SAMPLE)));
      if (use_lean) {
        int e2 = ensure(ctx, ctx->swcnt, LEAN_CTL_WORDS * 4, false);      // (the kernel takes the pointer in every mode)
        if (e2) return e2;
      }
      uint32_t *d_ctl = (uint32_t *)ctx->swcnt.p;
      // the run's launch-ahead estimate must reach past the all-due sweep: a chunk that ended inside the plan would pay a
      // read-back (and an emit) for nothing
      // (a context's first fused run starts from est_fused = 12: with the plan in front of them that would be six tail sweeps,
      // and every further chunk costs a read-back and a speculative emit — twelve tail sweeps are launched ahead instead;
      // a surplus launch exits at once)
      if (plan_on) ctx->est_fused = std::max(ctx->est_fused, plan_h + plan_nd + (ctx->est_seen ? 2u : 13u));
      ctx->est_seen = true;
      rb_ctl = plan_on ? d_ctl : nullptr;
      int r2 = run_phase(ctx->est_fused, pre_zeroed, [&](uint32_t sweep) {
#define HSPF_LAUNCH_FUSED2(ST_, MI_, CN_, UN_, stp_) hipLaunchKernelGGL((k_fused<ST_, MI_, CN_, UN_>), fgrid, dim3(256), 0, s, d_fg, d_changed, (int)sweep, d_stamp, (const uint8_t *)ctx->hnb.p, n, gd.in_ptr, gd.out_ptr, gd.vflags, stp_, d_roots, d_lf, net_nh, ignore_ovl, P, gd.in_src, gd.in_w, gd.out_dst, gd.e_in)
#define HSPF_LAUNCH_FUSED(ST_, MI_, CN_, stp_) do { if (units) HSPF_LAUNCH_FUSED2(ST_, MI_, CN_, true, stp_); else HSPF_LAUNCH_FUSED2(ST_, MI_, CN_, false, stp_); } while (0)
        const bool units = g->n_heavy_chunks != 0;
        if (use_lean) {
#define HSPF_LAUNCH_LEAN(CN_, MD_, HD_, grid_, pb_, base_, thr_) hipLaunchKernelGGL((k_fused_lean<CN_, MD_, HD_>), grid_, dim3(256), 0, s, d_fg, d_changed, (int)sweep, d_stamp, (const uint8_t *)ctx->hnb.p, n, (const uint32_t *)g->d_ell_so, (const uint32_t *)g->d_ell_w, (uint32_t *)d_st, (const uint32_t *)g->d_ell_od, d_roots, d_lf, net_nh, ignore_ovl, P, d_ctl, pb_, B, base_, thr_)
#define HSPF_LAUNCH_LEAN_C(MD_, HD_, grid_, pb_, base_, thr_) do { if (count_rows) HSPF_LAUNCH_LEAN(true, MD_, HD_, grid_, pb_, base_, thr_); else HSPF_LAUNCH_LEAN(false, MD_, HD_, grid_, pb_, base_, thr_); } while (0)
          // dense launch in batch-major placement (k_fused_lean<.., BMAJ>; pass_blocks = row blocks of one batch)
#define HSPF_LAUNCH_LEAN_BM(CN_, MD_, HD_, grid_, pb_, base_, thr_) hipLaunchKernelGGL((k_fused_lean<CN_, MD_, HD_, true>), grid_, dim3(256), 0, s, d_fg, d_changed, (int)sweep, d_stamp, (const uint8_t *)ctx->hnb.p, n, (const uint32_t *)g->d_ell_so, (const uint32_t *)g->d_ell_w, (uint32_t *)d_st, (const uint32_t *)g->d_ell_od, d_roots, d_lf, net_nh, ignore_ovl, P, d_ctl, pb_, B, base_, thr_)
#define HSPF_LAUNCH_LEAN_B(MD_, HD_, grid_, pb_, base_, thr_) do { if (count_rows) HSPF_LAUNCH_LEAN_BM(true, MD_, HD_, grid_, pb_, base_, thr_); else HSPF_LAUNCH_LEAN_BM(false, MD_, HD_, grid_, pb_, base_, thr_); } while (0)
          // at least 8 batches: each XCD takes whole batches (their state stays in its L2 across the passes); HSPF_VARIANT bit 24: off (A/B)
          const uint32_t bm_blocks = (n + (uint32_t)FVPB - 1u) / (uint32_t)FVPB;
          const bool bmaj = B >= 8u && 8ull * ((B + 7u) / 8u) * bm_blocks * per_launch < (1ull << 31);
          const dim3 bgrid(8u * ((B + 7u) / 8u) * bm_blocks);        // one pass over all batches, batch-major
          // (the stamped sweeps keep the row-major placement: in batch-major form they were slower — 4.48 against 4.33 ms for the
          // ten areas of configs[3], profiles/r04_notes.md r04y)
          if (sweep < plan_h) HSPF_LAUNCH_LEAN_C(0, true, fgrid, 0u, 0u, thr_enter);
          else if (sweep < plan_h + plan_nd) {
            const uint32_t base = (sweep - plan_h) * per_launch, np = std::min(per_launch, plan_P - base);
            if (bmaj)         HSPF_LAUNCH_LEAN_B(1, false, dim3(bgrid.x * np), bm_blocks, base, thr_stay);
            else if (np > 1u) HSPF_LAUNCH_LEAN_C(1, false, dim3(fgrid.x * B * np), fgrid.x, base, thr_stay);
            else              HSPF_LAUNCH_LEAN_C(1, false, fgrid, 0u, base, thr_stay);
          } else if (plan_on && sweep == plan_h + plan_nd) HSPF_LAUNCH_LEAN_C(2, false, fgrid, 0u, 0u, 0u);
          else HSPF_LAUNCH_LEAN_C(0, false, fgrid, 0u, 0u, 0u);
#undef HSPF_LAUNCH_LEAN_C
#undef HSPF_LAUNCH_LEAN_B
#undef HSPF_LAUNCH_LEAN_BM
#undef HSPF_LAUNCH_LEAN
          return;
        }
        if (giant) {                                   // slices of the due giant rows, ahead of the sweep that merges them
          const dim3 ggrid(g->n_giant_slices, B);
#define HSPF_LAUNCH_GIANT(ST_, MI_, stp_) do { if (P.hc) hipLaunchKernelGGL((k_giant_part<ST_, MI_, true>), ggrid, dim3(256), 0, s, d_fg, (const int *)d_changed, (int)sweep, (const uint32_t *)d_stamp, n, (const ST_ *)stp_, (const uint32_t *)d_roots, net_nh, ignore_ovl, P); \
                                            else hipLaunchKernelGGL((k_giant_part<ST_, MI_, false>), ggrid, dim3(256), 0, s, d_fg, (const int *)d_changed, (int)sweep, (const uint32_t *)d_stamp, n, (const ST_ *)stp_, (const uint32_t *)d_roots, net_nh, ignore_ovl, P); } while (0)
          if (nar)         HSPF_LAUNCH_GIANT(uint32_t, false, (uint32_t *)d_st);
          else if (maxinf) HSPF_LAUNCH_GIANT(uint64_t, true, d_st);
          else             HSPF_LAUNCH_GIANT(uint64_t, false, d_st);
#undef HSPF_LAUNCH_GIANT
        }
        if (count_rows) {
          if (nar)         HSPF_LAUNCH_FUSED(uint32_t, false, true, (uint32_t *)d_st);
          else if (maxinf) HSPF_LAUNCH_FUSED(uint64_t, true, true, d_st);
          else             HSPF_LAUNCH_FUSED(uint64_t, false, true, d_st);
        } else {
          if (nar)         HSPF_LAUNCH_FUSED(uint32_t, false, false, (uint32_t *)d_st);
          else if (maxinf) HSPF_LAUNCH_FUSED(uint64_t, true, false, d_st);
          else             HSPF_LAUNCH_FUSED(uint64_t, false, false, d_st);
        }
#undef HSPF_LAUNCH_FUSED2
#undef HSPF_LAUNCH_FUSED
      }, n_f, [&]() {
        // results out of the packed state (speculative: valid when this chunk reached the fixed point and, for the
        // 4-byte state, no lane overflowed; otherwise redone behind the next chunk / the wide run)
        (void)hipEventRecord(ctx->ev[2], s);
        if (dyn_parts) hipLaunchKernelGGL(k_lf_reduce, dim3((L + 255) / 256), dim3(256), 0, s, d_lf, (const uint32_t *)ctx->dyn_part.p, L);
        // (with the speculative fill on, the emit resets the tiles it has read: first half of the next run's scratch fill)
        const int rg = spec_fill ? (int)chunk_last : -1;
        emit_reset = rg >= 0;
        if (nar) hipLaunchKernelGGL((k_emit_fused<uint32_t>), dim3((n + 63) / 64, B), dim3(256), 0, s, n, n_roots, (uint32_t *)d_st, P, ode, ns, use_lean ? d_lf : (uint32_t *)nullptr, (const int *)d_changed, rg, fillw);
        else     hipLaunchKernelGGL((k_emit_fused<uint64_t>), dim3((n + 63) / 64, B), dim3(256), 0, s, n, n_roots, d_st, P, ode, ns, (uint32_t *)nullptr, (const int *)d_changed, rg, fillw);
        (void)hipEventRecord(ctx->ev[4], s);      // "results in place": the phase's read-back synchronises behind it
        tail_done = true;
      });
      spec_fill = nullptr;
      if (r2) return r2;
      // run_phase returned: its last chunk converged, so the speculative fill behind that chunk's read-back went through
      if (spec_done) ctx->prefill = hspf_ctx::Prefill{true, g->build_id, n, B, (uint32_t)esz, spec_nz, L, true, ns, fillw};
      ctx->est_fused = n_f + 1;
      st.n_relax_launches += n_f;
      if (count_rows) for (uint32_t i = 0; i < 256; ++i) st.rows_recomputed += ctx->h_lane_flags[L + i];
      if (plan_on) {
        // what the plan's launches decided (counters read back with the sweep flags): the next plan
        const uint32_t *hc = ctx->h_lane_flags + L + 256;
        uint32_t trig = plan_h;                                    // first head sweep that was skipped (the sentinel), or none
        for (uint32_t k = 0; k < plan_h; ++k) if (hc[LEAN_CTL_DUE + k * LEAN_CTL_STRIDE] == LEAN_SENTINEL) { trig = k; break; }
        uint32_t used = 0;                                         // dense passes that did work
        for (uint32_t k = 0; k < plan_P; ++k) if (hc[LEAN_CTL_PCH + k * LEAN_CTL_STRIDE] != 0u) used = k + 1u;
        // hspf_stats::dbg[1] (lean sweep): dense passes that did work | head sweeps that ran << 8 | passes planned << 16 | head sweeps planned << 24
        st.dbg[1] = (used & 0xFFu) | ((trig & 0xFFu) << 8) | ((plan_P & 0xFFu) << 16) | ((plan_h & 0x7Fu) << 24);
        // every head sweep ran and the last one still was below the threshold: one more next time (the dense stretch
        // starts a sweep later); some were skipped: exactly as many as ran
        const bool last_triggers = plan_h != 0u && hc[LEAN_CTL_DUE + (plan_h - 1u) * LEAN_CTL_STRIDE] != LEAN_SENTINEL && hc[LEAN_CTL_DUE + (plan_h - 1u) * LEAN_CTL_STRIDE] >= thr_enter;
        if (trig < plan_h) ctx->lean_head = std::max(trig, 1u);
        else if (plan_h != 0u && !last_triggers && used >= 2u) ctx->lean_head = std::min(plan_h + 1u, 60u);
        // the stretch: one spare pass beyond what did work; a stretch that used every pass grows by half
        ctx->lean_passes = used >= plan_P ? std::min(ctx->lean_max_passes, plan_P + plan_P / 2u + 1u) : std::max(used + 1u, 2u);
      }
      return HSPF_OK;
    }

// A few roots on a large graph: lane = vertex (k_lv), one launch per sweep over the root's own row-major state.
int Run::lv_run() {
      int r2;
      if ((r2 = ensure(ctx, ctx->stamp, (size_t)n_roots * n * 4))) return r2;
      uint32_t *a_stamp = (uint32_t *)ctx->stamp.p;
      hipError_t er = hipMemsetAsync(d_lf, 0, (size_t)L * 4, s);
      if (er == hipSuccess) er = hipMemsetAsync(d_st, 0xFF, (size_t)n_roots * n * 8, s);
      if (er == hipSuccess) er = hipMemsetAsync(a_stamp, 0, (size_t)n_roots * n * 4, s);
      if (er == hipSuccess && count_rows) er = hipMemsetAsync(d_kcnt, 0, 256 * 4, s);
      if (er != hipSuccess) { ctx->last_error = std::string("lv init: ") + hipGetErrorString(er); return HSPF_E_HIP; }
      hipLaunchKernelGGL(k_init_lv, dim3((n_roots + 63) / 64), dim3(64), 0, s, gd, d_st, a_stamp, d_roots, n_roots);
      LvArgs la{d_changed, 0, n, a_stamp, d_st, d_roots, gd.in_ptr, gd.rowflags, gd.vflags,
                gd, tabs, d_kcnt, fp_wide, net_nh, ignore_ovl, n_roots, count_rows ? 1u : 0u, d_lf,
                (const uint32_t *)g->d_ell_so, (const uint32_t *)g->d_ell_w, (const uint32_t *)g->d_ell_od,
                2u};
      const bool mi = g->max_path_metric == HSPF_DIST_INF;
      const dim3 lgrid((n + 255) / 256, n_roots);
      uint32_t n_f = 0;
      r2 = run_phase(ctx->est_lv, 0u, [&](uint32_t sweep) {
        la.sweep = (int)sweep;
        if (mi) hipLaunchKernelGGL((k_lv<true>), lgrid, dim3(256), 0, s, la);
        else    hipLaunchKernelGGL((k_lv<false>), lgrid, dim3(256), 0, s, la);
      }, n_f, [&]() {
        (void)hipEventRecord(ctx->ev[2], s);
        hipLaunchKernelGGL(k_emit_lv, lgrid, dim3(256), 0, s, n, (const uint64_t *)d_st, fp_wide, od);
        (void)hipEventRecord(ctx->ev[4], s);
        tail_done = true;
      });
      if (r2) return r2;
      ctx->est_lv = n_f + 1;
      st.n_relax_launches += n_f;
      st.lane_vertex = 1;
      if (count_rows) for (uint32_t i = 0; i < 128; ++i) st.rows_recomputed += ctx->h_lane_flags[L + i];
      return HSPF_OK;
    }

// One to eight roots on a mid-size graph: one XCD per root, the state replicated in every CU's LDS, ONE launch (k_xcd).
std::atomic<uint32_t> g_xcd_next{0};             // the XCD the next run's first root takes: concurrent runs (lanes, contexts) spread out
int Run::xcd_run() {
  int r2;
  const bool fresh = ctx->xcd_ctl.cap == 0;
  if ((r2 = ensure(ctx, ctx->xcd_ctl, (size_t)XCD_MAX_ROOTS * XCD_CTL_WORDS * 4, false))) return r2;
  if (fresh || ctx->xcd_epoch >= (1u << 19) - 2u) {                 // flags compare as (epoch, sweep): zero once, and when the epoch wraps
    HIPCHK(ctx, hipMemsetAsync(ctx->xcd_ctl.p, 0, (size_t)XCD_MAX_ROOTS * XCD_CTL_WORDS * 4, s));
    ctx->xcd_epoch = 0;
  }
  const uint32_t n_pad = (n + 1u) & ~1u;
  const uint32_t vw = std::max(64u, (((n + XCD_MAX_WG - 1u) / XCD_MAX_WG) + 1u) & ~1u);   // vertices per workgroup: even, one per thread
  const uint32_t n_wg = (n + vw - 1u) / vw;
  // one thread per vertex of the range — and enough threads for the re-read of changed ranges to be ONE batch of eight
  // 16-byte loads per thread (n / 16 threads; the kernel is built for at most 640)
  const uint32_t thr = std::min(640u, std::max((vw + 63u) / 64u * 64u, (n / 16u + 63u) / 64u * 64u));
  // every workgroup STORES its last word (status bits of its vertices, sweeps, "gave up") into pinned host memory
  uint32_t *d_hst = nullptr;
  uint32_t *h_st = ctx->h_lane_flags + L;                            // (the 256 words behind the per-root status bits)
  hipError_t er = hipHostGetDevicePointer((void **)&d_hst, h_st, 0);
  if (er != hipSuccess) { ctx->last_error = std::string("k_xcd: pinned status words: ") + hipGetErrorString(er); return HSPF_E_HIP; }
  std::fill(h_st, h_st + XCD_MAX_ROOTS * XCD_MAX_WG, 0u);
  const bool mi = g->max_path_metric == HSPF_DIST_INF;
  static const bool prof = getenv("HSPF_XCD_PROF") != nullptr;      // (tuning only: phase times of one workgroup on stderr)
  void (*kern)(XcdArgs) = prof ? (mi ? k_xcd<true, true> : k_xcd<false, true>) : (mi ? k_xcd<true, false> : k_xcd<false, false>);
  if (!(ctx->xcd_attr & (mi ? 2u : 1u))) {
    // (the replica of the largest graph the path takes; the kernel's own few words of LDS sit next to it)
    const hipError_t ea = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(XCD_MAX_N * 8u));
    if (ea != hipSuccess) { ctx->last_error = std::string("k_xcd: dynamic LDS attribute: ") + hipGetErrorString(ea); return HSPF_E_HIP; }
    ctx->xcd_attr |= mi ? 2u : 1u;
  }
  XcdArgs xa{d_fg, d_roots, fp_wide, net_nh, ignore_ovl, n_roots, ++ctx->xcd_epoch, n_wg, vw,
             g_xcd_next.fetch_add(n_roots, std::memory_order_relaxed) & 7u, n_pad, (uint32_t)(((1ull << 32) + vw - 1u) / vw), d_st, (uint32_t *)ctx->xcd_ctl.p, d_hst,
             (uint64_t)ctx->xcd_timeout_ms * 100000ull, ctx->xcd_skew ? 1u : 0u, od};
  hipLaunchKernelGGL(kern, dim3(8u * n_wg), dim3(thr), (size_t)n * 8, s, xa);
  (void)hipEventRecord(ctx->ev[2], s);
  (void)hipEventRecord(ctx->ev[4], s);            // the kernel wrote the results in place (as k_single)
  tail_done = true;
  const hipError_t el = hipGetLastError();
  er = hipStreamSynchronize(s);
  if (er != hipSuccess || el != hipSuccess) {
    ctx->last_error = std::string("k_xcd: ") + hipGetErrorString(er) + " / launch: " + hipGetErrorString(el) + " threads " + std::to_string(thr) +
                      " workgroups " + std::to_string(n_wg);
    return HSPF_E_HIP;
  }
  bool gave_up = false;
  uint32_t sweeps = 0;
  for (uint32_t r = 0; r < L; ++r) ctx->h_lane_flags[r] = 0u;
  for (uint32_t r = 0; r < n_roots; ++r)
    for (uint32_t w = 0; w < n_wg; ++w) {
      const uint32_t x = h_st[r * XCD_MAX_WG + w];
      if (!(x & XCD_ST_DONE) || (x & XCD_ST_ABORT)) gave_up = true;
      ctx->h_lane_flags[r] |= x & (LF_NEED_EXACT | LF_OVERFLOW | LF_DYN);
      sweeps = std::max(sweeps, (x >> 8) & 0xFFFFu);
      // The root's workgroups did not share an XCD: payload and flags are PLAIN stores, coherent only inside one XCD's L2 —
      // across L2s a flag line can become visible before the state lines it covers, and a run that finished is not proven
      // right.  Redo it on the launch-per-sweep path, like a barrier that gave up.
      if (((x ^ h_st[r * XCD_MAX_WG]) >> 24) & 15u) { st.dbg[1] |= 0x80000000u; gave_up = true; }
    }
  if (gave_up) {
    // a workgroup was not resident, or not where its partners' stores are visible: the launch-per-sweep path redoes the
    // run (nothing of this attempt is used), and this context stops trying
    ctx->xcd_off = true;
    delegated = true;
    return run_impl(ctx, g, roots, n_roots, run_flags, out, host_out, row_map, total_rows, no_fused, pk);
  }
  if (prof && n_roots < XCD_MAX_ROOTS) {
    const uint32_t *t = h_st + (XCD_MAX_ROOTS - 1) * XCD_MAX_WG + 28;
    fprintf(stderr, "[hspf k_xcd] n %u workgroups %u x %u threads, %u sweeps: evaluate %.1f us, publish %.1f, barrier wait %.1f, re-read %.1f\n", n, n_wg, thr,
            sweeps, t[0] / 100.0, t[1] / 100.0, t[2] / 100.0, t[3] / 100.0);
  }
  st.n_relax_launches = 1; st.single_wg = 2;                      // (2: the one-XCD-per-root kernel)
  st.dbg[1] |= sweeps;
  return HSPF_OK;
}

// Small graphs: ONE workgroup per root, the whole state in LDS, one launch (k_single / k_single_lean).
int Run::single_run() {
      // every workgroup STORES its root's status word straight into the pinned host array: no memset, no copy back
      uint32_t *d_hlf = nullptr;
      hipError_t er = hipHostGetDevicePointer((void **)&d_hlf, ctx->h_lane_flags, 0);
      if (er != hipSuccess) { ctx->last_error = std::string("single: pinned status words: ") + hipGetErrorString(er); return HSPF_E_HIP; }
      for (uint32_t r = n_roots; r < L; ++r) ctx->h_lane_flags[r] = 0u;
      if (er == hipSuccess && count_rows) er = hipMemsetAsync(d_kcnt, 0, 256 * 4, s);
      if (er != hipSuccess) { ctx->last_error = std::string("single init: ") + hipGetErrorString(er); return HSPF_E_HIP; }
      // the link records are staged in LDS when they fit next to the state; with many roots only while two workgroups
      // still fit a CU (the staging itself is a pass over the links per root)
      const size_t lds_full = single_lds_bytes(n, g->e_kept, true);
      const bool lds_links = lds_full <= (n_roots > 256 ? SINGLE_LDS_MAX / 2 : SINGLE_LDS_MAX);
      const size_t lds = single_lds_bytes(n, g->e_kept, lds_links);
      const uint32_t thr = std::min<uint32_t>(SINGLE_THREADS, std::max<uint32_t>(64u, (n + 63u) / 64u * 64u));
      const uint32_t need_vpt = (n + thr - 1) / thr;                  // 1 .. 8
      SingleArgs sa{d_fg, d_roots, fp_wide, net_nh, ignore_ovl, n_roots, count_rows ? 1u : 0u, lds_links ? 1u : 0u, d_hlf, od};
      const bool mi = g->max_path_metric == HSPF_DIST_INF;
      void (*kern)(SingleArgs) = nullptr;
#define HSPF_PICK(V_) (lds_links ? (mi ? k_single<true, V_, true> : k_single<false, V_, true>) : (mi ? k_single<true, V_, false> : k_single<false, V_, false>))
      kern = need_vpt <= 1 ? HSPF_PICK(1) : need_vpt <= 2 ? HSPF_PICK(2) : need_vpt <= 4 ? HSPF_PICK(4) : HSPF_PICK(8);
#undef HSPF_PICK
      const int slot = ((need_vpt <= 1 ? 0 : need_vpt <= 2 ? 1 : need_vpt <= 4 ? 2 : 3) * 2 + (mi ? 1 : 0)) * 2 + (lds_links ? 1 : 0);
      if (!(ctx->single_attr & (1u << slot))) {      // more than 64 KB of dynamic LDS has to be allowed per kernel, once
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SINGLE_LDS_MAX);
        ctx->single_attr |= 1u << slot;
      }
      // the plainest graphs (routers only, no row flag, in-degrees <= 8) with one vertex per thread: the lean kernel
      const bool lean = g->lean && n <= (uint32_t)SINGLE_THREADS && !(ctx->variant & 4096u);
      if (lean && g->max_in_deg <= 4u) {
        if (mi) hipLaunchKernelGGL((k_single_lean<true, 4>), dim3(n_roots), dim3(thr), (size_t)n * 8, s, sa);
        else    hipLaunchKernelGGL((k_single_lean<false, 4>), dim3(n_roots), dim3(thr), (size_t)n * 8, s, sa);
      } else if (lean) {
        if (mi) hipLaunchKernelGGL((k_single_lean<true, 8>), dim3(n_roots), dim3(thr), (size_t)n * 8, s, sa);
        else    hipLaunchKernelGGL((k_single_lean<false, 8>), dim3(n_roots), dim3(thr), (size_t)n * 8, s, sa);
      } else
      hipLaunchKernelGGL(kern, dim3(n_roots), dim3(thr), lds, s, sa);
      (void)hipEventRecord(ctx->ev[2], s);
      (void)hipEventRecord(ctx->ev[4], s);          // the kernel wrote the results in place: the run ends here, and the
      tail_done = true;                             // synchronisation below covers it (as behind the emit of the other paths)
      er = hipSuccess;
      if (count_rows) er = hipMemcpyAsync(ctx->h_lane_flags + L, d_kcnt, 256 * 4, hipMemcpyDeviceToHost, s);
      const hipError_t el = hipGetLastError();                         // the launch itself
      if (er == hipSuccess) er = hipStreamSynchronize(s);
      if (er != hipSuccess || el != hipSuccess) {
        ctx->last_error = std::string("k_single: ") + hipGetErrorString(er) + " / launch: " + hipGetErrorString(el) + " threads " +
                          std::to_string(thr) + " lds " + std::to_string(lds);
        return HSPF_E_HIP;
      }
      st.n_relax_launches = 1; st.single_wg = 1;
      if (count_rows) {
        for (uint32_t i = 0; i < 128; ++i) st.rows_recomputed += ctx->h_lane_flags[L + i];
        for (uint32_t i = 0; i < 4; ++i) st.dbg[i] = ctx->h_lane_flags[L + 128 + i];   // sweeps, shader cycles, 100 MHz ticks, set-up cycles of workgroup 0 (this flag only: overwrites the host times)
      }
      narrow = false;
  return HSPF_OK;
}

// Every root has at most 24 first-hop slots: one fused fixed point over a packed state — which kernel, which width.
int Run::path_fused() {
    d_fg = (const FusedGraph *)(d_up + w_fg);
    last_esz = 0;                    // state width of the last fused_run (0: none ran)
    last_ns = n; last_fillw = 0xFFFFFFFFu;
    spec_done = false;                   // the last fused_run's speculative fill was enqueued behind its last chunk
    emit_reset = false;                  // the emit of the current chunk resets the state it has read (k_emit_fused reset_guard)
    spec_nz = 0;
    // Small graphs: one workgroup per root, the whole state in LDS, ONE launch (k_single) instead of a launch per sweep.
    // Measured (profiles/r02e_single_threshold.jsonl, 4-neighbour grids, device time): 500 vertices 2.0-2.8x faster than
    // the sweep engine for 1 / 64 / 1024 roots, 1024 vertices 1.7x / 2.0x / 1.0x, 2048 vertices 1.2x / 1.3x / 0.4x,
    // 4096 vertices 0.4x: a sweep of the one-workgroup kernel is one long dependent chain at two waves per SIMD (~3 700
    // cycles), so it wins only while the sweep engine is bound by its ~50 kernel boundaries.  Hence: up to
    // ctx->single_max_n vertices (HSPF_SINGLE_MAX_N, default 1024; 0 switches the kernel off), twice that for at most
    // one batch of roots.
    // (smax / single: decided above, next to the output targets)
    // A few roots on a larger graph: lane = vertex (k_lv), one launch per sweep over the root's own row-major state.  The
    // lane = root engine spends a 256-byte row per useful 4-8 bytes there (isis-100k, one root: 25 launches x 21 us);
    // the scattered gathers of k_lv cost less than that up to a handful of roots (profiles/r02_notes.md, r02k).
    // (not on graphs with giant rows: a lane of k_lv walks its row alone — one root on isis-100k + a 5 000-router LAN took
    // 6.9 ms there against 0.9 ms for 64 roots on the sweep engine with the row in slices, r02t)
    // (lv: decided above)
    if (single) {
      if ((rc = single_run())) return rc;
    } else if (lv) {
      if ((rc = lv_run())) return rc;
      narrow = false;
    } else if (xcdp) {
      if ((rc = xcd_run())) return rc;
      if (delegated) return HSPF_OK;                             // gave up on a barrier: the launch-per-sweep path has done the run
      narrow = false;
    } else if (lean || narrow) {
      auto overflowed = [&]() { bool o = false; for (uint32_t r = 0; r < L; ++r) o = o || (ctx->h_lane_flags[r] & LF_OVERFLOW); return o; };
      bool done = false;
      if (lean) {
        if ((rc = fused_run(2))) return rc;
        // the lean state's fields are tighter (three tag bits, "not reached" one link cost lower, a saturating hop field):
        // a lane at their edge sends the run to k_fused, and the graph remembers
        if (overflowed()) { g->lean_bad = true; st.dbg[0] = 0; } else done = true;
      }
      if (!done && narrow) {
        if ((rc = fused_run(1))) return rc;
        // did any lane leave the 4-byte fields?  (run_phase has brought the per-root status bits back)
        if (overflowed()) { g->narrow_bad = true; st.narrow_overflow = 1; } else done = true;
      }
      narrow = done;                                             // a 4-byte run holds the results
    }
    if (!single && !lv && !xcdp && !narrow && (rc = fused_run(0))) return rc;
    if (!narrow && fp_wide.hmax < 0xFFFFu) {                       // more than 16 mask bits: did the hop field hold?
      bool ovf = false;
      for (uint32_t r = 0; r < L; ++r) ovf = ovf || (ctx->h_lane_flags[r] & LF_OVERFLOW);
      if (ovf) {
        g->wide24_bad = true;
        delegated = true;                                        // (that call is the run: go() hands its code on)
        return run_impl(ctx, g, roots, n_roots, run_flags, out, host_out, row_map, total_rows, true, pk);
      }
    }
    st.state_bytes = narrow ? 4 : 8;
    if (last_esz && !ctx->prefill.valid) {
      // the next run's scratch, behind this one's emit (same shape assumed: an SPF instance repeats its root set) — unless
      // the speculative fill behind the last chunk's read-back has done it already
      const uint32_t nz = std::min<uint32_t>(CHANGED_CAP, ctx->est_fused + 4096);
      hipLaunchKernelGGL(k_init_fill, dim3(2048), dim3(256), 0, s, (uint4 *)d_st, (size_t)B * last_ns * 64 * last_esz / 16, last_fillw, d_stamp, (size_t)B * n,
                         (const uint8_t *)g->d_rowflags, (uint8_t *)ctx->hnb.p, n, d_changed, nz, d_lf, L, d_kcnt, -1, (uint32_t *)ctx->swcnt.p, ctx->swcnt.p ? LEAN_CTL_WORDS : 0u);
      ctx->prefill = hspf_ctx::Prefill{true, g->build_id, n, B, last_esz, nz, L, true, last_ns, last_fillw};
    }
  return HSPF_OK;
}

int Run::path_wide() {
  // More than 24 first-hop slots: two ways.  k_fw = ONE fused fixed point over (distance, hops, W mask words): half the
  // launches, but a label-correcting sweep re-reads the masks of ALL in-links of a row every time the row is revisited.
  // k_relax + k_dag = distances first (4 bytes per lane and link), then the masks over the tight-link DAG.
  // Measured (profiles/r02g_wide_mask*.jsonl, r02m_*): while a chunk of 100-link rows was four rows per wave the two-phase
  // path won on the fat-tree (4.38 vs 5.04 ms); with heavy chunks cut into one-row-per-wave work units (GraphDev) k_fw
  // wins there too (3.04 vs 3.45 ms) and ties on isis-100k with its roots on a 48-router LAN (2.52 vs 2.55 ms).  So k_fw
  // it is, except for hop-count-like graphs with more than 4 mask words (the plateau rule doubles k_fw's mask
  // registers) and when HSPF_VARIANT bit6 asks for the two-phase path (kept: it is the independent second implementation
  // the "twophase" configuration of the GPU suite runs).
  if (use_fw) {
    d_fg = (const FusedGraph *)(d_up + w_fg);
    if (!defer) HIPCHK(ctx, hipMemsetAsync(d_mask, 0, rows * 8 * W, s));
    if (defer) st.dbg[1] |= 0x80000000u;                         // hspf_stats::dbg[1] bit 31: the leaves were left to the emit
    HIPCHK(ctx, hipMemsetAsync(d_stamp, 0, (size_t)B * n * 4, s));
    hipLaunchKernelGGL(k_init_fw, dim3((L + 3) / 4), dim3(256), 0, s, gd, d_dist, d_stamp, d_roots, L);
    if (giant) HIPCHK(ctx, hipMemsetAsync(ctx->giant_part.p, 0, giant_tags * 4, s));
    const bool mi = g->max_path_metric == HSPF_DIST_INF, hcl = g->hopcount_like;
    const dim3 ggrid(std::max(g->n_giant_slices, 1u), B);
    uint32_t n_fw = 0;
    rc = run_phase(ctx->est_fw, 0u, [&](uint32_t sweep) {
#define HSPF_FWG(W_, MI_, HC_) hipLaunchKernelGGL((k_fw_giant_part<W_, MI_, HC_>), ggrid, dim3(256), 0, s, d_fg, (const uint32_t *)d_dist, (const uint32_t *)d_hv, (const uint64_t *)d_mask, (const uint32_t *)d_stamp, (const uint32_t *)d_roots, net_nh, ignore_ovl, (const int *)d_changed, (int)sweep)
#define HSPF_FW(W_) do { \
      if (giant) { if (hcl) { if (mi) HSPF_FWG((W_ <= 4 ? W_ : 1), true, true); else HSPF_FWG((W_ <= 4 ? W_ : 1), false, true); } \
                   else     { if (mi) HSPF_FWG(W_, true, false); else if (defer) hipLaunchKernelGGL((k_fw_giant_part<W_, false, false, true>), ggrid, dim3(256), 0, s, d_fg, (const uint32_t *)d_dist, (const uint32_t *)d_hv, (const uint64_t *)d_mask, (const uint32_t *)d_stamp, (const uint32_t *)d_roots, net_nh, ignore_ovl, (const int *)d_changed, (int)sweep); else HSPF_FWG(W_, false, false); } } \
      if (hcl) { if (mi) hipLaunchKernelGGL((k_fw<(W_ <= 4 ? W_ : 1), true, true>), grid, dim3(256), 0, s, d_fg, d_dist, d_hv, d_mask, d_stamp, d_roots, g->max_path_metric, net_nh, ignore_ovl, d_changed, (int)sweep, d_lf); \
                 else    hipLaunchKernelGGL((k_fw<(W_ <= 4 ? W_ : 1), false, true>), grid, dim3(256), 0, s, d_fg, d_dist, d_hv, d_mask, d_stamp, d_roots, g->max_path_metric, net_nh, ignore_ovl, d_changed, (int)sweep, d_lf); } \
      else     { if (mi) hipLaunchKernelGGL((k_fw<W_, true, false>), grid, dim3(256), 0, s, d_fg, d_dist, d_hv, d_mask, d_stamp, d_roots, g->max_path_metric, net_nh, ignore_ovl, d_changed, (int)sweep, d_lf); \
                 else if (defer) hipLaunchKernelGGL((k_fw<W_, false, false, true>), grid, dim3(256), 0, s, d_fg, d_dist, d_hv, d_mask, d_stamp, d_roots, g->max_path_metric, net_nh, ignore_ovl, d_changed, (int)sweep, d_lf); \
                 else    hipLaunchKernelGGL((k_fw<W_, false, false>), grid, dim3(256), 0, s, d_fg, d_dist, d_hv, d_mask, d_stamp, d_roots, g->max_path_metric, net_nh, ignore_ovl, d_changed, (int)sweep, d_lf); } } while (0)
      switch (W) {
        case 1: HSPF_FW(1); break;
        case 2: HSPF_FW(2); break;
        case 4: HSPF_FW(4); break;
        case 8: HSPF_FW(8); break;
        default: HSPF_FW(16); break;
      }
#undef HSPF_FW
#undef HSPF_FWG
    }, n_fw, []() {});
    if (rc) return rc;
    ctx->est_fw = n_fw + 1;
    st.n_relax_launches = n_fw;
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], s));
    two_events = false;
  } else {
  uint32_t n_relax = 0;
  rc = run_phase(ctx->est_relax, 0u, [&](uint32_t sweep) {
    if (g->max_path_metric == HSPF_DIST_INF)
      hipLaunchKernelGGL((k_relax<true>), grid, dim3(256), 0, s, gd, d_dist, d_roots, g->max_path_metric, ignore_ovl, d_changed, (int)sweep, d_lf);
    else
      hipLaunchKernelGGL((k_relax<false>), grid, dim3(256), 0, s, gd, d_dist, d_roots, g->max_path_metric, ignore_ovl, d_changed, (int)sweep, d_lf);
  }, n_relax, []() {});
  if (rc) return rc;
  ctx->est_relax = n_relax + 1;
  st.n_relax_launches = n_relax;
  HIPCHK(ctx, hipEventRecord(ctx->ev[2], s));

  // ---- phase 2: hops + first-hop masks over the tight-edge DAG
  HIPCHK(ctx, hipMemsetAsync(d_mask, 0, rows * 8 * W, s));
  HIPCHK(ctx, hipMemsetAsync(d_stamp, 0, (size_t)B * n * 4, s));       // activation stamps: every row due in sweep 0
  uint32_t n_dag = 0;
  uint32_t epoch = 1;
  rc = run_phase(ctx->est_dag, 0u, [&](uint32_t sweep) {
    if (epoch >= HV_EPOCH_MAX) {
      hipLaunchKernelGGL(k_rebase, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d_hv, rows);
      epoch = 2;
    }
    switch (W) {
      case 1: launch_dag<1, true>(grid, s, gd, d_dist, d_hv, d_mask, d_roots, tabs, net_nh, ignore_ovl, d_changed, (int)sweep, epoch, d_lf, g->hopcount_like ? 1u : 0u, d_stamp); break;
      case 2: launch_dag<2>(grid, s, gd, d_dist, d_hv, d_mask, d_roots, tabs, net_nh, ignore_ovl, d_changed, (int)sweep, epoch, d_lf, g->hopcount_like ? 1u : 0u, d_stamp); break;
      case 4: launch_dag<4>(grid, s, gd, d_dist, d_hv, d_mask, d_roots, tabs, net_nh, ignore_ovl, d_changed, (int)sweep, epoch, d_lf, g->hopcount_like ? 1u : 0u, d_stamp); break;
      case 8: launch_dag<8>(grid, s, gd, d_dist, d_hv, d_mask, d_roots, tabs, net_nh, ignore_ovl, d_changed, (int)sweep, epoch, d_lf, g->hopcount_like ? 1u : 0u, d_stamp); break;
      default: launch_dag<16>(grid, s, gd, d_dist, d_hv, d_mask, d_roots, tabs, net_nh, ignore_ovl, d_changed, (int)sweep, epoch, d_lf, g->hopcount_like ? 1u : 0u, d_stamp); break;
    }
    ++epoch;
  }, n_dag, []() {});
  if (rc) return rc;
  ctx->est_dag = n_dag + 1;
  st.n_dag_launches = n_dag;
  HIPCHK(ctx, hipEventRecord(ctx->ev[3], s));

  }
  // ---- emit row-major results
  {
    const dim3 egrid((n + 63) / 64, B, 1 + W);
    const EmitLeaf el{(const FusedGraph *)(d_up + w_fg), d_roots, g->max_path_metric, net_nh, ignore_ovl};
    switch (W) {
      case 1: launch_emit<1>(egrid, s, n, n_roots, d_dist, d_hv, d_mask, od, defer, el); break;
      case 2: launch_emit<2>(egrid, s, n, n_roots, d_dist, d_hv, d_mask, od, defer, el); break;
      case 4: launch_emit<4>(egrid, s, n, n_roots, d_dist, d_hv, d_mask, od, defer, el); break;
      case 8: launch_emit<8>(egrid, s, n, n_roots, d_dist, d_hv, d_mask, od, defer, el); break;
      default: launch_emit<16>(egrid, s, n, n_roots, d_dist, d_hv, d_mask, od, defer, el); break;
    }
  }

  return HSPF_OK;
}

// Roots whose pop order is dynamic (LF_DYN: a vertex whose only way in is a zero-cost link from a higher-numbered source): their
// distances are final; hops and first-hop masks are recomputed in the true pop order on the row-major tables, one workgroup per
// root (spf_repair.hip.h).  Roots it cannot take (a group larger than the walk's list, tables the caller left out) join `ex`.
int Run::repair_roots() {
  if (dy.empty()) return HSPF_OK;
  const auto t0 = std::chrono::steady_clock::now();
  if ((rc = ensure(ctx, ctx->ex_list, (size_t)n_roots * 4, false))) return rc;
  bool unpacked = false;
  std::vector<uint32_t> todo = dy, ok;
  // R settles in a few rounds (the longest chain of zero-cost links a root's order hangs on); a root that needs more is tried
  // again with more rounds — nothing of a failed attempt has touched its rows — before the sequential kernel gets it
  for (uint32_t rounds : {std::max(RP_RELAX, ctx->est_rp_rounds), 8u * std::max(RP_RELAX, ctx->est_rp_rounds), 2048u}) {
    if (todo.empty()) break;
    const uint32_t nd = (uint32_t)todo.size();
    HIPCHK(ctx, hipMemcpyAsync(ctx->ex_list.p, todo.data(), (size_t)nd * 4, hipMemcpyHostToDevice, s));
    if (pk && !pk_full && !unpacked) {                               // packed words straight out of the fused emit: the rows of these roots back into tables
      if ((rc = pk_staging())) return rc;
      od.dist = (uint32_t *)ctx->o_dist.p; od.hops = (uint16_t *)ctx->o_hops.p; od.flags = (uint16_t *)ctx->o_flags.p; od.mask = (uint64_t *)ctx->o_mask.p;
      const FusedParams PP = pk_mode == 2 ? fp_lean : pk_mode == 1 ? fp_narrow : fp_wide;
      const dim3 ug((n + 255u) / 256u, nd);
      if (pk_mode == 2 || pk_mode == 1) hipLaunchKernelGGL((kr_unpack_rows<4>), ug, dim3(256), 0, s, n, nd, (const uint32_t *)ctx->ex_list.p, (const void *)pk_dev(4), PP, od.dist, od.hops, od.flags, od.mask);
      else                              hipLaunchKernelGGL((kr_unpack_rows<8>), ug, dim3(256), 0, s, n, nd, (const uint32_t *)ctx->ex_list.p, (const void *)pk_dev(8), PP, od.dist, od.hops, od.flags, od.mask);
      unpacked = true;
    }
    const size_t zl_off = ((size_t)n + 255u) & ~(size_t)255u;        // [zflag: n bytes | zl: n words | nz]
    const size_t ctl_words = RpCtl::words(nd);
    if ((rc = ensure(ctx, ctx->rp_z, zl_off + ((size_t)n + 1u) * 4, false))) return rc;
    if ((rc = ensure(ctx, ctx->rp_ord, (size_t)nd * n * 8, false))) return rc;
    if ((rc = ensure(ctx, ctx->rp_work, (size_t)nd * n * 12, false))) return rc;
    if ((rc = ensure(ctx, ctx->rp_status, ctl_words * 4, false))) return rc;
    if (ctx->h_rp_cap < ctl_words) {
      if (ctx->h_rp) (void)hipHostFree(ctx->h_rp);
      ctx->h_rp = nullptr; ctx->h_rp_cap = 0;
      HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_rp, (ctl_words + 1024) * 4, hipHostMallocDefault));
      ctx->h_rp_cap = ctl_words + 1024;
    }
    uint8_t *zflag = (uint8_t *)ctx->rp_z.p;
    uint32_t *zl = (uint32_t *)((char *)ctx->rp_z.p + zl_off), *nz = zl + n;
    HIPCHK(ctx, hipMemsetAsync(zflag, 0, n, s));
    HIPCHK(ctx, hipMemsetAsync(nz, 0, 4, s));
    HIPCHK(ctx, hipMemsetAsync(ctx->rp_work.p, 0, (size_t)nd * n * 4, s));                      // the stamps
    HIPCHK(ctx, hipMemsetAsync(ctx->rp_status.p, 0, ctl_words * 4, s));
    if (gd.e_in) hipLaunchKernelGGL(kr_zmark, dim3((gd.e_in + 255u) / 256u), dim3(256), 0, s, gd.e_in, gd.out_dst, gd.out_w, zflag);
    hipLaunchKernelGGL(kr_zcompact, dim3((n + 255u) / 256u), dim3(256), 0, s, n, (const uint8_t *)zflag, zl, nz);
    RepairArgs a{};
    a.g = gd; a.root_list = (const uint32_t *)ctx->ex_list.p; a.roots = d_roots; a.n_dyn = nd; a.net_nexthops = net_nh; a.ignore_ovl = ignore_ovl;
    a.tabs = tabs; a.dist = od.dist; a.hops = od.hops; a.flags = od.flags; a.mask = od.mask; a.words = out_words; a.row_map = od.row_map;
    a.zflag = zflag; a.zl = zl; a.nz = nz;
    a.R = (uint32_t *)ctx->rp_ord.p; a.pos = a.R + (size_t)nd * n;
    a.stamp = (uint32_t *)ctx->rp_work.p; a.wl = a.stamp + (size_t)nd * n;
    a.ctl = RpCtl{(uint32_t *)ctx->rp_status.p, nd};
    a.trace = nullptr;
    if (const char *tv = getenv("HSPF_REPAIR_TRACE")) {          // debugging: root:v0,v1,v2,v3 (see spf_repair.hip.h)
      unsigned tr = 0, t0 = ~0u, t1 = ~0u, t2 = ~0u, t3 = ~0u;
      if (sscanf(tv, "%u:%u,%u,%u,%u", &tr, &t0, &t1, &t2, &t3) >= 2 && ensure(ctx, ctx->rp_trace, (4u + 4u * RP_TRACE_CAP) * 4, false) == HSPF_OK) {
        a.trace = (uint32_t *)ctx->rp_trace.p; a.trace_root = tr; a.trace_v[0] = t0; a.trace_v[1] = t1; a.trace_v[2] = t2; a.trace_v[3] = t3;
        HIPCHK(ctx, hipMemsetAsync(a.trace, 0, 16, s));
      }
    }
    // a root's share of the chip.  Eight roots and more: RP_GX blocks each, a root's blocks all on ONE XCD (RpRoot: its tables meet
    // one L2; repair of 64 roots on isis-100k 1.00 -> 0.92 ms at 1 % zero-cost links, 7.3 -> 5.5 at 5 %, ospf-10k 0.49 -> 0.41 and
    // 1.13 -> 0.86; with 256 blocks per root — one root at a time per XCD — the launches' own cost ate the gain: 1.56).  Fewer roots:
    // spread over the whole chip, 256 blocks each (one root: 0.36 -> 0.29 ms at 1 %, 1.59 -> 1.05 at 5 %).  profiles/r06_notes.md r06zd
    a.xcd_map = nd >= 8u ? 1u : 0u;
    const uint32_t gx = a.xcd_map ? RP_GX : 256u;
    const dim3 pg(gx, a.xcd_map ? ((nd + 7u) & ~7u) : nd), tb(256);
    hipLaunchKernelGGL(kr_seed, pg, tb, 0, s, a);
    for (uint32_t r = 0; r < rounds; ++r) hipLaunchKernelGGL(kr_relax, pg, tb, 0, s, a, r);
    hipLaunchKernelGGL(kr_walks, pg, tb, 0, s, a, rounds);
    hipLaunchKernelGGL(kr_walks_deep, pg, tb, 0, s, a);
    hipLaunchKernelGGL(kr_due, pg, tb, 0, s, a);
    const RpCtl hc{ctx->h_rp, nd};
    uint32_t s_next = 1u, chunk = std::max(2u, ctx->est_rp);
    bool overrun = false;
    for (;;) {
      for (uint32_t i = 0; i < chunk; ++i) {
        const uint32_t sw = s_next + i;
        if (out_words <= 1)      hipLaunchKernelGGL((kr_sweep<1>), pg, tb, 0, s, a, sw);
        else if (out_words <= 2) hipLaunchKernelGGL((kr_sweep<2>), pg, tb, 0, s, a, sw);
        else if (out_words <= 4) hipLaunchKernelGGL((kr_sweep<4>), pg, tb, 0, s, a, sw);
        else                     hipLaunchKernelGGL((kr_sweep<16>), pg, tb, 0, s, a, sw);
      }
      s_next += chunk;
      HIPCHK(ctx, hipMemcpyAsync(ctx->h_rp, ctx->rp_status.p, ctl_words * 4, hipMemcpyDeviceToHost, s));
      HIPCHK(ctx, hipStreamSynchronize(s));
      if (hc.pend()[s_next] == 0u) break;                             // the last sweep launched woke nobody
      if (s_next + 8u > RP_MAX_SWEEPS) { overrun = true; break; }
      chunk = 8u;
    }
    if ((run_flags & HSPF_RUN_POP_RANK) && d_rank && !overrun) {
      // pop ranks of the attempt's roots (a root that failed gets them from a later attempt or from k_exact)
      const size_t total = (size_t)nd * n;
      unsigned jb = 1; while ((1ull << jb) < nd) ++jb;
      size_t t1 = 0, t2 = 0;
      HIPCHK(ctx, (hipError_t)hub_sort_pairs(nullptr, &t1, nullptr, nullptr, nullptr, nullptr, total, 32, s));
      HIPCHK(ctx, (hipError_t)hub_sort_pairs(nullptr, &t2, nullptr, nullptr, nullptr, nullptr, total, 32 + jb, s));
      const size_t tb = (std::max(t1, t2) + 255) & ~(size_t)255;
      if ((rc = ensure(ctx, ctx->rp_rank, total * 24 + tb + 1024, false))) return rc;
      uint64_t *k_a = (uint64_t *)ctx->rp_rank.p, *k_b = k_a + total;
      uint32_t *v_a = (uint32_t *)(k_b + total), *v_b = v_a + total;
      void *tmp = (void *)(((uintptr_t)(v_b + total) + 255) & ~(uintptr_t)255);
      size_t tbb = tb;
      hipLaunchKernelGGL(kr_rank_keys1, dim3((n + 255u) / 256u, nd), dim3(256), 0, s, a, k_a, v_a);
      HIPCHK(ctx, (hipError_t)hub_sort_pairs(tmp, &tbb, k_a, k_b, v_a, v_b, total, 32, s));
      hipLaunchKernelGGL(kr_rank_keys2, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, a, (const uint32_t *)v_b, k_a, total);
      tbb = tb;
      HIPCHK(ctx, (hipError_t)hub_sort_pairs(tmp, &tbb, k_a, k_b, v_b, v_a, total, 32 + jb, s));
      hipLaunchKernelGGL(kr_rank_write, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, a, (const uint32_t *)v_a, d_rank, total);
    }
    uint32_t used = 0;
    while (used + 1u <= RP_MAX_SWEEPS && hc.pend()[used + 1u]) ++used;
    ctx->est_rp = std::min(used + 2u, 64u);
    { uint32_t need = 0; for (uint32_t j = 0; j < nd; ++j) if (hc.fail()[j] != 1u) need = std::max(need, hc.rlast()[j] + 2u);
      if (rounds <= 64u) ctx->est_rp_rounds = std::min(std::max(need, RP_RELAX), 64u); }       // rounds the next repair launches (the last one's + a spare)
    uint32_t t_evals = 0, t_groups = 0, t_gmax = 0;
    for (uint32_t j = 0; j < nd; ++j) { t_evals += hc.tot(j)[0]; t_groups += hc.tot(j)[1]; t_gmax = std::max(t_gmax, hc.tot(j)[2]); }
    if (a.trace) {
      std::vector<uint32_t> tr(4u + 4u * RP_TRACE_CAP);
      HIPCHK(ctx, hipMemcpy(tr.data(), a.trace, tr.size() * 4, hipMemcpyDeviceToHost));
      const uint32_t cntt = std::min(tr[0], RP_TRACE_CAP);
      fprintf(stderr, "[hspf repair trace] %u records (sweeps %u)\n", tr[0], 0u);
      for (uint32_t i = 0; i < cntt; ++i) {
        const uint32_t *t = tr.data() + 4 + 4 * i;
        if ((t[0] >> 28) == 1u) fprintf(stderr, "  wake   for sweep %u: vertex %u by %u -> %s\n", t[0] & 0xFFFFFFu, t[1], t[2], t[3] ? "appended" : "already stamped");
        else fprintf(stderr, "  eval   in sweep %u: vertex %u first parent's hops %u -> hops %u (was %u), changed %u, R of first parent %u\n", t[0] & 0xFFFFFFu, t[1], t[2] >> 16, t[2] & 0xFFFFu, t[3] >> 16, t[3] & 1u, (t[3] & 0xFFFEu) >> 1);
      }
    }
    if (getenv("HSPF_REPAIR_PROF"))
      fprintf(stderr, "[hspf repair] %u roots, n %u, %u rounds: %u sweeps, %u evaluations, %u groups (largest deep one %u); host %.1f us\n", nd, n, rounds, used, t_evals, t_groups,
              t_gmax, std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count());
    std::vector<uint32_t> again;
    for (uint32_t j = 0; j < nd; ++j) {
      if (overrun || hc.fail()[j] >= 2u) ex.push_back(todo[j]);
      else if (hc.fail()[j] == 1u) again.push_back(todo[j]);           // R did not settle: more rounds
      else ok.push_back(todo[j]);
    }
    st.repair_sweeps = std::max(st.repair_sweeps, used); st.repair_evals += t_evals; st.repair_groups += t_groups;
    todo.swap(again);
  }
  ex.insert(ex.end(), todo.begin(), todo.end());
  std::sort(ok.begin(), ok.end());
  dy.swap(ok);
  std::sort(ex.begin(), ex.end());
  st.n_repaired_roots = (uint32_t)dy.size();
  st.ms_repair = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return HSPF_OK;
}

// Roots that need the literal sequential loop (u32 saturation, HSPF_RUN_FORCE_EXACT / HSPF_RUN_POP_RANK, what k_repair handed
// back): k_exact (status bits: last run_phase).
int Run::exact_roots() {
  ex.clear(); dy.clear(); oob.clear();
  const bool forced = (run_flags & HSPF_RUN_FORCE_EXACT) != 0;
  const bool tables = pk || (od.hops && od.flags && od.mask);      // everything k_repair reads and writes is there
  // HSPF_RUN_POP_RANK: every root goes through the repair (a root with a static order has nothing to put right there: its
  // ranks are the order of (distance, index)) and the ranks come out of two sorts; until round 6 the flag forced k_exact
  const bool want_rank = (run_flags & HSPF_RUN_POP_RANK) && d_rank && tables && !(ctx->variant & (1u << 27));
  for (uint32_t r = 0; r < n_roots; ++r) {
    if (roots[r] == HSPF_NO_ROOT) continue;
    const uint32_t lf = ctx->h_lane_flags[r];
    if (forced || (lf & LF_NEED_EXACT) || ((run_flags & HSPF_RUN_POP_RANK) && !want_rank)) ex.push_back(r);
    else if (want_rank) dy.push_back(r);
    else if (lf & LF_DYN) {
      if (!pk && !od.hops && !od.mask) continue;                     // distances only: they are final as they are
      if (tables && !(ctx->variant & (1u << 27))) dy.push_back(r); else ex.push_back(r);      // (HSPF_VARIANT bit 27: k_exact as before round 6 — tests, A/B)
    }
  }
  if ((rc = repair_roots())) return rc;
  oob = dy; oob.insert(oob.end(), ex.begin(), ex.end());
  st.n_exact_roots = (uint32_t)ex.size();
  if (!oob.empty()) {
    if ((rc = ensure(ctx, ctx->ex_list, (size_t)n_roots * 4, false))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ex_list.p, oob.data(), oob.size() * 4, hipMemcpyHostToDevice, s));
  }
  if (!ex.empty()) {
    // k_exact needs all four result arrays; use staging for the ones the caller skipped.
    ExactArgs a{};
    a.g = gd; a.roots = d_roots; a.n_exact = (uint32_t)ex.size();
    a.maxpath = g->max_path_metric; a.net_nexthops = net_nh; a.ignore_ovl = ignore_ovl; a.tabs = tabs;
    if (pk && !pk_full) {                                          // packed run on the fused emit: the sequential kernel's rows go through staging
      if ((rc = pk_staging())) return rc;
      od.dist = (uint32_t *)ctx->o_dist.p; od.hops = (uint16_t *)ctx->o_hops.p; od.flags = (uint16_t *)ctx->o_flags.p; od.mask = (uint64_t *)ctx->o_mask.p;
    }
    a.dist = od.dist; a.hops = od.hops; a.flags = od.flags; a.mask = od.mask; a.words = out_words; a.pop_rank = d_rank;
    a.row_map = od.row_map;
    if (!a.hops) { if ((rc = ensure(ctx, ctx->o_hops, rn * 2))) return rc; a.hops = (uint16_t *)ctx->o_hops.p; }
    if (!a.flags) { if ((rc = ensure(ctx, ctx->o_flags, rn * 2))) return rc; a.flags = (uint16_t *)ctx->o_flags.p; }
    if (!a.mask) { if ((rc = ensure(ctx, ctx->o_mask, rn * 8 * out_words))) return rc; a.mask = (uint64_t *)ctx->o_mask.p; }
    if ((rc = ensure(ctx, ctx->ex_heap, ex.size() * (size_t)n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->ex_pos, ex.size() * (size_t)n * 4))) return rc;
    a.root_list = (const uint32_t *)ctx->ex_list.p + dy.size();
    a.heap = (uint32_t *)ctx->ex_heap.p; a.pos = (uint32_t *)ctx->ex_pos.p;
    hipLaunchKernelGGL(k_exact, dim3((a.n_exact + 63) / 64), dim3(64), 0, s, a);
  }
  if ((run_flags & HSPF_RUN_POP_RANK) && d_rank) {
    // pop_rank of padding roots
    for (uint32_t r = 0; r < n_roots; ++r)
      if (roots[r] == HSPF_NO_ROOT) HIPCHK(ctx, hipMemsetAsync(d_rank + (size_t)(row_map ? row_map[r] : r) * n, 0xFF, (size_t)n * 4, s));
  }
  return HSPF_OK;
}

// Packed results: rows that did not come out of the fused emit, the layout; then whether the run is over already.
int Run::packed_finish() {
  // Nothing was enqueued behind the emit (no sequential roots, no padding ranks, device-resident results): the run is
  // complete and synchronised already — a second event record + stream synchronisation cost 12 us per run.
  // ---- packed results: rows that did not come out of the fused emit, the layout, the copy
  pk_esz = 0;
  if (pk) {
    const FusedParams PP = pk_mode == 2 ? fp_lean : pk_mode == 1 ? fp_narrow : fp_wide;
    pk_esz = (pk_mode == 2 || pk_mode == 1) ? 4 : 8;
    if ((pk_full || !oob.empty()) && !pk->host && !pk->dev_stage && (pk->row_off + (size_t)n_roots) * n * pk_esz > pk->cap) {
      ctx->last_error = "packed results: buffer too small, need " + std::to_string((pk->row_off + (size_t)n_roots) * n * pk_esz) + " bytes";
      return HSPF_E_INVAL;
    }
    if (pk_full || !oob.empty()) {
      const uint32_t nrows = pk_full ? n_roots : (uint32_t)oob.size();
      const uint32_t *rows_list = pk_full ? (const uint32_t *)nullptr : (const uint32_t *)ctx->ex_list.p;
      const dim3 pg((n + 255u) / 256u, nrows);
      if (pk_esz == 4) hipLaunchKernelGGL((k_pack_full<4>), pg, dim3(256), 0, s, n, nrows, rows_list, (const uint32_t *)od.dist, (const uint16_t *)od.hops, (const uint16_t *)od.flags, (const uint64_t *)od.mask, 1u, PP, (void *)pk_dev(4), d_misfit);
      else             hipLaunchKernelGGL((k_pack_full<8>), pg, dim3(256), 0, s, n, nrows, rows_list, (const uint32_t *)od.dist, (const uint16_t *)od.hops, (const uint16_t *)od.flags, (const uint64_t *)od.mask, 1u, PP, (void *)pk_dev(8), d_misfit);
      HIPCHK(ctx, hipMemcpyAsync(ctx->h_lane_flags + L + 256 + LEAN_CTL_WORDS, d_misfit, 4, hipMemcpyDeviceToHost, s));
    }
    hspf_packed_layout &ly = pk->layout;
    ly = hspf_packed_layout{};
    ly.word_bytes = (uint32_t)pk_esz;
    ly.dist_shift = pk_esz == 4 ? PP.sh : 32u; ly.hops_shift = PP.mbits; ly.hops_mask = PP.hmax; ly.mask_bits = PP.mbits;
    ly.not_reached = pk_esz == 4 ? (uint64_t)PP.inf_t : ~0ull;
    if (pk->root_status) {
      for (uint32_t r = 0; r < n_roots; ++r) pk->root_status[pk->row_off + r] = 0;
      for (uint32_t r : oob) pk->root_status[pk->row_off + r] = HSPF_ROOT_EXACT;
    }
  }
  finished = tail_done && oob.empty() && !host_out && !pk_full && !((run_flags & HSPF_RUN_POP_RANK) && d_rank);
  if (!finished) HIPCHK(ctx, hipEventRecord(ctx->ev[4], s));
  return HSPF_OK;
}

// Results to the caller (copies for host destinations), the last synchronisation, the statistics.
int Run::deliver() {
  // ---- results to the caller
  if (pk) {
    if (pk->host) {
      const size_t bytes = (size_t)n_roots * n * pk_esz, off = pk->row_off * (size_t)n * pk_esz;
      if (off + bytes > pk->cap) { (void)hipStreamSynchronize(s); ctx->last_error = "packed results: buffer too small, need " + std::to_string(off + bytes) + " bytes"; return HSPF_E_INVAL; }
      if (pk->dev_stage) pk->copy_bytes = bytes;                   // (the lane copies, on the copy stream, once this run is over)
      else if ((rc = copy_to_host(ctx, (char *)pk->dst + off, pk_dev(pk_esz), bytes, s))) return rc;
    } else if (pk->row_off * (size_t)n * pk_esz + (size_t)n_roots * n * pk_esz > pk->cap) {
      (void)hipStreamSynchronize(s); ctx->last_error = "packed results: buffer too small"; return HSPF_E_INVAL;
    }
  } else if (host_out) {
    HIPCHK(ctx, hipMemcpyAsync(out->dist, od.dist, rn * 4, hipMemcpyDeviceToHost, s));
    if (out->hops) HIPCHK(ctx, hipMemcpyAsync(out->hops, od.hops, rn * 2, hipMemcpyDeviceToHost, s));
    if (out->vflags_out) HIPCHK(ctx, hipMemcpyAsync(out->vflags_out, od.flags, rn * 2, hipMemcpyDeviceToHost, s));
    if (out->first_hop_mask) HIPCHK(ctx, hipMemcpyAsync(out->first_hop_mask, od.mask, rn * 8 * out_words, hipMemcpyDeviceToHost, s));
    if ((run_flags & HSPF_RUN_POP_RANK) && out->pop_rank) HIPCHK(ctx, hipMemcpyAsync(out->pop_rank, d_rank, rn * 4, hipMemcpyDeviceToHost, s));
  }
  if (host_out) HIPCHK(ctx, hipEventRecord(ctx->ev[5], s));
  if (!finished) HIPCHK(ctx, hipStreamSynchronize(s));
  if (pk && (pk_full || !oob.empty()) && ctx->h_lane_flags[L + 256 + LEAN_CTL_WORDS] != 0u) {
    ctx->last_error = "packed results: a value of a row from the one-workgroup / lane = vertex / sequential kernel does not fit the run's fields";
    return HSPF_E_NO_PACKED;
  }
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { ctx->last_error = std::string("kernel launch: ") + hipGetErrorString(le); return HSPF_E_HIP; }
  }
  {
    hipEvent_t e1 = fused ? ctx->ev[0] : ctx->ev[1], e3 = two_events ? ctx->ev[3] : ctx->ev[2];
    (void)hipEventElapsedTime(&st.ms_total, ctx->ev[0], ctx->ev[4]);
    (void)hipEventElapsedTime(&st.ms_relax, e1, ctx->ev[2]);
    st.ms_dag = 0.f;
    if (two_events) (void)hipEventElapsedTime(&st.ms_dag, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&st.ms_finish, e3, ctx->ev[4]);
    st.ms_d2h = 0.f;
    if (host_out) (void)hipEventElapsedTime(&st.ms_d2h, ctx->ev[4], ctx->ev[5]);
  }
  if (!(count_rows && st.single_wg))
    st.dbg[3] = (uint32_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_entry).count();
  return HSPF_OK;
}
}  // namespace

extern "C" {

static int run_impl(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                    hspf_result *out, bool host_out, const uint32_t *row_map, uint32_t total_rows,
                    bool no_fused, PackedReq *pk) {
  // packed results (hspf_run_packed*): `out` is not read; host_out says whether pk->dst is host memory
  hspf_result pk_none{};
  if (pk) { out = &pk_none; host_out = pk->host; }
  if (!ctx || !g || !roots || !out || n_roots == 0 || (!pk && !out->dist) || (row_map && host_out) || (pk && (row_map || !pk->dst))) return HSPF_E_INVAL;
  if (pk && (run_flags & HSPF_RUN_POP_RANK)) { ctx->last_error = "HSPF_RUN_POP_RANK with packed results"; return HSPF_E_INVAL; }
  if (g && g->invalid) { ctx->last_error = "the graph is invalid after a failed hspf_graph_patch (free it and upload again)"; return HSPF_E_INVAL; }
  if (pk && no_fused) { ctx->last_error = "packed results: hop counts beyond the hop field of a run with more than 16 first-hop slots"; return HSPF_E_NO_PACKED; }
  if (!row_map) total_rows = n_roots;
  (void)hipSetDevice(ctx->device);
  const uint32_t n = g->n;
  // Bound the scratch (state, stamps, staging) of one pass: the batch axis is processed in groups of
  // at most ~2^26 (vertex, root) pairs (>= 1 batch of 64 roots), each group a complete run of its own,
  // so that "every router as a root" on a large LSDB does not need state for all roots at once.
  {
    const uint64_t max_pairs = 1ull << 26;
    uint32_t group = (uint32_t)std::max<uint64_t>(64, (max_pairs / std::max<uint32_t>(n, 1)) / 64 * 64);
    if (n_roots > group && pk) {
      // every group must come in ONE layout: the field split is made from the slots of ALL roots (min_slots), and when a
      // group still comes back with another layout than the first (a 4-byte overflow that only its roots run into, a ragged
      // last group on the lane = vertex path) the call starts over with 8-byte words for everybody
      uint32_t all_slots = pk->min_slots;
      {
        std::vector<uint32_t> hv, hb;
        for (uint32_t r = 0; r < n_roots; ++r) {
          if (roots[r] == HSPF_NO_ROOT) continue;
          if (roots[r] >= n) { ctx->last_error = "root out of range"; return HSPF_E_INVAL; }
          uint32_t total = 0;
          build_slot_table(g, roots[r], hv, hb, total, ctx->mark, next_mark(ctx, n));
          all_slots = std::max(all_slots, total);
        }
      }
      for (int attempt = 0; attempt < 2; ++attempt) {
        hspf_stats acc{};
        bool again = false;
        hspf_packed_layout first{};
        for (uint32_t off = 0; off < n_roots && !again; off += group) {
          const uint32_t nr = std::min(group, n_roots - off);
          PackedReq part = *pk;
          part.row_off = pk->row_off + off; part.min_slots = all_slots; part.force_wide = pk->force_wide || attempt == 1;
          const int rc = run_impl(ctx, g, roots + off, nr, run_flags, nullptr, host_out, nullptr, 0, no_fused, &part);
          if (rc) return rc;
          if (off == 0) first = part.layout;
          else if (memcmp(&first, &part.layout, sizeof(first)) != 0) { again = true; break; }
          const hspf_stats &p = ctx->stats;
          acc.n_roots += p.n_roots; acc.n_batches += p.n_batches; acc.n_relax_launches += p.n_relax_launches;
          acc.n_exact_roots += p.n_exact_roots; acc.n_repaired_roots += p.n_repaired_roots; acc.repair_sweeps = std::max(acc.repair_sweeps, p.repair_sweeps); acc.repair_evals += p.repair_evals; acc.repair_groups += p.repair_groups; acc.ms_repair += p.ms_repair; acc.n_mask_words = std::max(acc.n_mask_words, p.n_mask_words);
          acc.ms_total += p.ms_total; acc.ms_relax += p.ms_relax; acc.ms_finish += p.ms_finish; acc.ms_d2h += p.ms_d2h;
          acc.state_bytes = std::max(acc.state_bytes, p.state_bytes); acc.narrow_overflow += p.narrow_overflow; acc.rows_recomputed += p.rows_recomputed;
        }
        if (!again) { pk->layout = first; ctx->stats = acc; return HSPF_OK; }
      }
      ctx->last_error = "packed results: the groups of the call did not agree on a layout";
      return HSPF_E_INTERNAL;
    }
    if (n_roots > group) {
      hspf_stats acc{};
      if (out->first_hop_mask && out->n_mask_words == 0) return HSPF_E_INVAL;
      for (uint32_t off = 0; off < n_roots; off += group) {
        const uint32_t nr = std::min(group, n_roots - off);
        hspf_result part = *out;
        const size_t o = row_map ? 0 : (size_t)off * n;          // mapped rows are addressed through the map
        part.dist = out->dist + o;
        if (out->hops) part.hops = out->hops + o;
        if (out->vflags_out) part.vflags_out = out->vflags_out + o;
        if (out->first_hop_mask) part.first_hop_mask = out->first_hop_mask + o * out->n_mask_words;
        if (out->pop_rank) part.pop_rank = out->pop_rank + o;
        const int rc = run_impl(ctx, g, roots + off, nr, run_flags, &part, host_out, row_map ? row_map + off : nullptr, total_rows, no_fused);
        if (rc) return rc;
        const hspf_stats &p = ctx->stats;
        acc.n_roots += p.n_roots; acc.n_batches += p.n_batches; acc.n_relax_launches += p.n_relax_launches;
        acc.n_dag_launches += p.n_dag_launches; acc.n_exact_roots += p.n_exact_roots; acc.n_repaired_roots += p.n_repaired_roots; acc.repair_sweeps = std::max(acc.repair_sweeps, p.repair_sweeps); acc.repair_evals += p.repair_evals; acc.repair_groups += p.repair_groups; acc.ms_repair += p.ms_repair;
        acc.n_mask_words = std::max(acc.n_mask_words, p.n_mask_words);
        acc.ms_total += p.ms_total; acc.ms_relax += p.ms_relax; acc.ms_dag += p.ms_dag; acc.ms_finish += p.ms_finish;
        acc.ms_d2h += p.ms_d2h; acc.state_bytes = std::max(acc.state_bytes, p.state_bytes);
        acc.narrow_overflow += p.narrow_overflow; acc.rows_recomputed += p.rows_recomputed;
      }
      ctx->stats = acc;
      return HSPF_OK;
    }
  }
  Run run(ctx, g, roots, n_roots, run_flags, out, host_out, row_map, total_rows, no_fused, pk);
  return run.go();
}


// A run takes the state its most demanding root needs: one root with more than 24 first-hop slots sends every root of
// the call down the two-phase path, one with 15-24 slots makes the packed state 8 bytes wide for all.  With many roots
// ("every router of the area") that is the common case, so the roots are regrouped by what
// they need — narrow fused / wide fused / two-phase —, each class runs on its own and writes its rows straight to
// their places in the caller's order (row map).  A class is only split off when that saves work: an extra run costs a whole
// fixed point.
static int run_classes(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                       hspf_result *out, bool host_out) {
  if (!ctx || !g || !roots || !out || n_roots <= 64 || !out->dist || (run_flags & (HSPF_RUN_FORCE_EXACT | HSPF_RUN_POP_RANK)))
    return run_impl(ctx, g, roots, n_roots, run_flags, out, host_out);
  const uint32_t n = g->n;
  // widest mask field the 4-byte state can take on this graph (same rule as run_impl)
  uint32_t m_narrow = 0;
  if (!g->narrow_bad && !(ctx->variant & 2u))
    for (uint32_t M = 16; M >= 1; --M) {
      const uint32_t H = (32u - M - 7u >= 13u) ? 7u : 6u, D = 32u - M - H;
      if (D >= 12u && g->wmax < (1u << (D - 3))) { m_narrow = M; break; }
    }
  std::vector<uint8_t> cls(n_roots, 0);
  uint32_t cnt[3] = {0, 0, 0};
  {
    std::vector<uint32_t> hv, hb;
    for (uint32_t r = 0; r < n_roots; ++r) {
      uint32_t total = 0;
      if (roots[r] != HSPF_NO_ROOT) {
        if (roots[r] >= n) { ctx->last_error = "root out of range"; return HSPF_E_INVAL; }
        build_slot_table(g, roots[r], hv, hb, total, ctx->mark, next_mark(ctx, n));
      }
      cls[r] = total > (g->wide24_bad ? 16u : 24u) ? 2 : (total > m_narrow ? 1 : 0);
      cnt[cls[r]]++;
    }
  }
  // LEAF ROOTS: a root whose only kept link leads to a router that is a root of the same call (a single-homed host next to
  // its switch, a stub router next to its neighbour: "every neighbour of this router" root sets hold many) needs no run of
  // its own — its rows are its neighbour's, one link further (k_leaf_root_rows, where the rule is spelled out).  Class 3;
  // configs[4]: 50 of the 101 roots.  Not on hop-count-like graphs, not when sums may saturate (max-path metric
  // 0xFFFFFFFF: the sweep kernels send such roots to the sequential kernel), not behind an overloaded or non-expandable
  // neighbour (its tree as a root is not its tree as a transit hop).  HSPF_VARIANT bit 23: off (A/B).
  std::vector<LeafRootJob> leaf_jobs;
  if (!(ctx->variant & 8388608u) && !g->hopcount_like && g->max_path_metric != HSPF_DIST_INF && g->twoway.size() == g->col.size()) {
    std::vector<std::pair<uint32_t, uint32_t>> where;          // (vertex, first index among the roots)
    where.reserve(n_roots);
    for (uint32_t i = 0; i < n_roots; ++i) if (roots[i] != HSPF_NO_ROOT) where.emplace_back(roots[i], i);
    std::sort(where.begin(), where.end());
    auto row_of = [&](uint32_t vtx) -> int64_t {
      auto it = std::lower_bound(where.begin(), where.end(), std::make_pair(vtx, 0u));
      return it != where.end() && it->first == vtx ? (int64_t)it->second : -1;
    };
    std::vector<int64_t> parent_row(n_roots, -1);
    std::vector<uint32_t> link_of(n_roots, 0);
    for (uint32_t i = 0; i < n_roots; ++i) {
      const uint32_t h = roots[i];
      if (h == HSPF_NO_ROOT || (g->vflags[h] & (HSPF_VF_NETWORK | HSPF_VF_NO_EXPAND))) continue;
      uint32_t kept = 0, link = 0;
      for (uint32_t k = g->rb(h); k < g->re(h) && kept < 2u; ++k) if (g->twoway[k]) { ++kept; link = k; }
      if (kept != 1u) continue;
      const uint32_t pv = g->col[link];
      if (pv == h || (g->vflags[pv] & (HSPF_VF_NETWORK | HSPF_VF_NO_TRANSIT | HSPF_VF_NO_EXPAND))) continue;
      if (g->rlen[pv] > 65536u) continue;   // (the scan below; such a neighbour's leaves take the ordinary path)
      uint32_t back = 0;                                             // p's links to h: exactly one (two would be two kept in-links)
      for (uint32_t k = g->rb(pv); k < g->re(pv); ++k) back += g->col[k] == h ? 1u : 0u;
      if (back != 1u) continue;
      const int64_t pr = row_of(pv);
      if (pr < 0) continue;
      parent_row[i] = pr; link_of[i] = link;
    }
    for (uint32_t i = 0; i < n_roots; ++i) {
      if (parent_row[i] < 0 || parent_row[(size_t)parent_row[i]] >= 0) continue;    // (two leaves facing each other: both run)
      leaf_jobs.push_back(LeafRootJob{roots[i], (uint32_t)parent_row[i], i, link_of[i], link_of[i] - g->rb(roots[i])});
      cnt[cls[i]]--;
      cls[i] = 3;
    }
  }
  const uint32_t n_run = n_roots - (uint32_t)leaf_jobs.size();
  const bool fused_ok = n < (1u << 23) && !(ctx->variant & 1u);
  // off the two-phase path with everybody who does not need it, when that saves at least one 64-root batch there (a
  // two-phase batch moves 8W-byte mask rows and costs several fused ones)
  const bool split2 = fused_ok && cnt[2] > 0 && cnt[0] + cnt[1] > 0 && (cnt[2] + 63) / 64 < (n_run + 63) / 64;
  const bool split1 = fused_ok && m_narrow > 0 && cnt[1] > 0 && cnt[0] >= 256;  // keep the narrow ones narrow
  if ((cnt[2] > 0 && !split2) || (!split1 && !split2)) {       // one run as before (two-phase for all / fused for all)
    if (leaf_jobs.empty()) return run_impl(ctx, g, roots, n_roots, run_flags, out, host_out);
    for (auto &c : cls) if (c != 3) c = 0;                      // ... of the roots that are not derived
  } else if (!split1) {
    for (auto &c : cls) if (c == 1) c = 0;                      // classes 0 and 1 share one fused run
  }
  // class order, stable inside a class
  std::vector<uint32_t> proots(n_run), dest(n_run);
  uint32_t start[4] = {0, 0, 0, 0};
  { uint32_t c0 = 0, c1 = 0; for (auto c : cls) { c0 += c == 0; c1 += c == 1; } start[1] = c0; start[2] = c0 + c1; start[3] = n_run; }
  { uint32_t pos[3] = {start[0], start[1], start[2]};
    for (uint32_t r = 0; r < n_roots; ++r) { if (cls[r] == 3) continue; const uint32_t i = pos[cls[r]]++; proots[i] = roots[r]; dest[i] = r; } }
  const size_t rn = (size_t)n_roots * n;
  const bool want_mask = out->first_hop_mask != nullptr, want_rank = (run_flags & HSPF_RUN_POP_RANK) && out->pop_rank;
  if (want_mask && out->n_mask_words == 0) return HSPF_E_INVAL;
  const uint32_t W = want_mask ? out->n_mask_words : 0;
  int rc;
  // Every class writes its rows straight to their final places (row map): into the caller's device buffers, or into
  // the staging that is copied out once at the end.
  hipStream_t s = ctx->stream;
  hspf_result U = *out;
  if (host_out) {
    if ((rc = ensure(ctx, ctx->o_dist, rn * 4))) return rc;
    U.dist = (uint32_t *)ctx->o_dist.p;
    if (out->hops) { if ((rc = ensure(ctx, ctx->o_hops, rn * 2))) return rc; U.hops = (uint16_t *)ctx->o_hops.p; }
    if (out->vflags_out) { if ((rc = ensure(ctx, ctx->o_flags, rn * 2))) return rc; U.vflags_out = (uint16_t *)ctx->o_flags.p; }
    if (want_mask) { if ((rc = ensure(ctx, ctx->o_mask, rn * 8 * W))) return rc; U.first_hop_mask = (uint64_t *)ctx->o_mask.p; }
    if (want_rank) { if ((rc = ensure(ctx, ctx->o_rank, rn * 4))) return rc; U.pop_rank = (uint32_t *)ctx->o_rank.p; }
  }
  hspf_stats acc{};
  // The classes are independent runs (own roots, own state, own rows of the output): on a context that may have lanes
  // (hspf_run_device_async) all but the last go to the lanes and run NEXT TO each other.  configs[4] (fat-tree, 51 switch
  // roots through k_fw + 50 host roots through k_fused): the k_fused class is a chain of latency-bound sweeps over the
  // switches' 100-link rows, the k_fw class is bound by the bytes of its 24-byte state — side by side they take little
  // more than the longer one (profiles/r04_notes.md, r04w).  HSPF_VARIANT bit 20: one after the other (A/B).
  int n_cls = 0, last_cls = -1;
  for (int c = 0; c < 3; ++c) if (start[c + 1] > start[c]) { ++n_cls; last_cls = c; }
  const bool side_by_side = n_cls >= 2 && !ctx->parent && ctx->lanes_cfg > 0 && !(ctx->variant & 1048576u) &&
                            (uint64_t)n * n_roots >= (1ull << 22) && lanes_ensure(ctx) == HSPF_OK;
  uint64_t cls_ticket[3] = {0, 0, 0};
  const auto t_cls = std::chrono::steady_clock::now();
  if (side_by_side)
    for (int c = 0; c < 3; ++c) {
      const uint32_t off = start[c], nr = start[c + 1] - start[c];
      if (nr == 0 || c == last_cls) continue;
      hspf_lane::Job job{g, std::vector<uint32_t>(proots.begin() + off, proots.begin() + off + nr), run_flags, U, 0,
                         std::vector<uint32_t>(dest.begin() + off, dest.begin() + off + nr), n_roots};
      cls_ticket[c] = lane_submit(ctx, std::move(job));
    }
  int rc_first = HSPF_OK;
  for (int c = 2; c >= 0; --c) {                                 // the class this context runs itself first, then the lanes' results
    const uint32_t off = start[c], nr = start[c + 1] - start[c];
    if (nr == 0) continue;
    hspf_stats lane_st{};
    if (cls_ticket[c]) rc = lane_collect(ctx, cls_ticket[c], &lane_st);
    else rc = run_impl(ctx, g, proots.data() + off, nr, run_flags, &U, false, dest.data() + off, n_roots);
    if (rc) { if (!rc_first) rc_first = rc; continue; }          // (every submitted class is collected before the call returns)
    const hspf_stats &p = cls_ticket[c] ? lane_st : ctx->stats;
    acc.n_roots += p.n_roots; acc.n_batches += p.n_batches; acc.n_relax_launches += p.n_relax_launches;
    acc.n_dag_launches += p.n_dag_launches; acc.n_exact_roots += p.n_exact_roots; acc.n_repaired_roots += p.n_repaired_roots; acc.repair_sweeps = std::max(acc.repair_sweeps, p.repair_sweeps); acc.repair_evals += p.repair_evals; acc.repair_groups += p.repair_groups; acc.ms_repair += p.ms_repair;
    acc.n_mask_words = std::max(acc.n_mask_words, p.n_mask_words);
    acc.ms_total += p.ms_total; acc.ms_relax += p.ms_relax; acc.ms_dag += p.ms_dag; acc.ms_finish += p.ms_finish;
    acc.state_bytes = std::max(acc.state_bytes, p.state_bytes); acc.narrow_overflow += p.narrow_overflow;
    acc.rows_recomputed += p.rows_recomputed;
    acc.dbg[0] |= p.dbg[0]; acc.dbg[1] = ((acc.dbg[1] | p.dbg[1]) & 0x80000000u) | ((acc.dbg[1] & 0x7FFFFFFFu) + (p.dbg[1] & 0x7FFFFFFFu));
    acc.dbg[2] += p.dbg[2]; acc.dbg[3] += p.dbg[3];
  }
  if (rc_first) return rc_first;
  if (!leaf_jobs.empty()) {                                      // the leaf roots' rows, from their neighbours' finished ones
    const auto t_leaf = std::chrono::steady_clock::now();
    (void)hipSetDevice(ctx->device);
    if ((rc = ensure(ctx, ctx->leaf_jobs, leaf_jobs.size() * sizeof(LeafRootJob), false))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->leaf_jobs.p, leaf_jobs.data(), leaf_jobs.size() * sizeof(LeafRootJob), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_leaf_root_rows, dim3((n + 255u) / 256u, (uint32_t)leaf_jobs.size()), dim3(256), 0, s, n,
                       (const LeafRootJob *)ctx->leaf_jobs.p, (const uint32_t *)g->d_row_ptr[g->cur], (const uint32_t *)g->d_metric[g->cur], g->max_path_metric, W, U.dist, U.hops,
                       U.vflags_out, U.first_hop_mask);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(s));                         // (the pageable job list; and a run returns when its results are there)
    acc.n_roots += (uint32_t)leaf_jobs.size();
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_leaf).count();
    acc.ms_total += ms; acc.ms_finish += ms;                       // (host clock: copy of the job list, the launch, the wait)
  }
  if (side_by_side) {                                            // what the classes took together, not their sum
    const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_cls).count();
    if (wall < acc.ms_total) {
      const float k = wall / acc.ms_total;
      acc.ms_total = wall; acc.ms_relax *= k; acc.ms_dag *= k; acc.ms_finish *= k;
    }
  }
  if (host_out) {
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], s));
    HIPCHK(ctx, hipMemcpyAsync(out->dist, U.dist, rn * 4, hipMemcpyDeviceToHost, s));
    if (out->hops) HIPCHK(ctx, hipMemcpyAsync(out->hops, U.hops, rn * 2, hipMemcpyDeviceToHost, s));
    if (out->vflags_out) HIPCHK(ctx, hipMemcpyAsync(out->vflags_out, U.vflags_out, rn * 2, hipMemcpyDeviceToHost, s));
    if (want_mask) HIPCHK(ctx, hipMemcpyAsync(out->first_hop_mask, U.first_hop_mask, rn * 8 * W, hipMemcpyDeviceToHost, s));
    if (want_rank) HIPCHK(ctx, hipMemcpyAsync(out->pop_rank, U.pop_rank, rn * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) acc.ms_d2h = ms;
  }
  ctx->stats = acc;
  return HSPF_OK;
}

int hspf_run(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags, hspf_result *out) {
  return guarded(ctx, [&]() { return run_classes(ctx, g, roots, n_roots, run_flags, out, true); });
}

int hspf_run_device(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags, hspf_result *out_device) {
  return guarded(ctx, [&]() { return run_classes(ctx, g, roots, n_roots, run_flags, out_device, false); });
}

// ---- packed results (ABI 7) ---------------------------------------------------------------------
int hspf_host_alloc(hspf_ctx *ctx, size_t bytes, void **out) {
  if (!ctx || !out || bytes == 0) return HSPF_E_INVAL;
  *out = nullptr;
  (void)hipSetDevice(ctx->device);
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { *out = nullptr; try { ctx->last_error = std::string("hipHostMalloc: ") + hipGetErrorString(e); } catch (...) {} return HSPF_E_NOMEM; }
  return HSPF_OK;
}
void hspf_host_free(hspf_ctx *ctx, void *p) {
  if (!p) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  (void)hipHostFree(p);
}

int hspf_device_alloc(hspf_ctx *ctx, size_t bytes, void **out) {
  if (!ctx || !out || bytes == 0) return HSPF_E_INVAL;
  *out = nullptr;
  (void)hipSetDevice(ctx->device);
  const hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) { *out = nullptr; try { ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(e); } catch (...) {} return HSPF_E_NOMEM; }
  return HSPF_OK;
}
void hspf_device_free(hspf_ctx *ctx, void *p) {
  if (!p) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  (void)hipFree(p);
}
int hspf_device_to_host(hspf_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
  if (!ctx || (bytes && (!dst_host || !src_dev))) return HSPF_E_INVAL;
  if (bytes == 0) return HSPF_OK;
  (void)hipSetDevice(ctx->device);
  return guarded(ctx, [&]() -> int {
    const int rc = copy_to_host(ctx, dst_host, src_dev, bytes, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return HSPF_OK;
  });
}
int hspf_host_to_device(hspf_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
  if (!ctx || (bytes && (!dst_dev || !src_host))) return HSPF_E_INVAL;
  if (bytes == 0) return HSPF_OK;
  (void)hipSetDevice(ctx->device);
  return guarded(ctx, [&]() -> int { HIPCHK(ctx, hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice)); return HSPF_OK; });
}

static int run_packed_entry(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                            void *words, size_t cap_bytes, bool host, hspf_packed_layout *layout, uint8_t *root_status) {
  if (!ctx || !g || !roots || n_roots == 0 || !words || !layout) return HSPF_E_INVAL;
  return guarded(ctx, [&]() {
    PackedReq pk;
    pk.dst = words; pk.cap = cap_bytes; pk.host = host; pk.root_status = root_status;
    const int rc = run_impl(ctx, g, roots, n_roots, run_flags, nullptr, host, nullptr, 0, false, &pk);
    if (rc == HSPF_OK) *layout = pk.layout;
    return rc;
  });
}
int hspf_run_packed(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                    void *words_host, size_t cap_bytes, hspf_packed_layout *layout, uint8_t *root_status) {
  return run_packed_entry(ctx, g, roots, n_roots, run_flags, words_host, cap_bytes, true, layout, root_status);
}
int hspf_run_packed_device(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                           void *words_dev, size_t cap_bytes, hspf_packed_layout *layout, uint8_t *root_status) {
  return run_packed_entry(ctx, g, roots, n_roots, run_flags, words_dev, cap_bytes, false, layout, root_status);
}

int hspf_get_stats(const hspf_ctx *ctx, hspf_stats *out) {
  if (!ctx || !out) return HSPF_E_INVAL;
  *out = ctx->stats;
  return HSPF_OK;
}

// ---- asynchronous runs ------------------------------------------------------------------------
// The reference runs one SPF per instance thread at a time (holo-protocol/src/lib.rs:427-430); what it does have is
// SEVERAL independent runs per event: one per area (holo-ospf/src/spf.rs:540-542), per level and topology
// (holo-isis/src/spf.rs:746-761), per neighbour (flooding/manet.rs:59-69).  These entry points let ONE caller thread keep
// several of them in flight on one GPU.

static int lanes_ensure(hspf_ctx *ctx) {
  if (!ctx->lanes.empty()) return HSPF_OK;
  if (ctx->parent) { ctx->last_error = "asynchronous runs on a lane context"; return HSPF_E_INVAL; }
  (void)hipSetDevice(ctx->device);
  for (uint32_t i = 0; i < ctx->lanes_cfg; ++i) {
    hspf_lane *ln = new (std::nothrow) hspf_lane();
    if (!ln) { lanes_shutdown(ctx); return HSPF_E_NOMEM; }
    int rc = hspf_init(ctx->device, &ln->sub);
    if (rc == HSPF_OK) {
      hspf_ctx *c = ln->sub;
      c->parent = ctx;
      // the caller's context decides the tunables (a lane's own hspf_init read the environment of a later moment)
      c->variant = ctx->variant; c->single_max_n = ctx->single_max_n; c->lv_max_roots = ctx->lv_max_roots; c->lv_min_n = ctx->lv_min_n; c->xcd_max_roots = ctx->xcd_max_roots; c->xcd_timeout_ms = ctx->xcd_timeout_ms; c->xcd_skew = ctx->xcd_skew; c->xcd_always = ctx->xcd_always;
      c->lean_dense_pct = ctx->lean_dense_pct; c->lean_stay_pct = ctx->lean_stay_pct; c->lean_dense_passes = ctx->lean_dense_passes;
      c->lean_head = ctx->lean_head; c->lean_passes = ctx->lean_passes; c->hub_deg = ctx->hub_deg; c->lean_multi_min_wgs = ctx->lean_multi_min_wgs;
      c->unit_heavy_deg = ctx->unit_heavy_deg; c->xcd_row_cost = ctx->xcd_row_cost;
    }
    if (rc != HSPF_OK) { if (ln->sub) hspf_shutdown(ln->sub); delete ln; lanes_shutdown(ctx); ctx->last_error = "could not create a lane context"; return rc; }
    for (auto &e : ln->stage_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) ctx->copy_stream = nullptr;
    ctx->lanes.push_back(ln);
    ln->th = std::thread([ln, nl = ctx->lanes_cfg]() {
      for (;;) {
        std::unique_lock<std::mutex> lk(ln->mu);
        ln->cv.wait(lk, [&] { return !ln->jobs.empty() || ln->quit; });
        if (ln->jobs.empty()) return;                               // quit, nothing pending
        hspf_lane::Job job = std::move(ln->jobs.front());
        ln->jobs.pop_front();
        ln->running = true;
        lk.unlock();
        ln->cv.notify_all();                                        // (a submitter may be waiting for room in the queue)
        (void)hipSetDevice(ln->sub->device);
        hipEvent_t job_copy_ev = nullptr;
        const int rc = guarded(ln->sub, [&]() {
          if (job.packed) {                                         // hspf_run_packed_async: the run and its copy to the host
            hspf_ctx *par = ln->sub->parent;
            const size_t need = (size_t)8 * job.roots.size() * job.g->n;
            const uint64_t max_pairs = 1ull << 26;                  // (run_impl's pass size: a call of several passes copies pass by pass itself)
            const bool one_pass = (uint64_t)job.roots.size() <= std::max<uint64_t>(64, (max_pairs / std::max<uint32_t>(job.g->n, 1)) / 64 * 64);
            int sel = -1;
            if (job.pk_pinned && one_pass && par && par->copy_stream) {
              sel = ln->stage_sel; ln->stage_sel ^= 1;
              if (ln->stage_busy[sel]) { (void)hipEventSynchronize(ln->stage_ev[sel]); ln->stage_busy[sel] = false; }   // its last copy is long over
              const int e2 = ensure(ln->sub, ln->stage[sel], need, false);
              if (e2) return e2;
              job.pk.dev_stage = ln->stage[sel].p;
            }
            int r2 = run_impl(ln->sub, job.g, job.roots.data(), (uint32_t)job.roots.size(), job.flags, nullptr, true, nullptr, 0, false, &job.pk);
            if (r2 == HSPF_OK && sel >= 0 && job.pk.copy_bytes) {
              hipError_t er = hipMemcpyAsync(job.pk.dst, job.pk.dev_stage, job.pk.copy_bytes, hipMemcpyDeviceToHost, par->copy_stream);
              if (er == hipSuccess) er = hipEventRecord(ln->stage_ev[sel], par->copy_stream);
              if (er != hipSuccess) { ln->sub->last_error = std::string("packed copy: ") + hipGetErrorString(er); return (int)HSPF_E_HIP; }
              ln->stage_busy[sel] = true;
              job_copy_ev = ln->stage_ev[sel];
            }
            return r2;
          }
          if (!job.dest.empty())                                    // one class of a run that the caller's context split (run_classes)
            return run_impl(ln->sub, job.g, job.roots.data(), (uint32_t)job.roots.size(), job.flags, &job.out, false, job.dest.data(), job.n_total);
          return run_classes(ln->sub, job.g, job.roots.data(), (uint32_t)job.roots.size(), job.flags, &job.out, false);
        });
        lk.lock();
        hspf_lane::Done &d = ln->done[(job.ticket / nl) & 7u];
        d.ticket = job.ticket; d.rc = rc; d.st = ln->sub->stats; d.layout = job.pk.layout; d.copy_ev = job_copy_ev;
        try { d.err = rc ? ln->sub->last_error : std::string(); } catch (...) {}
        ln->last_done = job.ticket; ln->running = false;
        lk.unlock();
        ln->cv.notify_all();
      }
    });
  }
  return HSPF_OK;
}

static void lanes_quiesce(hspf_ctx *ctx) {
  for (hspf_lane *ln : ctx->lanes) {
    std::unique_lock<std::mutex> lk(ln->mu);
    ln->cv.wait(lk, [&] { return ln->idle(); });
    // a finished run leaves the next run's scratch fill behind on the lane's stream, and that launch reads the graph's row flags
    (void)hipSetDevice(ln->sub->device);
    if (ln->sub->stream) (void)hipStreamSynchronize(ln->sub->stream);
  }
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
}

static void lanes_shutdown(hspf_ctx *ctx) {
  for (hspf_lane *ln : ctx->lanes) {
    { std::lock_guard<std::mutex> lk(ln->mu); ln->quit = true; }
    ln->cv.notify_all();
    if (ln->th.joinable()) ln->th.join();
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (auto &b : ln->stage) release(b);
    for (auto &e : ln->stage_ev) if (e) (void)hipEventDestroy(e);
    if (ln->sub) hspf_shutdown(ln->sub);
    delete ln;
  }
  ctx->lanes.clear();
  if (ctx->copy_stream) { (void)hipStreamDestroy(ctx->copy_stream); ctx->copy_stream = nullptr; }
}

int hspf_run_device_async(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                          const hspf_result *out_device, uint64_t *ticket) {
  if (!ctx || !g || !roots || !out_device || !ticket || n_roots == 0 || !out_device->dist) return HSPF_E_INVAL;
  return guarded(ctx, [&]() {
    int rc = lanes_ensure(ctx);
    if (rc) return rc;
    hspf_lane::Job job{g, std::vector<uint32_t>(roots, roots + n_roots), run_flags, *out_device, 0, {}, 0};
    *ticket = lane_submit(ctx, std::move(job));
    return (int)HSPF_OK;
  });
}

int hspf_run_packed_async(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                          void *words_host, size_t cap_bytes, uint8_t *root_status, uint64_t *ticket) {
  if (!ctx || !g || !roots || !words_host || !ticket || n_roots == 0) return HSPF_E_INVAL;
  return guarded(ctx, [&]() {
    int rc = lanes_ensure(ctx);
    if (rc) return rc;
    hspf_lane::Job job{g, std::vector<uint32_t>(roots, roots + n_roots), run_flags, hspf_result{}, 0, {}, 0};
    job.packed = true;
    job.pk.dst = words_host; job.pk.cap = cap_bytes; job.pk.host = true; job.pk.root_status = root_status;
    {
      hipPointerAttribute_t at{};
      job.pk_pinned = hipPointerGetAttributes(&at, words_host) == hipSuccess && at.type == hipMemoryTypeHost;
      if (!job.pk_pinned) (void)hipGetLastError();
    }
    *ticket = lane_submit(ctx, std::move(job));
    return (int)HSPF_OK;
  });
}

// Next ticket, onto its lane's queue (lanes exist: lanes_ensure).
static uint64_t lane_submit(hspf_ctx *ctx, hspf_lane::Job &&job) {
  const uint64_t t = ctx->next_ticket++;
  hspf_lane *ln = ctx->lanes[t % ctx->lanes.size()];
  job.ticket = t;
  std::unique_lock<std::mutex> lk(ln->mu);
  ln->cv.wait(lk, [&] { return ln->jobs.size() < LANE_QUEUE_MAX; });   // room in the lane's queue (it runs its jobs in ticket order)
  ln->jobs.push_back(std::move(job));
  lk.unlock();
  ln->cv.notify_all();
  return t;
}

// Waits for a ticket; its result code, statistics and (on failure) error text.  Does not touch ctx->stats.
static int lane_collect(hspf_ctx *ctx, uint64_t ticket, hspf_stats *stats, hspf_packed_layout *layout) {
  hspf_lane *ln = ctx->lanes[ticket % ctx->lanes.size()];
  std::unique_lock<std::mutex> lk(ln->mu);
  ln->cv.wait(lk, [&] { return ln->last_done >= ticket; });
  const hspf_lane::Done &d = ln->done[(ticket / ctx->lanes.size()) & 7u];
  if (d.ticket != ticket) { ctx->last_error = "hspf_wait: the ticket's result is gone (more than eight later runs on its lane)"; return HSPF_E_INVAL; }
  if (stats) *stats = d.st;
  if (layout) *layout = d.layout;
  if (d.rc) { try { ctx->last_error = d.err; } catch (...) {} }
  const hipEvent_t cev = d.copy_ev;
  const int drc = d.rc;
  lk.unlock();
  // a packed ticket: its words are on their way on the copy stream (a later copy out of the same staging block re-records
  // the event: waiting for that one covers this one, the stream is in order)
  if (drc == HSPF_OK && cev && hipEventSynchronize(cev) != hipSuccess) { ctx->last_error = "hspf_wait: the copy of the packed words failed"; return HSPF_E_HIP; }
  return drc;
}

int hspf_wait(hspf_ctx *ctx, uint64_t ticket, hspf_stats *stats) {
  if (!ctx || ticket == 0 || ticket >= ctx->next_ticket || ctx->lanes.empty()) return HSPF_E_INVAL;
  hspf_stats st{};
  const int rc = lane_collect(ctx, ticket, &st);
  if (rc == HSPF_E_INVAL && ctx->last_error.rfind("hspf_wait: the ticket", 0) == 0) return rc;
  ctx->stats = st;
  if (stats) *stats = st;
  return rc;
}

int hspf_wait_packed(hspf_ctx *ctx, uint64_t ticket, hspf_packed_layout *layout, hspf_stats *stats) {
  if (!ctx || !layout || ticket == 0 || ticket >= ctx->next_ticket || ctx->lanes.empty()) return HSPF_E_INVAL;
  hspf_stats st{};
  const int rc = lane_collect(ctx, ticket, &st, layout);
  if (rc == HSPF_E_INVAL && ctx->last_error.rfind("hspf_wait: the ticket", 0) == 0) return rc;
  ctx->stats = st;
  if (stats) *stats = st;
  return rc;
}

int hspf_wait_all(hspf_ctx *ctx) {
  if (!ctx) return HSPF_E_INVAL;
  lanes_quiesce(ctx);
  return HSPF_OK;
}

uint32_t hspf_async_lanes(const hspf_ctx *ctx) { return ctx ? (ctx->lanes.empty() ? ctx->lanes_cfg : (uint32_t)ctx->lanes.size()) : 0u; }

static int routes_device_impl(hspf_ctx *ctx, uint32_t n_vertices, uint32_t n_roots, uint32_t n_mask_words,
                              const uint32_t *dist_dev, const uint16_t *flags_dev, const uint64_t *mask_dev,
                              const hspf_prefix_table *t, hspf_routes *out) {
  if (!ctx || !t || !out || !dist_dev || !flags_dev || !mask_dev || !out->best_metric || !out->best_entry ||
      !out->nexthop_mask || n_roots == 0 || n_mask_words == 0 || !t->pfx_ptr || (t->n_entries && (!t->pfx_vertex || !t->pfx_metric)))
    return HSPF_E_INVAL;
  if (t->flags & ~(HSPF_PFX_SATURATING | HSPF_PFX_LAST_MIN | HSPF_PFX_ORDERED | HSPF_PFX_RESIDENT)) { ctx->last_error = "hspf_prefix_table: unknown flags"; return HSPF_E_INVAL; }
  const bool ordered = (t->flags & HSPF_PFX_ORDERED) != 0;
  if (ordered && ((t->flags & HSPF_PFX_LAST_MIN) || (t->n_entries && !t->pfx_origin) ||
                  (t->init_exists && (!t->init_metric || !t->init_origin)))) {
    ctx->last_error = "hspf_prefix_table: HSPF_PFX_ORDERED needs pfx_origin (and init_metric / init_origin with init_exists), not LAST_MIN";
    return HSPF_E_INVAL;
  }
  // A prefix table changes with the LSDB, not with every SPF run.  HSPF_PFX_RESIDENT is the caller's word that the three
  // main arrays are exactly what its previous call on this context passed (same pointers, same sizes, contents untouched):
  // the checks and the three pageable copies — most of the call at 120 000 prefixes — are skipped.  Anything that does not
  // match what was recorded then takes the full path.  Ordered tables (origins, initial state) always do.
  const bool resident = (t->flags & HSPF_PFX_RESIDENT) && !ordered && ctx->pf_shadow_ok && ctx->pf_shadow_nv == n_vertices &&
                        ctx->pf_res_np == t->n_prefixes && ctx->pf_res_ne == t->n_entries && ctx->pf_res_ptr == (const void *)t->pfx_ptr &&
                        ctx->pf_res_vtx == (const void *)t->pfx_vertex && ctx->pf_res_met == (const void *)t->pfx_metric;
  if (!resident) {
    ctx->pf_shadow_ok = false;
    if (t->pfx_ptr[0] != 0 || t->pfx_ptr[t->n_prefixes] != t->n_entries) { ctx->last_error = "pfx_ptr malformed"; return HSPF_E_INVAL; }
    for (uint32_t p = 0; p < t->n_prefixes; ++p)
      if (t->pfx_ptr[p + 1] < t->pfx_ptr[p]) { ctx->last_error = "pfx_ptr not monotone"; return HSPF_E_INVAL; }
    for (uint32_t e = 0; e < t->n_entries; ++e)
      if ((ordered ? (t->pfx_vertex[e] & ~HSPF_PFX_ENTRY_NETWORK) : t->pfx_vertex[e]) >= n_vertices) { ctx->last_error = "pfx_vertex out of range"; return HSPF_E_INVAL; }
  }
  if (t->n_prefixes == 0) return HSPF_OK;
  (void)hipSetDevice(ctx->device);
  int rc;
  hipStream_t s = ctx->stream;
  if (!resident) {
    if ((rc = ensure(ctx, ctx->pf_ptr, (size_t)(t->n_prefixes + 1) * 4, false))) return rc;
    if ((rc = ensure(ctx, ctx->pf_vtx, std::max<size_t>(t->n_entries, 1) * 4, false))) return rc;
    if ((rc = ensure(ctx, ctx->pf_met, std::max<size_t>(t->n_entries, 1) * 4, false))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf_ptr.p, t->pfx_ptr, (size_t)(t->n_prefixes + 1) * 4, hipMemcpyHostToDevice, s));
    if (t->n_entries) {
      HIPCHK(ctx, hipMemcpyAsync(ctx->pf_vtx.p, t->pfx_vertex, (size_t)t->n_entries * 4, hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(ctx->pf_met.p, t->pfx_metric, (size_t)t->n_entries * 4, hipMemcpyHostToDevice, s));
    }
    if (!ordered) {
      ctx->pf_res_np = t->n_prefixes; ctx->pf_res_ne = t->n_entries; ctx->pf_res_ptr = t->pfx_ptr; ctx->pf_res_vtx = t->pfx_vertex;
      ctx->pf_res_met = t->pfx_metric; ctx->pf_shadow_nv = n_vertices;
      ctx->pf_shadow_ok = true;       // (the copies above are enqueued on the context's stream: every later use is ordered behind them)
    }
  }
  const uint8_t *d_iex = nullptr;
  const uint32_t *d_imet = nullptr, *d_iorg = nullptr;
  if (ordered) {
    // origins, then the optional initial state: [n_entries] u32 | [n_prefixes] u32 metric | [n_prefixes] u32 origin | [n_prefixes] u8
    const size_t ne = std::max<size_t>(t->n_entries, 1), np = t->n_prefixes;
    if ((rc = ensure(ctx, ctx->pf_org, (ne + 2 * np) * 4 + np))) return rc;
    uint32_t *base = (uint32_t *)ctx->pf_org.p;
    if (t->n_entries) HIPCHK(ctx, hipMemcpyAsync(base, t->pfx_origin, (size_t)t->n_entries * 4, hipMemcpyHostToDevice, s));
    if (t->init_exists) {
      HIPCHK(ctx, hipMemcpyAsync(base + ne, t->init_metric, np * 4, hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(base + ne + np, t->init_origin, np * 4, hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(base + ne + 2 * np, t->init_exists, np, hipMemcpyHostToDevice, s));
      d_imet = base + ne; d_iorg = base + ne + np; d_iex = (const uint8_t *)(base + ne + 2 * np);
    }
  }
  // gridDim.y is capped at 65535: "every router as a root" on a large LSDB goes in slabs of roots
  for (uint32_t r0 = 0; r0 < n_roots; r0 += 65535u) {
    const uint32_t nr = std::min(65535u, n_roots - r0);
    const size_t ov = (size_t)r0 * n_vertices, op = (size_t)r0 * t->n_prefixes;
    if (ordered)
      hipLaunchKernelGGL(k_routes_ordered, dim3((t->n_prefixes + 255) / 256, nr), dim3(256), 0, s, n_vertices, nr, n_mask_words,
                         t->n_prefixes, (const uint32_t *)ctx->pf_ptr.p, (const uint32_t *)ctx->pf_vtx.p,
                         (const uint32_t *)ctx->pf_met.p, (const uint32_t *)ctx->pf_org.p, d_iex, d_imet, d_iorg,
                         dist_dev + ov, flags_dev + ov, mask_dev + ov * n_mask_words,
                         out->best_metric + op, out->best_entry + op, out->nexthop_mask + op * n_mask_words);
    else
    hipLaunchKernelGGL(k_routes, dim3((t->n_prefixes + 255) / 256, nr), dim3(256), 0, s, n_vertices, nr, n_mask_words,
                       t->n_prefixes, (const uint32_t *)ctx->pf_ptr.p, (const uint32_t *)ctx->pf_vtx.p,
                       (const uint32_t *)ctx->pf_met.p, dist_dev + ov, flags_dev + ov, mask_dev + ov * n_mask_words,
                       out->best_metric + op, out->best_entry + op, out->nexthop_mask + op * n_mask_words, t->flags);
  }
  HIPCHK(ctx, hipStreamSynchronize(s));       // the table was read from caller-owned host memory
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { ctx->last_error = std::string("k_routes: ") + hipGetErrorString(le); return HSPF_E_HIP; }
  return HSPF_OK;
}

// (entry points that assign strings / grow scratch run through guarded(): nothing unwinds across the C boundary)
int hspf_routes_device(hspf_ctx *ctx, uint32_t n_vertices, uint32_t n_roots, uint32_t n_mask_words,
                       const uint32_t *dist_dev, const uint16_t *flags_dev, const uint64_t *mask_dev,
                       const hspf_prefix_table *t, hspf_routes *out) {
  if (!ctx) return HSPF_E_INVAL;
  return guarded(ctx, [&]() -> int { return routes_device_impl(ctx, n_vertices, n_roots, n_mask_words, dist_dev, flags_dev, mask_dev, t, out); });
}

// ---- several areas, one RIB: the fold on the device (include/holo_spf_hip.h) ----------------------------------------
int hspf_rib_clear_device(hspf_ctx *ctx, const hspf_rib_device *rib) {
  if (!ctx || !rib || !rib->best_metric || !rib->best_entry || !rib->nexthop_mask || !rib->origin || rib->n_mask_words == 0) return HSPF_E_INVAL;
  if (rib->n_prefixes == 0) return HSPF_OK;
  return guarded(ctx, [&]() -> int {
    (void)hipSetDevice(ctx->device);
    hipLaunchKernelGGL(k_rib_clear, dim3((rib->n_prefixes + 255) / 256), dim3(256), 0, ctx->stream, rib->n_prefixes, rib->n_mask_words,
                       rib->best_metric, rib->best_entry, rib->nexthop_mask, rib->origin);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return HSPF_OK;
  });
}

int hspf_rib_fold_device(hspf_ctx *ctx, uint32_t n_vertices, uint32_t area_mask_words, const uint32_t *dist_dev, const uint16_t *flags_dev,
                         const uint64_t *mask_dev, const hspf_prefix_table *t, const uint32_t *prefix_map, uint32_t area_index, uint32_t word_offset,
                         const hspf_rib_device *rib) {
  if (!ctx || !t || !rib || !dist_dev || !flags_dev || !mask_dev || !rib->best_metric || !rib->best_entry || !rib->nexthop_mask || !rib->origin ||
      area_mask_words == 0 || rib->n_mask_words == 0 || !t->pfx_ptr || (t->n_prefixes && !prefix_map) || (t->n_entries && (!t->pfx_vertex || !t->pfx_metric || !t->pfx_origin)))
    return HSPF_E_INVAL;
  return guarded(ctx, [&]() -> int {
    if (!(t->flags & HSPF_PFX_ORDERED) || (t->flags & HSPF_PFX_LAST_MIN)) { ctx->last_error = "hspf_rib_fold_device: the table must be HSPF_PFX_ORDERED"; return HSPF_E_INVAL; }
    if (area_index >= 255u || t->n_entries >= (1u << 24)) { ctx->last_error = "hspf_rib_fold_device: at most 255 areas, 2^24 entries per area table"; return HSPF_E_INVAL; }
    if ((uint64_t)word_offset + area_mask_words > rib->n_mask_words) { ctx->last_error = "hspf_rib_fold_device: the area's mask words do not fit the instance-wide numbering"; return HSPF_E_INVAL; }
    if (t->pfx_ptr[0] != 0 || t->pfx_ptr[t->n_prefixes] != t->n_entries) { ctx->last_error = "pfx_ptr malformed"; return HSPF_E_INVAL; }
    for (uint32_t p = 0; p < t->n_prefixes; ++p) {
      if (t->pfx_ptr[p + 1] < t->pfx_ptr[p]) { ctx->last_error = "pfx_ptr not monotone"; return HSPF_E_INVAL; }
      if (prefix_map[p] >= rib->n_prefixes) { ctx->last_error = "prefix_map out of range"; return HSPF_E_INVAL; }
    }
    {                                                          // every instance prefix at most once (two threads must not share a state row)
      std::vector<uint32_t> seen(prefix_map, prefix_map + t->n_prefixes);
      std::sort(seen.begin(), seen.end());
      if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) { ctx->last_error = "prefix_map names an instance prefix twice"; return HSPF_E_INVAL; }
    }
    for (uint32_t e = 0; e < t->n_entries; ++e)
      if ((t->pfx_vertex[e] & ~HSPF_PFX_ENTRY_NETWORK) >= n_vertices) { ctx->last_error = "pfx_vertex out of range"; return HSPF_E_INVAL; }
    if (t->n_prefixes == 0) return HSPF_OK;
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    ctx->pf_shadow_ok = false;                                 // (the resident plain table, if any, is overwritten below)
    int rc;
    const size_t ne = std::max<size_t>(t->n_entries, 1), np = t->n_prefixes;
    if ((rc = ensure(ctx, ctx->pf_ptr, (np + 1) * 4, false))) return rc;
    if ((rc = ensure(ctx, ctx->pf_vtx, ne * 4, false))) return rc;
    if ((rc = ensure(ctx, ctx->pf_met, ne * 4, false))) return rc;
    if ((rc = ensure(ctx, ctx->pf_org, (ne + np) * 4, false))) return rc;
    uint32_t *d_org = (uint32_t *)ctx->pf_org.p, *d_map = d_org + ne;
    HIPCHK(ctx, hipMemcpyAsync(ctx->pf_ptr.p, t->pfx_ptr, (np + 1) * 4, hipMemcpyHostToDevice, s));
    if (t->n_entries) {
      HIPCHK(ctx, hipMemcpyAsync(ctx->pf_vtx.p, t->pfx_vertex, (size_t)t->n_entries * 4, hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(ctx->pf_met.p, t->pfx_metric, (size_t)t->n_entries * 4, hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(d_org, t->pfx_origin, (size_t)t->n_entries * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(ctx, hipMemcpyAsync(d_map, prefix_map, np * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_rib_fold, dim3((t->n_prefixes + 255) / 256), dim3(256), 0, s, n_vertices, area_mask_words, t->n_prefixes,
                       (const uint32_t *)ctx->pf_ptr.p, (const uint32_t *)ctx->pf_vtx.p, (const uint32_t *)ctx->pf_met.p, (const uint32_t *)d_org,
                       (const uint32_t *)d_map, dist_dev, flags_dev, mask_dev, area_index << 24, word_offset, rib->n_mask_words,
                       rib->best_metric, rib->best_entry, rib->nexthop_mask, rib->origin);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(s));                      // the table was read from caller-owned host memory
    return HSPF_OK;
  });
}

static int routes_diff_device_impl(hspf_ctx *ctx, uint32_t n_roots, uint32_t n_prefixes, uint32_t n_mask_words,
                                   const hspf_routes *old_dev, const hspf_routes *new_dev,
                                   uint8_t *action_dev, uint32_t *changed_dev, uint32_t *changed_ptr_dev) {
  if (!ctx || !old_dev || !new_dev || !action_dev || !changed_dev || !changed_ptr_dev || n_roots == 0 || n_mask_words == 0 ||
      !old_dev->best_metric || !old_dev->best_entry || !old_dev->nexthop_mask ||
      !new_dev->best_metric || !new_dev->best_entry || !new_dev->nexthop_mask)
    return HSPF_E_INVAL;
  const uint64_t count64 = (uint64_t)n_roots * n_prefixes;
  if (count64 >= 0xFFFFFFF0ull) { ctx->last_error = "hspf_routes_diff_device: more than 2^32 (root, prefix) pairs"; return HSPF_E_INVAL; }
  (void)hipSetDevice(ctx->device);
  hipStream_t s = ctx->stream;
  const uint32_t count = (uint32_t)count64;
  ctx->last_diff_count = 0;
  if (count == 0) {                                          // no prefixes: every root's list is empty
    HIPCHK(ctx, hipMemsetAsync(changed_ptr_dev, 0, ((size_t)n_roots + 1) * 4, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return HSPF_OK;
  }
  // scratch: pos[count + 1] | sums | flag[count] (u8)
  const size_t nsums = (size_t)count / GB_TILE + 4;
  int rc = ensure(ctx, ctx->gb, ((size_t)count + 1 + nsums) * 4 + count + 64);
  if (rc) return rc;
  uint32_t *pos = (uint32_t *)ctx->gb.p, *sums = pos + count + 1;
  uint8_t *flag = (uint8_t *)(sums + nsums);
  const dim3 grid((unsigned)(((size_t)count + 1 + 255) / 256));
  if (count)
    hipLaunchKernelGGL(k_routes_diff, grid, dim3(256), 0, s, (size_t)count, n_mask_words, old_dev->best_metric, old_dev->best_entry,
                       old_dev->nexthop_mask, new_dev->best_metric, new_dev->best_entry, new_dev->nexthop_mask, action_dev, flag);
  gb_scan<uint8_t>(s, flag, count, pos, sums);
  hipLaunchKernelGGL(k_routes_diff_scatter, dim3((unsigned)(((size_t)std::max(count, n_roots + 1) + 255) / 256)), dim3(256), 0, s,
                     (size_t)count, n_prefixes, n_roots, (const uint8_t *)flag, (const uint32_t *)pos, changed_dev, changed_ptr_dev);
  // the total rides back with the synchronisation this call ends on anyway (the hand-off sizes its one copy by it)
  HIPCHK(ctx, hipMemcpyAsync(&ctx->h_info->kept, pos + count, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  ctx->last_diff_count = ctx->h_info->kept;
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { ctx->last_error = std::string("k_routes_diff: ") + hipGetErrorString(le); return HSPF_E_HIP; }
  return HSPF_OK;
}

int hspf_routes_diff_device(hspf_ctx *ctx, uint32_t n_roots, uint32_t n_prefixes, uint32_t n_mask_words,
                            const hspf_routes *old_dev, const hspf_routes *new_dev,
                            uint8_t *action_dev, uint32_t *changed_dev, uint32_t *changed_ptr_dev) {
  if (!ctx) return HSPF_E_INVAL;
  return guarded(ctx, [&]() -> int { return routes_diff_device_impl(ctx, n_roots, n_prefixes, n_mask_words, old_dev, new_dev, action_dev, changed_dev, changed_ptr_dev); });
}

uint32_t hspf_routes_diff_count(const hspf_ctx *ctx) { return ctx ? ctx->last_diff_count : 0u; }

int hspf_routes_pack(hspf_ctx *ctx, uint32_t n_roots, uint32_t n_prefixes, uint32_t n_mask_words, const hspf_routes *new_dev,
                     const uint8_t *action_dev, const uint32_t *changed_dev, const uint32_t *changed_ptr_dev,
                     uint32_t n_records, uint32_t *records_host) {
  if (!ctx || !new_dev || !action_dev || !changed_dev || !changed_ptr_dev || n_roots == 0 || n_mask_words == 0 ||
      !new_dev->best_metric || !new_dev->best_entry || !new_dev->nexthop_mask || (n_records && !records_host))
    return HSPF_E_INVAL;
  if (n_records == 0) return HSPF_OK;
  // more records than the last hspf_routes_diff_device of this context compacted: the kernel would read list entries nobody wrote
  if (n_records > ctx->last_diff_count) { ctx->last_error = "hspf_routes_pack: n_records exceeds hspf_routes_diff_count()"; return HSPF_E_INVAL; }
  return guarded(ctx, [&]() -> int {
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    const size_t bytes = (size_t)n_records * (ROUTE_REC_WORDS + 2u * n_mask_words) * 4;
    int rc = ensure(ctx, ctx->pack, bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(k_routes_pack, dim3((n_records + 255u) / 256u), dim3(256), 0, s, n_records, n_roots, n_prefixes, n_mask_words,
                       changed_dev, changed_ptr_dev, action_dev, (const uint32_t *)new_dev->best_metric, (const uint32_t *)new_dev->best_entry,
                       (const uint64_t *)new_dev->nexthop_mask, (uint32_t *)ctx->pack.p);
    HIPCHK(ctx, hipMemcpyAsync(records_host, ctx->pack.p, bytes, hipMemcpyDeviceToHost, s));      // THE copy of the hand-off
    HIPCHK(ctx, hipStreamSynchronize(s));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { ctx->last_error = std::string("k_routes_pack: ") + hipGetErrorString(le); return HSPF_E_HIP; }
    return HSPF_OK;
  });
}

int hspf_ancestors_device(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                          const uint32_t *dist_dev, const uint16_t *hops_dev, const uint16_t *flags_dev,
                          uint32_t level, uint32_t n_words, uint32_t *level_rank_dev, uint32_t *level_count_dev,
                          uint64_t *anc_dev) {
  if (!ctx || !g || !roots || !dist_dev || !hops_dev || !flags_dev || !level_count_dev || !anc_dev || n_roots == 0 ||
      n_words == 0 || level == 0 || level > 0xFFFFu)
    return HSPF_E_INVAL;
  return guarded(ctx, [&]() -> int {
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    ctx->prefill.valid = false;                // ctx->changed serves as flag scratch below: what a previous run prefilled is gone
    const uint32_t n = g->n, nb = (n + 255) / 256;
    for (uint32_t r = 0; r < n_roots; ++r)
      if (roots[r] != HSPF_NO_ROOT && roots[r] >= n) { ctx->last_error = "root out of range"; return HSPF_E_INVAL; }
    int rc;
    // scratch: roots [n_roots] | per-block counts / offsets [n_roots][nb]
    if ((rc = ensure(ctx, ctx->gb_delta, ((size_t)n_roots + (size_t)n_roots * nb) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->changed, (size_t)CHANGED_CAP * 4))) return rc;
    uint32_t *d_roots = (uint32_t *)ctx->gb_delta.p, *d_blk = d_roots + n_roots;
    int *d_changed = (int *)ctx->changed.p;
    HIPCHK(ctx, hipMemcpyAsync(d_roots, roots, (size_t)n_roots * 4, hipMemcpyHostToDevice, s));
    const GraphDev gd = g->dev();
    for (uint32_t r0 = 0; r0 < n_roots; r0 += 65535u) {            // gridDim.y
      const uint32_t nr = std::min(65535u, n_roots - r0);
      const size_t o = (size_t)r0 * n;
      hipLaunchKernelGGL(k_anc_count, dim3(nb, nr), dim3(256), 0, s, gd, d_roots + r0, hops_dev + o, flags_dev + o, level, d_blk + (size_t)r0 * nb);
    }
    hipLaunchKernelGGL(k_anc_scan, dim3((n_roots + 63) / 64), dim3(64), 0, s, n_roots, nb, d_roots, flags_dev, n, d_blk, level_count_dev);
    std::vector<uint32_t> cnt(n_roots);
    HIPCHK(ctx, hipMemcpyAsync(cnt.data(), level_count_dev, (size_t)n_roots * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    for (uint32_t r = 0; r < n_roots; ++r)
      if (cnt[r] != 0xFFFFFFFFu && cnt[r] > 64u * n_words) { ctx->last_error = "hspf_ancestors_device: need " + std::to_string((cnt[r] + 63) / 64) + " words"; return HSPF_E_TOO_MANY_SLOTS; }
    for (uint32_t r0 = 0; r0 < n_roots; r0 += 65535u) {
      const uint32_t nr = std::min(65535u, n_roots - r0);
      const size_t o = (size_t)r0 * n;
      hipLaunchKernelGGL(k_anc_init, dim3(nb, nr), dim3(256), 0, s, gd, d_roots + r0, hops_dev + o, flags_dev + o, level, n_words,
                         d_blk + (size_t)r0 * nb, level_rank_dev ? level_rank_dev + o : nullptr, anc_dev + o * n_words);
    }
    const uint32_t ignore_ovl = (run_flags & HSPF_RUN_IGNORE_OVERLOAD) ? 1u : 0u, hc = g->hopcount_like ? 1u : 0u;
    // the fixed point of a monotone OR: chunks of sweeps launched ahead, one flag read-back per chunk
    uint32_t sweep = 0;
    for (;;) {
      const uint32_t chunk = 8;
      if (sweep + chunk > CHANGED_CAP) { ctx->last_error = "ancestor sweeps did not converge"; return HSPF_E_INTERNAL; }
      HIPCHK(ctx, hipMemsetAsync(d_changed + sweep, 0, (size_t)chunk * 4, s));
      for (uint32_t i = 0; i < chunk; ++i)
        for (uint32_t r0 = 0; r0 < n_roots; r0 += 65535u) {
          const uint32_t nr = std::min(65535u, n_roots - r0);
          const size_t o = (size_t)r0 * n;
          hipLaunchKernelGGL(k_anc_sweep, dim3(nb, nr), dim3(256), 0, s, gd, d_roots + r0, dist_dev + o, hops_dev + o, flags_dev + o,
                             level, n_words, ignore_ovl, hc, anc_dev + o * n_words, d_changed, (int)(sweep + i));
        }
      sweep += chunk;
      int last = 0;
      HIPCHK(ctx, hipMemcpyAsync(&last, d_changed + sweep - 1, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(ctx, hipStreamSynchronize(s));
      if (!last) break;
    }
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { ctx->last_error = std::string("k_anc: ") + hipGetErrorString(le); return HSPF_E_HIP; }
    return HSPF_OK;
  });
}

}  // extern "C"

#include "spf_multi.hip.h"
