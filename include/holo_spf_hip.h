/*
 * holo_spf_hip.h — C ABI of libholo_spf_hip.so, the MI355X (gfx950) SPF engine.
 *
 * This is the drop-in boundary for the link-state SPF hot path of holo-routing/holo.
 * The reference has no FFI for this path; the boundary is drawn at
 * the two functions that do all shortest-path-tree work:
 *
 *   holo-ospf/src/spf.rs:587-729   run_area<V>()     one SPT per area, root = self
 *   holo-isis/src/spf.rs:527-709   compute_spt()     one SPT per (level, MT, arbitrary root)
 *   holo-isis/src/flooding/manet.rs:47-69            the batched-roots caller (one SPT per adjacency)
 *
 * What crosses the boundary is the *graph* those loops walk (CSR restatement of
 * `vertex_lsa_links` holo-ospf/src/ospfv2/spf.rs:389-460, ospfv3/spf.rs:348-419 and
 * `vertex_edges` holo-isis/src/spf.rs:1013-1128) and, per root, the per-vertex result the
 * loops produce (`Vertex{distance,hops,nexthops}` holo-ospf/src/spf.rs:38-46,
 * holo-isis/src/spf.rs:78-88).  Everything that needs Interface / Adjacency / LSA objects
 * (next-hop address resolution, prefix attachment) stays with the caller and consumes
 * dist / hops / first_hop_mask.  See INTEGRATION.md for the Rust-side binding.
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no C++ or torch types;
 *   - every entry point returns 0 (HSPF_OK) or a negative HSPF_E* code; nothing aborts or
 *     throws across the boundary (mirrors `Error::SpfRootNotFound(..).log(); return;`
 *     holo-ospf/src/spf.rs:605-610 — the caller logs and keeps its own loop as fallback);
 *   - a ctx is used by one thread at a time; distinct ctxs may run concurrently (one HIP
 *     stream each), matching the one-OS-thread-per-instance contract of
 *     holo-protocol/src/lib.rs:427-430;
 *   - there is NO CPU fallback inside the library: without a HIP device hspf_init fails.
 */
#ifndef HOLO_SPF_HIP_H
#define HOLO_SPF_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSPF_ABI_VERSION 8u   /* 8: hspf_stats + n_repaired_roots / repair_* (dynamic pop order resolved in parallel); HSPF_RF_EXACT = "the pop order is dynamic";
                                 7: + packed results (hspf_run_packed / _device / _async, hspf_wait_packed, hspf_packed_layout + decode helpers),
                                    hspf_host_alloc / hspf_host_free, hspf_device_alloc / _free / _to_host / hspf_host_to_device, hspf_rib_clear_device / hspf_rib_fold_device (several areas, one RIB, on the device), HSPF_E_NO_PACKED (additions only); the library no longer sets
                                    GPU_MAX_HW_QUEUES at load time;
                                 6: + hspf_run_device_async / hspf_wait / hspf_wait_all / hspf_async_lanes, hspf_multi_run_async / hspf_multi_run_wait, hspf_recommend_cpu (additions only);
                                 5: + hspf_routes_diff_count / hspf_routes_pack, hspf_multi_init_error, HSPF_PFX_RESIDENT, HSPF_GX_ELL_* / LEAF / SUMMARY */

/* ---- error codes ------------------------------------------------------------------- */
#define HSPF_OK                 0
#define HSPF_E_INVAL           -1   /* bad argument (NULL, out-of-range root, malformed CSR)   */
#define HSPF_E_NODEV           -2   /* no usable HIP device / device ordinal out of range     */
#define HSPF_E_HIP             -3   /* a HIP runtime call failed; see hspf_last_error(ctx)     */
#define HSPF_E_NOMEM           -4   /* host or device allocation failed                        */
#define HSPF_E_TOO_MANY_SLOTS  -5   /* a root has more first-hop slots than n_mask_words*64    */
#define HSPF_E_INTERNAL        -6   /* invariant violated (never expected)                     */
#define HSPF_E_NO_PACKED       -7   /* hspf_run_packed*: this run's results do not fit packed words
                                       (more than 24 first-hop slots, hop counts beyond the hop field);
                                       nothing was written, call hspf_run / hspf_run_device instead */

/* ---- vertex flags (hspf_csr.vflags) -------------------------------------------------- */
/* bit0: vertex is an OSPF network vertex / IS-IS pseudonode.  Reaching it does not
 *       increment `hops` (holo-ospf/src/spf.rs:675-678, holo-isis/src/spf.rs:650-653).      */
#define HSPF_VF_NETWORK    0x01u
/* bit1: IS-IS overload bit set for the topology of this run: the vertex stays in the SPT
 *       but its links are skipped unless it is the root (holo-isis/src/spf.rs:568-574).
 *       Routers only, as in the reference (`!vertex.id.is_pseudonode()`): ignored on a vertex
 *       that also has HSPF_VF_NETWORK.                                                          */
#define HSPF_VF_NO_TRANSIT 0x02u
/* bit2: never expanded, root included: zeroth LSP missing / seqno 0 / lifetime 0
 *       (holo-isis/src/spf.rs:558-561) or protocols-supported gate (:582-604).               */
#define HSPF_VF_NO_EXPAND  0x04u

/* ---- run flags (hspf_run*.run_flags) -------------------------------------------------- */
/* OSPF semantics for first hops: a network reached from a hops==0 parent gets its own
 * first-hop slot (holo-ospf/src/ospfv2/spf.rs:296-302).  Without it (IS-IS) a pseudonode
 * reached from a hops==0 parent gets no next hop (holo-isis/src/spf.rs:680-701).            */
#define HSPF_RUN_NET_NEXTHOPS     0x01u
/* IS-IS flooding-topology run (mt_id == None): overload bit is not applied
 * (holo-isis/src/spf.rs:566-574).                                                           */
#define HSPF_RUN_IGNORE_OVERLOAD  0x02u
/* Force the sequential exact kernel for every root (test hook).                             */
#define HSPF_RUN_FORCE_EXACT      0x04u
/* Also produce hspf_result.pop_rank.                                                        */
#define HSPF_RUN_POP_RANK         0x08u
/* Count the rows the fused fixed point evaluates (hspf_stats.rows_recomputed).  A diagnostic: the counting
 * instantiation of the kernel is a few per cent slower, the results are the same.                           */
#define HSPF_RUN_COUNT_ROWS       0x10u

/* ---- per-(root,vertex) result flags (hspf_result.vflags_out) --------------------------- */
#define HSPF_RF_IN_SPT   0x0001u    /* vertex was popped into the SPT                         */
#define HSPF_RF_EXACT    0x0002u    /* the root's pop order is NOT the static (distance, index) order (zero-cost links, saturation):
                                       rows from k_repair or the sequential kernel; ask for pop_rank when the order matters */

#define HSPF_DIST_INF    0xFFFFFFFFu /* dist of a vertex that is not in the SPT               */
#define HSPF_NO_ROOT     0xFFFFFFFFu /* padding entry in a roots[] array: produces an empty SPT */

typedef struct hspf_ctx   hspf_ctx;    /* one per protocol-instance thread: device, stream, scratch */
typedef struct hspf_graph hspf_graph;  /* device-resident graph of one LSDB generation              */

/*
 * The LSDB graph of one area (OSPF) or one level x topology (IS-IS), as forward CSR.
 *   - vertex index == rank in the reference's VertexId order, so that integer compare is
 *     the candidate-list tie-break (holo-ospf/src/ospfv2/spf.rs:41-45: networks before
 *     routers, then numeric; holo-isis/src/spf.rs:96-100: pseudonodes first, then the
 *     7-byte LAN id);
 *   - row u lists the links of u in LSA / LSP order exactly as the reference iterates them
 *     (incl. parallel links and the TLV2+TLV22 duplicates of metric-type "both"); links whose
 *     target LSA/LSP is absent are simply not listed;
 *   - the two-way connectivity check (holo-ospf/src/spf.rs:654-664,
 *     holo-isis/src/spf.rs:616-627) is applied by the library, not the caller;
 *   - arrays are caller-owned host memory, borrowed for the duration of the call only.
 */
typedef struct {
  uint32_t        n_vertices;       /* <= 2^24                                                */
  uint32_t        n_edges;
  const uint32_t *row_ptr;          /* [n_vertices+1], row_ptr[0]==0, non-decreasing          */
  const uint32_t *col;              /* [n_edges] target vertex index                          */
  const uint32_t *metric;           /* [n_edges] link cost (OSPF u16, IS-IS u8/u24, widened)  */
  const uint8_t  *vflags;           /* [n_vertices] HSPF_VF_*                                 */
  uint32_t        max_path_metric;  /* relaxations with dist > this are dropped: 0xFFFFFFFF
                                       for OSPF (saturating add, holo-ospf/src/spf.rs:672),
                                       1023 / 0xFE000000 for IS-IS (holo-isis/src/spf.rs:44-49,
                                       637-647)                                               */
} hspf_csr;

/*
 * Per-root results, row-major [n_roots][n_vertices].  With hspf_run() the pointers are
 * caller-owned HOST memory; with hspf_run_device() they are DEVICE (HBM) pointers and the
 * results never leave the GPU (for device-side consumers and the RCCL all-gather).
 * Any pointer may be NULL to skip that output (dist must not be NULL).
 *
 * first_hop_mask: bit k of word k/64 == first-hop slot k of that root is one of the vertex's
 * next hops.  Slots are numbered per root over the out-links of the vertices that can have
 * hops == 0:  H = [root] ++ (network vertices reachable from the root through network
 * vertices only, breadth-first, links in row order, each vertex once);
 * slot(p, j) = sum(deg(H[i]) for H[i] before p) + j   for the j-th link of row p
 * (positions count ALL links of the row as passed in hspf_csr).  hspf_slot_table() returns
 * H and the bases so the caller can map a slot back to (parent vertex, link) and call its
 * own `calc_nexthops` / `resolve_nexthop` on it.
 */
typedef struct {
  uint32_t *dist;            /* HSPF_DIST_INF when not in SPT                                 */
  uint16_t *hops;            /* holo `Vertex.hops` (first-discoverer rule)                    */
  uint16_t *vflags_out;      /* HSPF_RF_*                                                     */
  uint64_t *first_hop_mask;  /* [n_roots][n_vertices][n_mask_words]                           */
  uint32_t  n_mask_words;    /* capacity in u64 words per (root,vertex); >= hspf_mask_words() */
  uint32_t *pop_rank;        /* position in the reference's pop order (needs HSPF_RUN_POP_RANK),
                                0xFFFFFFFF when not in SPT                                     */
} hspf_result;

/* Timing / work counters of the last hspf_run*() on a ctx (HIP-event timed, on the ctx stream). */
typedef struct {
  uint32_t n_roots;
  uint32_t n_batches;          /* 64-root wavefront batches                                   */
  uint32_t n_relax_launches;   /* launches of the fused sweep / distance relaxation kernel    */
  uint32_t n_dag_launches;     /* launches of the SPT-DAG (hops / first-hop) kernel           */
  uint32_t n_exact_roots;      /* roots that needed the sequential exact kernel               */
  uint32_t n_mask_words;
  float    ms_total;           /* first launch -> results in place (device time)              */
  float    ms_relax;
  float    ms_dag;
  float    ms_finish;          /* transpose to row-major outputs (+ exact kernel)             */
  float    ms_d2h;             /* only for hspf_run(): device->host copies                    */
  uint32_t state_bytes;        /* per-(vertex,root) state of the fused path: 4 or 8; 0 = two-phase path */
  uint32_t narrow_overflow;    /* 1: the 4-byte state overflowed and the run was redone with 8 bytes   */
  uint64_t rows_recomputed;    /* HSPF_RUN_COUNT_ROWS: (vertex, 64-root batch) rows the fused fixed point evaluated, summed
                                  over its launches (0 without the flag); the reference settles each vertex once per root
                                  (holo-isis/src/spf.rs:552-556), i.e. n_batches * n_vertices rows would be 1x */
  uint32_t single_wg;          /* 1: small graph, the run took the one-workgroup-per-root kernel (one launch);
                                  2: one to eight roots on a mid-size graph, one XCD per root (k_xcd, one launch):
                                  dbg[1] then holds its sweeps (bits 0-15; bit 31: a workgroup ran on another XCD) */
  uint32_t lane_vertex;        /* 1: a few roots on a larger graph, the run took the lane = vertex kernel (k_lv)   */
  uint32_t dbg[4];             /* [0]: 1 = the run took the lean sweep (k_fused_lean); [1]: lean sweep: bits 0-7 = dense passes that did
                                  work, 8-15 = head sweeps that ran, 16-23 = dense passes planned, 24-30 = head sweeps planned
                                  (the plan is sized from the previous run, the launches decide on the device);
                                  bit 31 = a wide-mask run left the graph's leaves to the emit;
                                  [2], [3]: host microseconds from the entry of the (last) run to its first enqueue /
                                  to its return.  HSPF_RUN_COUNT_ROWS on the one-workgroup path instead: sweeps,
                                  shader cycles, 100 MHz wall ticks and set-up cycles of the first root's workgroup  */
  /* ABI 8: roots whose pop order is dynamic (zero-cost router links) — distances from the sweep kernels, hops and first-hop
     masks recomputed in the true pop order by k_repair, one workgroup per root (holo_amd/csrc/spf_repair.hip.h) */
  uint32_t n_repaired_roots;   /* roots k_repair put right (n_exact_roots counts only what still needed the sequential kernel) */
  uint32_t repair_sweeps;      /* worklist sweeps of the slowest of them                               */
  uint32_t repair_evals;       /* (root, vertex) evaluations in the true order, all of them together    */
  uint32_t repair_groups;      /* groups of vertices released by a higher-numbered one that were walked  */
  float    ms_repair;          /* host time of the repair step (enqueue to completion)                   */
} hspf_stats;

/* ---- lifecycle -------------------------------------------------------------------------- */
uint32_t    hspf_abi_version(void);
int         hspf_device_count(void);                       /* >=0, or HSPF_E_*                 */
int         hspf_init(int device_ordinal, hspf_ctx **out);
void        hspf_shutdown(hspf_ctx *ctx);
const char *hspf_strerror(int code);
const char *hspf_last_error(const hspf_ctx *ctx);          /* detail of the last failure       */
/* Where the GPU path pays (no GPU needed; pure arithmetic): 1 when the caller's own CPU loop is expected to finish
 * these runs sooner than the engine can, 0 otherwise.  One root on a small LSDB is the reference's everyday call
 * (run_area / compute_spt, one root per area / level: holo-ospf/src/spf.rs:540-542, holo-isis/src/spf.rs:746-761);
 * the engine's floor for any run is one launch + one stream synchronisation (measured wall on MI355X, one root on 4-neighbour
 * grids: 0.033 ms at 25-64 vertices, 0.036 at 100, 0.039 at 144-256, 0.046 at 400, 0.050 at 500: ~0.032 + 3.5e-5 n), the
 * reference-shaped loop (ordered map, linear candidate scan, per-edge two-way rescan) costs
 * ~(2.4e-4 n + 7e-7 n^2) ms per root on one host core (n = 100: 0.03, 196: 0.09, 484: 0.29 ms; tools/cpu_small_graph_table.py),
 * so: CPU iff n_roots * (2.4e-4 n + 7e-7 n^2) < 0.032 + 3.5e-5 n — one root below ~110 vertices, two below ~60, never from
 * 8 roots up.  n_edges is taken for callers whose links-per-router ratio is far from the grid-like 4-8 the table was measured on
 * (the per-root cost is scaled by max(1, n_edges / (8 n))).  INTEGRATION.md section 6 shows the call site. */
int hspf_recommend_cpu(uint32_t n_vertices, uint32_t n_edges, uint32_t n_roots);
/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the ctx's own. */
int         hspf_set_stream(hspf_ctx *ctx, void *hip_stream);
void       *hspf_get_stream(const hspf_ctx *ctx);

/* ---- graph ------------------------------------------------------------------------------ */
int      hspf_graph_upload(hspf_ctx *ctx, const hspf_csr *csr, hspf_graph **out);

/* The same from LSDB records (ABI 8; SURVEY.md §8f-1: "LSDB -> CSR extraction"): vertices by 64-bit KEY in any order, links as
 * (target key, cost) in LSA / LSP link order, targets unresolved.  Ascending key order must be the reference's VertexId order
 * (IS-IS: !pseudonode << 56 | the 7 LAN-id bytes, big-endian — holo-isis/src/spf.rs:96-100; OSPFv2: router << 32 | id —
 * holo-ospf/src/ospfv2/spf.rs:41-45).  The device ranks the keys, resolves every link's target, drops links whose target is
 * not a vertex of the LSDB (vertex_lsa_links / vertex_edges do not yield them) and builds the graph; what it built can be
 * read back with hspf_graph_export (ROW_PTR / COL / METRIC / VFLAGS: the caller's CSR as hspf_graph_upload would have got it).
 * rank_out (may be NULL): [n_vertices] index of input vertex i in the graph.  Duplicate keys: HSPF_E_INVAL. */
typedef struct {
  uint32_t n_vertices, n_links;
  const uint64_t *vertex_key;   /* [n_vertices]                                                   */
  const uint32_t *row_ptr;      /* [n_vertices+1] rows of the vertices in the order of vertex_key  */
  const uint64_t *target_key;   /* [n_links]                                                      */
  const uint32_t *metric;       /* [n_links]                                                      */
  const uint8_t  *vflags;       /* [n_vertices] HSPF_VF_*                                         */
  uint32_t max_path_metric;
} hspf_keyed_lsdb;
int      hspf_graph_upload_keyed(hspf_ctx *ctx, const hspf_keyed_lsdb *lsdb, hspf_graph **out, uint32_t *rank_out);
void     hspf_graph_free(hspf_ctx *ctx, hspf_graph *g);
uint32_t hspf_graph_n_vertices(const hspf_graph *g);
uint32_t hspf_graph_n_edges(const hspf_graph *g);          /* links of the caller's CSR (after patches) */
uint32_t hspf_graph_n_edges_kept(const hspf_graph *g);     /* links surviving the two-way check */

/*
 * Incremental update.  An LSP / LSA that is re-originated, purged or aged out changes
 * exactly the links OUT OF its own vertex and that vertex's gates — holo-isis keys a partial run by the
 * changed LSPs (`trigger_lsps`, holo-isis/src/spf.rs:144,735), holo-ospf by the changed LSAs
 * (`SpfTriggerLsa`, holo-ospf/src/spf.rs:120-139) — so the unit of change is "replace whole rows":
 * for each listed vertex the new row (links in LSA order, as in hspf_csr) and its new flags.  The vertex set
 * is fixed; a new or vanished vertex needs a fresh hspf_graph_upload.  After the call the graph is
 * indistinguishable from one uploaded from the patched CSR (two-way check, kept links and the whole device
 * layout are rebuilt on the device from the resident CSR; only the replaced rows cross the bus).
 * On HSPF_E_INVAL nothing has changed; after HSPF_E_HIP / HSPF_E_NOMEM the graph must be freed.
 */
typedef struct {
  uint32_t        n_changed;
  const uint32_t *vertex;     /* [n_changed]   strictly ascending vertex indices                   */
  const uint32_t *row_ptr;    /* [n_changed+1] bounds of the replacement rows inside col / metric  */
  const uint32_t *col;        /* [row_ptr[n_changed]]                                              */
  const uint32_t *metric;     /* [row_ptr[n_changed]]                                              */
  const uint8_t  *vflags;     /* [n_changed]   HSPF_VF_* of those vertices after the change        */
} hspf_rows;
int hspf_graph_patch(hspf_ctx *ctx, hspf_graph *g, const hspf_rows *rows);

/* Copies one array of the device-resident graph back to the host (inspection, tests, debugging).
 * dst == NULL: only *out_bytes is set.  The layout: kept links = two-way and source expandable; in-rows
 * (links INTO a vertex) ordered by (cost descending, source ascending, position in the source row
 * ascending), IN_SRC carries the source's HSPF_VF_NO_TRANSIT in bit 31 and whether it is a leaf (HSPF_GX_LEAF) in
 * bit 30; out-rows in the caller's order. */
#define HSPF_GX_ROW_PTR   0u   /* u32 [n+1]    caller's CSR as resident on the device             */
#define HSPF_GX_COL       1u   /* u32 [e]                                                         */
#define HSPF_GX_METRIC    2u   /* u32 [e]                                                         */
#define HSPF_GX_VFLAGS    3u   /* u8  [n]                                                         */
#define HSPF_GX_IN_PTR    4u   /* u32 [n+1]                                                       */
#define HSPF_GX_IN_SRC    5u   /* u32 [kept]                                                      */
#define HSPF_GX_IN_COST   6u   /* u32 [kept]                                                      */
#define HSPF_GX_IN_POS    7u   /* u32 [kept]   position of the link inside its source row         */
#define HSPF_GX_OUT_PTR   8u   /* u32 [n+1]                                                       */
#define HSPF_GX_OUT_DST   9u   /* u32 [kept]                                                      */
#define HSPF_GX_OUT_COST 10u   /* u32 [kept]                                                      */
#define HSPF_GX_OUT_POS  11u   /* u32 [kept]                                                      */
#define HSPF_GX_ROWFLAGS 12u   /* u8  [n]      internal per-row flags of the fused sweep          */
#define HSPF_GX_TWOWAY   13u   /* u8  [e]      1 = the target's row lists the source              */
#define HSPF_GX_UNITS    14u   /* u32 [...]    work units of the sweep kernels: empty when no 16-vertex chunk holds a row
                                  of more than 32 in-links; else [4 units per heavy chunk | 1 unit per other chunk],
                                  each class in vertex order, entry = first vertex (| 0x80000000: one row per wave) */
#define HSPF_GX_BUILD_MODE 15u /* u32 [1]      how the last upload / patch derived the layout: 0 = per-link row scans,
                                  1 = hub mode (a row of more than 512 links: two device-wide sorts, O(log degree) per
                                  link), 2 = the last patch changed costs only (same targets, order and flags in every
                                  replaced row): the affected rows were re-ranked in place, nothing was rebuilt;
                                  3 = the last patch was structural and only the affected rows (the replaced ones, their
                                  old and new targets) were re-derived, the compact arrays behind them shifted.  The
                                  layout itself does not depend on the mode                                          */
#define HSPF_GX_ELL_SRC  16u   /* u32 [16(n+1)] fixed-stride copy of the in-rows of at most 16 links: source << 8, the low
                                  byte of a row's first entry = in-degree (0x1F: more than 16) | more than 16 out-links
                                  << 5 | network << 7; unused entries name the pad row n                            */
#define HSPF_GX_ELL_COST 17u   /* u32 [16(n+1)] their costs (unused entries 0)                                       */
#define HSPF_GX_ELL_OUT  18u   /* u32 [16(n+1)] out-neighbour j << 2 (unused entries 0xFFFFFFFF)                     */
#define HSPF_GX_LEAF     20u   /* u8  [n]      1 = leaf: exactly one kept in-link, and the kept out-links (at most one) lead
                                  back to its source.  IN_SRC carries the source's leaf bit in bit 30                */
#define HSPF_GX_SUMMARY  19u   /* u32 [12]     what a build derives from the links and a patch must keep current:
                                  largest kept cost, hop-count shape (0/1), smallest-graph kernel allowed (0/1), OR of
                                  the row flags, rows with a zero-cost link from a higher-numbered source, rows off the
                                  hop-count shape, largest in-degree, kept links; of the caller's rows (a structural
                                  patch updates them from the replaced rows alone): longest row, network vertices,
                                  links in rows of more than 32, "a quarter of the links sit in such rows" (0/1)      */
#define HSPF_GX_HOST_ROW_PTR 21u /* u32 [n+1]  the HOST mirror of the caller's row bounds (slot tables are made from it; a
                                  structural patch splices it in place): equals ROW_PTR                              */
#define HSPF_GX_HOST_COL 22u   /* u32 [e]      ... and of the caller's targets: equals COL                           */
#define HSPF_GX_ZCYC     23u   /* u8  [n]      (ABI 8) 1 = the vertex may lie on a cycle of zero-cost kept links: what survives 8
                                  rounds of "keeps a zero-cost in-link from and a zero-cost out-link to a survivor".  0 bytes
                                  when no row has a zero-cost link from a higher- or equal-numbered source (array unused)  */
int hspf_graph_export(hspf_ctx *ctx, const hspf_graph *g, uint32_t which, void *dst, size_t cap_bytes,
                      size_t *out_bytes);

/* ---- first-hop slots -------------------------------------------------------------------- */
/* Number of u64 mask words needed for these roots on this graph (>= 1). */
int hspf_mask_words(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots,
                    uint32_t *out_words);
/* Slot table of one root: writes up to `cap` entries of H (vertex index) and its slot base;
 * returns the number of entries of H (may exceed cap), or HSPF_E_*. */
int hspf_slot_table(hspf_ctx *ctx, const hspf_graph *g, uint32_t root,
                    uint32_t *h_vertex, uint32_t *h_base, uint32_t cap, uint32_t *out_total_slots);

/* ---- run -------------------------------------------------------------------------------- */
/* Synchronous: on return the results are in the caller's host buffers. */
int hspf_run(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots,
             uint32_t run_flags, hspf_result *out);
/* Results are written to device buffers; returns after the work has been *enqueued and
 * completed* on the ctx stream (the call synchronises the stream once at the end so that
 * error flags can be read).  roots is a host array. */
int hspf_run_device(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots,
                    uint32_t run_flags, hspf_result *out_device);
int hspf_get_stats(const hspf_ctx *ctx, hspf_stats *out);

/* ---- asynchronous runs (ABI 6) ----------------------------------------------------------- */
/*
 * The reference runs ONE SPF at a time per instance thread (holo-protocol/src/lib.rs:427-430), but an SPF event holds
 * several independent runs: one per area (holo-ospf/src/spf.rs:540-542), per level and topology
 * (holo-isis/src/spf.rs:746-761), per neighbour (holo-isis/src/flooding/manet.rs:59-69).  hspf_run_device_async
 * hands a run to one of the context's LANES — private engine contexts on the same device, each with its own stream,
 * scratch and host thread (HSPF_ASYNC_LANES, default 3) — and returns a ticket at once; hspf_wait blocks until that run
 * is over and returns its code (and statistics).  Lanes are taken in ticket order (ticket % lanes), each runs its tickets
 * one after the other from a short queue: the call blocks only while three tickets of that lane are already waiting.  The roots are copied; `out_device` buffers must stay
 * valid, and must not be shared between runs in flight, until the ticket has been waited for.  A run alone leaves most
 * of the chip idle during its sparse first and last sweeps (chains of small dependent launches); runs in flight on
 * several lanes move in lockstep and interleave those chains (isis-100k, 64-root runs: 150 k runs/s with three in flight
 * against 124 k one after the other).  Results are bit-identical to hspf_run_device's.  hspf_graph_patch /
 * hspf_graph_free on the context wait for its runs in flight first; results of a ticket are kept until eight later runs
 * of its lane have finished.  Same threading contract as every other call: one caller thread per context.
 * The lanes are also where a SYNCHRONOUS run of more than 64 roots goes when its roots fall into different state classes
 * (a few roots with many first-hop slots among many with few: each class is a run of its own): all classes but one are
 * handed to lanes and run side by side (fat-tree k=100, 51 switch roots + 50 host roots: 1.69 ms instead of 2.01).  The
 * lanes are created by the first call that needs them (~15 ms once per context).  In the same calls a LEAF root — a router
 * whose only two-way link leads to a router that is a root of the same call — is not run at all: its rows are its
 * neighbour's, one link further (distance + link cost up to the max-path metric, hops + 1, its one first-hop slot).
 */
int hspf_run_device_async(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots,
                          uint32_t run_flags, const hspf_result *out_device, uint64_t *ticket);
int hspf_wait(hspf_ctx *ctx, uint64_t ticket, hspf_stats *stats /* may be NULL */);
int hspf_wait_all(hspf_ctx *ctx);                  /* every run in flight is over (codes: hspf_wait)        */
uint32_t hspf_async_lanes(const hspf_ctx *ctx);    /* how many runs can be in flight                        */

/* ---- packed results (ABI 7) ---------------------------------------------------------------- */
/*
 * hspf_run delivers 16 bytes per (root, vertex) — 102 MB for 64 roots on a 100 000-vertex level, 1.8 ms over PCIe against
 * 0.45 ms of compute — although the engine holds the whole answer in ONE machine word per (root, vertex):
 * [distance | hop count | first-hop mask], 4 bytes when the fields fit (the usual case: checked on the device, never
 * assumed), else 8.  hspf_run_packed hands exactly those words over, row-major [n_roots][n_vertices], together with the
 * field positions of THIS run; the caller (whose next step looks at a vertex once: rebuilding `Vertex{distance, hops,
 * nexthops}`, holo-isis/src/spf.rs:78-88, holo-ospf/src/spf.rs:38-46) decodes a word where it needs it:
 *     in SPT     word <  not_reached                        (else: dist = HSPF_DIST_INF, hops = 0, mask = 0)
 *     distance   word >> dist_shift
 *     hops       (word >> hops_shift) & hops_mask
 *     mask       word & ((1 << mask_bits) - 1)              first-hop slots 0 .. mask_bits - 1, as hspf_result.first_hop_mask
 * (inline helpers below).  The values are bit for bit those of hspf_run.  What has no room in a word comes per root:
 * root_status[r] & HSPF_ROOT_EXACT = the root went through the sequential kernel (HSPF_RF_EXACT of hspf_run: its pop
 * order is not the static (distance, index) order; ask for pop_rank with hspf_run when the caller needs that order).
 * HSPF_RUN_POP_RANK is not accepted.  A run whose roots have more than 24 first-hop slots, or whose hop counts outgrow
 * the field, returns HSPF_E_NO_PACKED before / without writing anything: the caller then calls hspf_run.
 *
 * `words` must have room for 8 bytes per (root, vertex) — cap_bytes >= 8 * n_roots * n_vertices always suffices; only
 * layout->word_bytes * n_roots * n_vertices bytes are written (and cross the bus).  Host destinations: memory from
 * hspf_host_alloc (page-locked: the copy runs at bus speed) or any other host memory (the library then stages the copy
 * through its own page-locked blocks, chunk by chunk, at the speed of one host memcpy).
 */
typedef struct {
  uint32_t word_bytes;     /* 4 or 8                                                              */
  uint32_t dist_shift;
  uint32_t hops_shift;
  uint32_t hops_mask;
  uint32_t mask_bits;
  uint32_t reserved;       /* 0                                                                   */
  uint64_t not_reached;    /* word >= not_reached: the vertex is not in the SPT of that root      */
} hspf_packed_layout;
#define HSPF_ROOT_EXACT 0x01u

static inline uint64_t hspf_packed_word(const hspf_packed_layout *l, const void *words, size_t index) {
  return l->word_bytes == 4u ? (uint64_t)((const uint32_t *)words)[index] : ((const uint64_t *)words)[index];
}
static inline int      hspf_packed_in_spt(const hspf_packed_layout *l, uint64_t w) { return w < l->not_reached; }
static inline uint32_t hspf_packed_dist(const hspf_packed_layout *l, uint64_t w) { return w < l->not_reached ? (uint32_t)(w >> l->dist_shift) : HSPF_DIST_INF; }
static inline uint16_t hspf_packed_hops(const hspf_packed_layout *l, uint64_t w) { return w < l->not_reached ? (uint16_t)((w >> l->hops_shift) & l->hops_mask) : (uint16_t)0; }
static inline uint64_t hspf_packed_mask(const hspf_packed_layout *l, uint64_t w) { return w < l->not_reached ? (w & ((1ull << l->mask_bits) - 1ull)) : 0ull; }

/* Page-locked host memory for result buffers (hipHostMalloc / hipHostFree behind the boundary, so that the Rust side
 * needs no HIP binding of its own).  Valid on every context of the process. */
int  hspf_host_alloc(hspf_ctx *ctx, size_t bytes, void **out);
void hspf_host_free(hspf_ctx *ctx, void *p);

/* Device memory for callers that keep tables in HBM (hspf_run_device -> hspf_routes_device -> hspf_routes_diff_device)
 * without a HIP binding of their own (the Rust wrapper): hipMalloc / hipFree / a synchronous device-to-host copy on the
 * context's device. */
int  hspf_device_alloc(hspf_ctx *ctx, size_t bytes, void **out);
void hspf_device_free(hspf_ctx *ctx, void *p);
int  hspf_device_to_host(hspf_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int  hspf_host_to_device(hspf_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);

/* Synchronous, words in HOST memory (see above).  root_status: [n_roots] or NULL. */
int hspf_run_packed(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                    void *words_host, size_t cap_bytes, hspf_packed_layout *layout, uint8_t *root_status);
/* The same with `words_dev` in DEVICE memory (a quarter of the bytes of hspf_run_device's tables for consumers that stay
 * on the GPU or send the rows on with hspf_multi_allgather_rows); root_status is a HOST array. */
int hspf_run_packed_device(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                           void *words_dev, size_t cap_bytes, hspf_packed_layout *layout, uint8_t *root_status);
/* Asynchronous form of hspf_run_packed on the context's lanes (hspf_run_device_async): the run AND its copy to the host
 * belong to the ticket, so the copy of one batch crosses the bus while the next batch computes.  words_host /
 * root_status must stay valid, and must not be shared between tickets in flight, until hspf_wait_packed has returned
 * for the ticket; that call also fills `layout`.  hspf_wait works on such a ticket too (without the layout). */
int hspf_run_packed_async(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                          void *words_host, size_t cap_bytes, uint8_t *root_status, uint64_t *ticket);
int hspf_wait_packed(hspf_ctx *ctx, uint64_t ticket, hspf_packed_layout *layout, hspf_stats *stats /* may be NULL */);

/* ---- route derivation on device (the step right after the SPT) ------------------- */
/*
 * Prefix attachment of holo-isis compute_routes (holo-isis/src/spf.rs:864-918) for every root of a
 * previous hspf_run_device(): for each IP prefix, over the vertices that advertise it (iterated in
 * VertexId order, i.e. ascending vertex index), route metric = dist[v] + prefix metric; the smallest
 * wins (`Ordering::Less` replaces the route, :902-905), equal metrics merge their next hops
 * (`merge_nexthops`, :906-909).  What needs addresses — building Nexthop objects, the max-paths
 * truncation by address order (:920-929) — stays with the caller and consumes best_metric /
 * best_entry / nexthop_mask.
 *
 * The prefix table is CSR by prefix: entries pfx_ptr[p] .. pfx_ptr[p+1] are the (vertex, metric)
 * advertisements of prefix p, sorted by vertex index.  All result / input table pointers are DEVICE
 * pointers; pfx_* are caller-owned HOST arrays (uploaded per call).
 *
 * holo-ospf's update_rib_intra_area (holo-ospf/src/route.rs:343-448) is the same reduction with two twists,
 * selected by `flags`:
 *   HSPF_PFX_SATURATING   route metric = dist[v].saturating_add(metric)  (route.rs:362-366); the reached / not
 *                         reached distinction is then carried by best_entry alone (0xFFFFFFFF = no route).
 *   HSPF_PFX_LAST_MIN     the transit-network rule (route.rs:388-400): among equal metrics the LATER entry —
 *                         the larger LS-ID, entries being in ascending vertex = LS-ID order — REPLACES the
 *                         route instead of merging into it: best_entry = last entry attaining the minimum,
 *                         nexthop_mask = that entry's mask only.
 * An OSPF caller evaluates one table of Network-LSA prefixes with both flags and one table of Router-LSA stub
 * prefixes with HSPF_PFX_SATURATING, and folds the two per-prefix results in that order (networks sort before
 * routers in VertexId order): holo_amd/routes.py ospf_intra_area_device_routes.
 */
#define HSPF_PFX_SATURATING 0x1u
#define HSPF_PFX_LAST_MIN   0x2u
/*
 *   HSPF_PFX_ORDERED      the literal, ORDER-DEPENDENT fold of update_rib_intra_area (route.rs:343-448) for tables
 *                         whose entries do not come in vertex order: OSPFv3 reads its stub networks off the
 *                         Intra-Area-Prefix-LSAs in LSDB order (holo-ospf/src/ospfv3/spf.rs:421-478), router and
 *                         network vertices interleaved.  The entries of a prefix are listed in the reference's
 *                         iteration order; bit 31 of pfx_vertex (HSPF_PFX_ENTRY_NETWORK) marks an entry whose vertex
 *                         is a network, pfx_origin[e] is the Link State ID of that vertex's LSA (`stub.vertex.lsa
 *                         .origin().lsa_id`); metrics add saturating.  Per entry, as the reference: worse than the
 *                         route so far -> skipped (:371-375); a network entry meeting a route replaces it when shorter
 *                         or equal with a GREATER origin and is skipped otherwise (:388-400); then route_update
 *                         (:918-965): better replaces, equal merges the next hops.  init_* (optional, [n_prefixes],
 *                         the same for every root): the route an earlier area left in the RIB for that prefix
 *                         (the RIB is shared by the areas, route.rs:146-160) — init_exists[p] != 0, its metric and
 *                         origin; best_entry == HSPF_PFX_KEPT_INIT then says that this route still owns the prefix
 *                         and nexthop_mask holds what the table's entries merged INTO it.
 */
#define HSPF_PFX_ORDERED    0x4u
/*
 *   HSPF_PFX_RESIDENT     the caller's word that pfx_ptr / pfx_vertex / pfx_metric are exactly what its previous
 *                         hspf_routes_device call on this context passed (same pointers, same sizes, contents
 *                         untouched since — a prefix table changes with the LSDB, not with every SPF run): the range
 *                         checks and the host-to-device copies are skipped.  A table that does not match what was
 *                         recorded then (or an ordered one) is uploaded as usual.
 */
#define HSPF_PFX_RESIDENT   0x8u
#define HSPF_PFX_ENTRY_NETWORK 0x80000000u
#define HSPF_PFX_KEPT_INIT  0xFFFFFFFEu
typedef struct {
  uint32_t        n_prefixes;
  uint32_t        n_entries;
  const uint32_t *pfx_ptr;      /* [n_prefixes+1]                                                */
  const uint32_t *pfx_vertex;   /* [n_entries] advertising vertex (| HSPF_PFX_ENTRY_NETWORK with HSPF_PFX_ORDERED) */
  const uint32_t *pfx_metric;   /* [n_entries] advertised metric                                 */
  uint32_t        flags;        /* HSPF_PFX_*; 0 = the IS-IS rule                                */
  /* HSPF_PFX_ORDERED only (ABI 4; NULL otherwise): */
  const uint32_t *pfx_origin;   /* [n_entries]  Link State ID of the entry's vertex LSA          */
  const uint8_t  *init_exists;  /* [n_prefixes] or NULL                                          */
  const uint32_t *init_metric;  /* [n_prefixes] or NULL                                          */
  const uint32_t *init_origin;  /* [n_prefixes] or NULL                                          */
} hspf_prefix_table;

typedef struct {
  uint32_t *best_metric;        /* [n_roots][n_prefixes]  HSPF_DIST_INF: no advertising vertex in the SPT */
  uint32_t *best_entry;         /* [n_roots][n_prefixes]  index of the FIRST entry attaining it (route
                                   attributes — level, external flag, CONNECTED — come from that vertex,
                                   holo-isis/src/route.rs:79-107); 0xFFFFFFFF when unreachable    */
  uint64_t *nexthop_mask;       /* [n_roots][n_prefixes][n_mask_words] union of the first-hop masks of
                                   every entry attaining it                                        */
} hspf_routes;

int hspf_routes_device(hspf_ctx *ctx, uint32_t n_vertices, uint32_t n_roots, uint32_t n_mask_words,
                       const uint32_t *dist_dev, const uint16_t *flags_dev, const uint64_t *mask_dev,
                       const hspf_prefix_table *table, hspf_routes *out_dev);

/* ---- several areas, ONE RIB: the fold on device (ABI 7) ---------------------------------------
 * An OSPF instance computes one SPT per area and folds the areas' stub networks into ONE routing table, area after area
 * (the per-area loop holo-ospf/src/spf.rs:540-542; `update_rib_intra_area`, holo-ospf/src/route.rs:343-448, works on the
 * RIB the earlier areas left: better replaces, equal merges, the transit-network rule on the larger LS-ID).  hspf_rib_device
 * is that table on the device, over the INSTANCE-wide prefix list and an instance-wide first-hop slot numbering (area a's
 * slots start at mask word `word_offset` of its fold call: word-aligned, so placing an area's mask is a word copy);
 * hspf_rib_fold_device runs the literal ordered fold (HSPF_PFX_ORDERED) of ONE area's table with the state the earlier
 * areas left as its initial state and writes the new state back — nothing crosses the bus between areas.  After the last
 * area (best_metric, best_entry, nexthop_mask) IS an hspf_routes of one root over n_prefixes prefixes and n_mask_words
 * words: hspf_routes_diff_device / hspf_routes_pack compare it with the tables of the previous SPF event as for one area.
 *   best_entry   0xFFFFFFFF: no route; else area_index << 24 | entry index in that area's table (the route's owner:
 *                attributes that need LSA objects come from it on the host)
 *   origin       LS-ID of the owner's vertex LSA (the transit-network rule of the NEXT area's fold needs it)
 * All hspf_rib_device pointers are DEVICE pointers; `table` and `prefix_map` (area prefix -> instance prefix index, every
 * index at most once) are caller-owned HOST arrays.  table->flags must carry HSPF_PFX_ORDERED; init_* are ignored. */
typedef struct {
  uint32_t  n_prefixes;
  uint32_t  n_mask_words;
  uint32_t *best_metric;     /* [n_prefixes] */
  uint32_t *best_entry;      /* [n_prefixes] */
  uint64_t *nexthop_mask;    /* [n_prefixes][n_mask_words] */
  uint32_t *origin;          /* [n_prefixes] */
} hspf_rib_device;
int hspf_rib_clear_device(hspf_ctx *ctx, const hspf_rib_device *rib);           /* the empty RIB in front of the first area */
int hspf_rib_fold_device(hspf_ctx *ctx, uint32_t n_vertices, uint32_t area_mask_words,
                         const uint32_t *dist_dev, const uint16_t *flags_dev, const uint64_t *mask_dev,   /* one root's rows of the area's run */
                         const hspf_prefix_table *table, const uint32_t *prefix_map,
                         uint32_t area_index, uint32_t word_offset, const hspf_rib_device *rib);

/* ---- RIB diff on device (the first half of the wire step after the path) -------------------------
 * update_global_rib (holo-isis/src/route.rs:254-312, holo-ospf/src/route.rs:856-916) walks the new RIB, skips every route
 * whose metric and next hops are what the old RIB held, sends a RouteIpAdd for the rest and a RouteIpDel for what
 * vanished — per-route host work and messages that dominate once an LSDB carries 100 k+ prefixes.  On
 * the tables hspf_routes_device() writes, that comparison is "same best_metric and same nexthop_mask": this call does it
 * for every (root, prefix) of two such result sets in HBM and compacts the indices that need a message.
 * PRECONDITION (the caller's): both sets index the same prefix list, and a first-hop slot means the same next hop in both
 * runs (the rows of the root and of its hops-0 networks did not change) — otherwise compare on the host.
 *   action[r][p]   HSPF_DIFF_SAME      nothing to send (:268-277)
 *                  HSPF_DIFF_INSTALL   new or changed, and the new route has next hops: RouteIpAdd (:283-295)
 *                  HSPF_DIFF_WITHDRAW  the old route had next hops (was installed) and the prefix has no route any
 *                                      more: RouteIpDel (:303-310)
 *                  HSPF_DIFF_SILENT    changed, but nothing goes on the wire (a route without next hops: CONNECTED, or
 *                                      unresolved next hops)
 *   changed / changed_ptr   per root r the prefix indices with INSTALL or WITHDRAW, ascending (the reference's emission
 *                  order is the prefix order): changed[changed_ptr[r] .. changed_ptr[r+1]); changed has room for
 *                  n_roots * n_prefixes entries.  All pointers are DEVICE pointers.
 */
#define HSPF_DIFF_SAME     0u
#define HSPF_DIFF_INSTALL  1u
#define HSPF_DIFF_WITHDRAW 2u
#define HSPF_DIFF_SILENT   3u
int hspf_routes_diff_device(hspf_ctx *ctx, uint32_t n_roots, uint32_t n_prefixes, uint32_t n_mask_words,
                            const hspf_routes *old_dev, const hspf_routes *new_dev,
                            uint8_t *action_dev, uint32_t *changed_dev, uint32_t *changed_ptr_dev);

/* ---- the hand-off of the diff (second quarter) ----------------------------------------------------
 * route_install sends ONE message per route (holo-isis/src/ibus/tx.rs:35-110, holo-ospf/src/ibus/tx.rs:32-77 ->
 * holo-routing/src/rib.rs:92-135).  hspf_routes_pack turns the changed list of the last hspf_routes_diff_device into ONE
 * contiguous record stream in the reference's emission order and brings it to the host with ONE copy: what a single
 * batched RouteIpAdd / RouteIpDel message carries.  Record k (HSPF_ROUTE_REC_WORDS + 2 * n_mask_words u32 words):
 *     [0] root index   [1] prefix index   [2] action (HSPF_DIFF_INSTALL | HSPF_DIFF_WITHDRAW)   [3] metric of the NEW route
 *     [4] best_entry of the new route (0xFFFFFFFF: the prefix has no route any more)   [5] 0
 *     [6 ..] the new route's next-hop mask words, low half first
 * roots ascending, prefixes ascending inside a root.  The caller expands a record into its message: prefix text from its
 * own prefix list, next hops = the resolved first-hop slots of the mask (holo_amd.routes.expand_route_records is the twin
 * that reproduces the reference's recorded RouteIpAdd / RouteIpDel sequence from it).
 * hspf_routes_diff_count: number of records the last hspf_routes_diff_device left (it came back with that call's own
 * synchronisation), i.e. n_records and the size of records_host. */
#define HSPF_ROUTE_REC_WORDS 6u
uint32_t hspf_routes_diff_count(const hspf_ctx *ctx);
int hspf_routes_pack(hspf_ctx *ctx, uint32_t n_roots, uint32_t n_prefixes, uint32_t n_mask_words, const hspf_routes *new_dev,
                     const uint8_t *action_dev, const uint32_t *changed_dev, const uint32_t *changed_ptr_dev,
                     uint32_t n_records, uint32_t *records_host);

/* ---- ancestor sets on device (the queries of flooding::manet::reflood_list) ------------------------
 * For every root of a previous hspf_run_device() and a level L (1 = first hops / remote-neighbour list, 2 = second
 * hops): the root's level-L routers = router vertices of its SPT with hops == L, numbered in ascending vertex index
 * (level_rank, 0xFFFFFFFF elsewhere; level_count = how many), and anc[root][v] = the n_words x 64 bit set of the level-L
 * routers that are ancestors of v in the SPT's parent DAG, or v itself — holo-isis Spt::is_on_path(a, d)
 * (holo-isis/src/spf.rs:261-286) for a level-L router a is then bit level_rank[a] of anc[root][d].
 * roots / run_flags: as passed to the run.  dist / hops / flags / level_rank / level_count / anc are DEVICE pointers
 * ([n_roots][n_vertices] row-major, anc with n_words words per entry; level_rank may be NULL).
 * A root that needed the sequential exact kernel (dynamic pop order) gets level_count 0xFFFFFFFF and no sets: the
 * caller keeps its own walk for it.  Returns HSPF_E_TOO_MANY_SLOTS when a root has more than 64 x n_words level-L
 * routers (level_count is filled in either way: call again with enough words). */
int hspf_ancestors_device(hspf_ctx *ctx, const hspf_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t run_flags,
                          const uint32_t *dist_dev, const uint16_t *hops_dev, const uint16_t *flags_dev,
                          uint32_t level, uint32_t n_words, uint32_t *level_rank_dev, uint32_t *level_count_dev,
                          uint64_t *anc_dev);

/* ---- several GPUs --------------------------------------------------------------------------
 * SPF roots are independent units over a read-only graph: the graph is replicated on every GPU, whole 64-root
 * wavefront batches are dealt to the ranks (hspf_shard_bounds), every rank runs its slice, and ONE all-gather per
 * table gives every rank the rows of all roots.  The reference has no analogue — its only multi-root caller is the
 * sequential loop of holo-isis/src/flooding/manet.rs:47-69; what this replaces on the holo side is that loop and the
 * per-area fan-out of holo-ospf/src/spf.rs:540-542.
 *
 * A rank is one engine context on one device.  Two job shapes:
 *   single process  hspf_multi_config.unique_id == NULL, world == n_local: the process drives every device (one host
 *                   thread per device inside hspf_multi_run); the gather is direct device-to-device copies over
 *                   xGMI (every device pushes its slice to every other one: the all-to-all pattern a fully
 *                   connected xGMI mesh is built for).  device_ordinals may repeat an ordinal — several contexts and
 *                   streams on one GPU — which is how the path is tested on a one-GPU box.
 *   one process per GPU  (n_local = 1, world = number of processes): rank 0 creates a communicator id with
 *                   hspf_multi_unique_id, the HOST transports its 128 bytes to the other processes (holo would use its
 *                   ibus; bench.py uses torch.distributed), every process calls hspf_multi_init with it; the gather is
 *                   ncclAllGather / ncclBroadcast of RCCL (librccl.so, loaded at run time) on the contexts' streams.
 * Every call is collective over the ranks of the job and synchronous. */
typedef struct hspf_multi hspf_multi;
typedef struct hspf_multi_graph hspf_multi_graph;
#define HSPF_COMM_ID_BYTES 128
typedef struct {
  uint32_t       n_local;          /* devices driven by this process                                              */
  const int     *device_ordinals;  /* [n_local]                                                                   */
  uint32_t       world;            /* ranks of the job                                                            */
  uint32_t       first_rank;       /* rank of local device 0 (local device i is rank first_rank + i)              */
  const uint8_t *unique_id;        /* NULL, or HSPF_COMM_ID_BYTES bytes from hspf_multi_unique_id                 */
} hspf_multi_config;

int         hspf_multi_unique_id(uint8_t id[HSPF_COMM_ID_BYTES]);      /* needs librccl.so                        */
int         hspf_multi_init(const hspf_multi_config *cfg, hspf_multi **out);
/* Detail of the last failed hspf_multi_init on the calling thread (e.g. RCCL's own message); "" if none. */
const char *hspf_multi_init_error(void);
void        hspf_multi_shutdown(hspf_multi *m);
const char *hspf_multi_last_error(const hspf_multi *m);
hspf_ctx   *hspf_multi_ctx(hspf_multi *m, uint32_t local_index);      /* the rank's context, for the one-device calls */
uint32_t    hspf_multi_n_local(const hspf_multi *m);

int  hspf_multi_graph_upload(hspf_multi *m, const hspf_csr *csr, hspf_multi_graph **out);   /* a replica per local device */
int  hspf_multi_graph_patch(hspf_multi *m, hspf_multi_graph *g, const hspf_rows *rows);
void hspf_multi_graph_free(hspf_multi *m, hspf_multi_graph *g);
hspf_graph *hspf_multi_graph_local(hspf_multi_graph *g, uint32_t local_index);

/* [begin, end) of rank `rank` in a list of n_roots roots: whole 64-root batches dealt as evenly as possible, the
 * first n_batches % world ranks take one more, the ragged tail goes to the last rank that has work. */
void hspf_shard_bounds(uint32_t n_roots, uint32_t world, uint32_t rank, uint32_t *begin, uint32_t *end);

/* Areas first, then roots (BASELINE configs[3]): every area is its own graph with its own root list;
 * the (area, 64-root batch) units, in area order, are cut into `world` contiguous runs of equal batch counts, so a
 * rank touches as few areas as possible (it uploads only those) and no batch is split.  Writes one slice per (rank,
 * area) pair that has work, rank-major; returns the number of slices (may exceed cap; nothing beyond cap is written). */
typedef struct { uint32_t rank, area, root_begin, root_end; } hspf_area_slice;
uint32_t hspf_plan_areas(uint32_t n_areas, const uint32_t *roots_per_area, uint32_t world,
                         hspf_area_slice *out, uint32_t cap);

/* Number of mask words the roots of the whole job need (every rank passes the same list). */
int hspf_multi_mask_words(hspf_multi *m, const hspf_multi_graph *g, const uint32_t *roots, uint32_t n_roots, uint32_t *out_words);

/* Sharded run.  `all[i]` (i < n_local) are DEVICE buffers on local device i sized for ALL n_roots roots, laid out as
 * for hspf_run_device ([n_roots][n_vertices] ...).  Rank r computes rows [begin_r, end_r) straight into its own
 * buffers; the tables selected in `gather` (present in `all`) are then all-gathered in place, so that on return every
 * local device holds those tables for every root.  gather = 0: no exchange (each device holds its own rows only). */
#define HSPF_GATHER_DIST   0x1u
#define HSPF_GATHER_HOPS   0x2u
#define HSPF_GATHER_FLAGS  0x4u
#define HSPF_GATHER_MASK   0x8u
/* The exchange is enqueued on the ranks' communication streams and the call returns when the ranks' OWN rows are
 * complete: the gather then overlaps whatever the caller does next — typically the next hspf_multi_run into a second
 * set of tables.  A later hspf_multi_run into the SAME tables first waits for their pending gather; hspf_multi_wait
 * waits for all of them (call it before reading gathered rows). */
#define HSPF_GATHER_ASYNC  0x100u
int hspf_multi_wait(hspf_multi *m);
int hspf_multi_run(hspf_multi *m, const hspf_multi_graph *g, const uint32_t *roots, uint32_t n_roots,
                   uint32_t run_flags, hspf_result *all, uint32_t gather);

/* hspf_multi_run in two halves (ABI 6): the first hands every local device's slice to a lane of its context
 * (hspf_run_device_async) and returns a ticket; the second waits for the slices and then does what hspf_multi_run does
 * with `gather` (the exchange, synchronous or HSPF_GATHER_ASYNC).  Several tickets may be in flight, into DIFFERENT
 * tables; wait for them in ticket order (every rank the same order: the exchange is a collective).  No host thread is
 * created per call: the lanes' threads live as long as the context (hspf_multi_run itself uses them when it drives
 * more than one local device). */
int hspf_multi_run_async(hspf_multi *m, const hspf_multi_graph *g, const uint32_t *roots, uint32_t n_roots,
                         uint32_t run_flags, const hspf_result *all, uint64_t *ticket);
int hspf_multi_run_wait(hspf_multi *m, uint64_t ticket, hspf_result *all, uint32_t gather);

/* The collective on its own, for any per-root table (route tables after hspf_routes_device: "a single all-gather of
 * per-root route tables", BASELINE north_star): tables[i] is a device buffer on local device i of n_roots rows of
 * row_bytes bytes whose rows [begin_r, end_r) are filled; on return all rows are, on every device. */
int hspf_multi_allgather_rows(hspf_multi *m, void *const *tables, size_t row_bytes, uint32_t n_roots);

/* Statistics of the last hspf_multi_run on local device i (as hspf_get_stats). */
int hspf_multi_get_stats(const hspf_multi *m, uint32_t local_index, hspf_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* HOLO_SPF_HIP_H */
