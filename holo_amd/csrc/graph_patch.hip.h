// graph_patch.hip.h — a STRUCTURAL hspf_graph_patch without the rebuild (`trigger_lsps`,
// holo-isis/src/spf.rs:144, 733-735: one LSP re-originated with a link gone or a new neighbour).
//
// Until round 6 every patch that changed a row's targets rebuilt the whole layout from the spliced raw CSR (graph_build.hip.h:
// ~25 launches, each a pass over all links: 0.20 ms at a million links).  What a replaced row u can change is local:
//
//   AFFECTED rows A = the replaced rows, their old targets and their new targets.
//   * the kept OUT-row of v changes only when v is replaced (links, NO_EXPAND) or v lists a replaced u whose row gained or
//     lost v (the two-way check of v -> u): v is an old or new target of u;
//   * the IN-row of t changes only when a source of it is replaced (links, costs, the source's overload bit) or t itself is
//     (two-way checks of the links into it): t in A;
//   * row flags, the hop-count facts, the ELL record and the leaf mark of a row are functions of its in- and out-row: A;
//   * SRC_LEAF on the links FROM a vertex whose leaf mark flipped: those sit in the in-row of the leaf's one out-neighbour,
//     which need not be in A — every in-link passes through the shift kernel anyway, which sets the bit from the new marks;
//   * the compact in- / out-arrays behind the first affected row move by the length changes in front of them.
//
// So: one workgroup per affected row derives its new kept out-row and in-row FROM THE NEW RAW CSR (the same rules as kb_links /
// kb_rank / kb_rowflags / kb_ell / kb_leaf_mark) into a staging area and rewrites the row's fixed-stride records in place
// (kb_pa_rows; the workgroup that finishes last turns the rows' length changes into shifts); one streaming kernel writes the six link
// arrays into their second set — unaffected entries from the old arrays at their old place, affected rows from the staging
// area — sets SRC_LEAF, finds the largest cost, and moves the row bounds in place (kb_pa_shift); one pass over the PER-ROW arrays
// re-derives the build's summary (kb_pa_summary) and the work units / XCD ranges come from the build's own kb_units_small.
// Derived work is O(affected rows x their degrees); the two streaming passes (raw splice, link shift) move ~40 bytes per link
// at HBM speed in one launch each.  The layout equals a fresh upload's array by array (tests/test_gpu_graph_build.py).
//
// Anything the staging area cannot hold (a row of more than PA_OUT_STRIDE links, more than PA_IN_STRIDE kept in-links, more
// than PA_MAX_ROWS affected rows, giant / hub graphs) takes the rebuild: decided by the host before the launches, or by
// kb_pa_rows (GB_ERR_PATCH in BuildInfo::err: the later kernels then do nothing and the host rebuilds from the raw CSR, which
// is complete either way).
#pragma once
#include "graph_build.hip.h"

namespace hspf {

constexpr uint32_t GB_ERR_PATCH = 4u;       // kb_pa_*: an affected row does not fit the staging area -> rebuild
constexpr uint32_t PA_IN_STRIDE = 256u;     // staged kept in-links per affected row
constexpr uint32_t PA_OUT_STRIDE = 512u;    // staged kept out-links per affected row = longest raw row this path takes
constexpr uint32_t PA_MAX_ROWS = 2048u;     // affected rows per patch
constexpr uint32_t PA_LDS_ROWS = 255u;      // kb_pa_shift keeps the rows' metadata in LDS up to this many affected rows
constexpr uint32_t PA_META = 8u;            // per affected row j (arrays of na + 1 words): old in-start, old in-length, old out-start,
                                            // old out-length, new in-length, new out-length, in-shift, out-shift

// The new raw CSR of a structural patch in ONE launch: row bounds (old ones + the length changes of the replaced rows in front),
// the replaced rows' flags, targets and costs (a link's row is looked for among the REPLACED rows only), and the zeroing of the
// incremental path's BuildInfo block — until round 6 three dependent launches of a few microseconds each.  The parts touch
// disjoint arrays.
__global__ void __launch_bounds__(GB_BLOCK)
kb_patch_raw(uint32_t n, uint32_t e_new, const uint32_t *__restrict__ old_row_ptr, const uint32_t *__restrict__ old_col,
             const uint32_t *__restrict__ old_metric, uint32_t n_changed, const uint32_t *__restrict__ changed, const uint32_t *__restrict__ shift,
             const uint32_t *__restrict__ delta_ptr, const uint32_t *__restrict__ delta_col, const uint32_t *__restrict__ delta_metric,
             uint32_t *__restrict__ new_row_ptr, uint32_t *__restrict__ new_col, uint32_t *__restrict__ new_metric, const uint8_t *__restrict__ nf,
             uint8_t *__restrict__ vflags, uint32_t *__restrict__ clear_words, uint32_t n_clear) {
  const uint32_t k = blockIdx.x * GB_BLOCK + threadIdx.x;
  if (k < n_clear) clear_words[k] = 0u;
  for (uint32_t i = k; i < n_changed; i += gridDim.x * GB_BLOCK) vflags[changed[i]] = nf[i];
  if (k <= n) {
    uint32_t lo = 0, hi = n_changed;                       // replaced rows in front of row k
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (changed[mid] < k) lo = mid + 1; else hi = mid;
    }
    new_row_ptr[k] = old_row_ptr[k] + shift[lo];           // modulo 2^32
  }
  if (k >= e_new) return;
  uint32_t lo = 0, hi = n_changed;                         // first replaced row whose new start is behind k
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (old_row_ptr[changed[mid]] + shift[mid] <= k) lo = mid + 1; else hi = mid;
  }
  if (lo != 0u) {
    const uint32_t j = lo - 1u;
    const uint32_t off = k - (old_row_ptr[changed[j]] + shift[j]);
    if (off < delta_ptr[j + 1] - delta_ptr[j]) {
      new_col[k] = delta_col[delta_ptr[j] + off];
      new_metric[k] = delta_metric[delta_ptr[j] + off];
      return;
    }
  }
  const uint32_t ko = k - shift[lo];
  new_col[k] = old_col[ko];
  new_metric[k] = old_metric[ko];
}

__device__ __forceinline__ void pa_scan_body(uint32_t na, uint32_t *meta, uint32_t kept_old, BuildInfo *info, uint32_t *sh);

// one workgroup per affected row
__global__ void __launch_bounds__(256)
kb_pa_rows(uint32_t n, uint32_t na, const uint32_t *__restrict__ A, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col,
           const uint32_t *__restrict__ metric, const uint8_t *__restrict__ vflags, const uint32_t *__restrict__ in_ptr,
           const uint32_t *__restrict__ out_ptr, uint32_t *meta, uint32_t *__restrict__ st_in, uint32_t *__restrict__ st_out,
           uint8_t *__restrict__ rowflags, uint8_t *__restrict__ rowaux, uint8_t *__restrict__ leaf, uint32_t *__restrict__ ell_so,
           uint32_t *__restrict__ ell_w, uint32_t *__restrict__ ell_od, uint32_t giant_deg, uint32_t kept_old, uint32_t *done, BuildInfo *info) {
  __shared__ uint32_t s_col[PA_OUT_STRIDE], s_met[PA_OUT_STRIDE];            // the row's raw links; later: the ranked in-row (source, cost)
  __shared__ uint32_t s_isrc[PA_IN_STRIDE], s_iw[PA_IN_STRIDE], s_ipos[PA_IN_STRIDE];
  __shared__ uint32_t s_scan[GB_BLOCK];
  __shared__ uint32_t s_odst[16];
  __shared__ uint32_t s_cnt;
  const uint32_t j = blockIdx.x, a = A[j], tid = threadIdx.x;
  const uint32_t r0 = row_ptr[a];
  uint32_t len = row_ptr[a + 1] - r0;
  const uint32_t na1 = na + 1u;
  if (tid == 0) {
    s_cnt = 0u;
    const uint32_t i0 = in_ptr[a], o0 = out_ptr[a];
    meta[j] = i0; meta[na1 + j] = in_ptr[a + 1] - i0; meta[2u * na1 + j] = o0; meta[3u * na1 + j] = out_ptr[a + 1] - o0;
  }
  if (len > PA_OUT_STRIDE) {                                   // (the host sends no such patch down this path)
    if (tid == 0) atomicOr(&info->err, GB_ERR_PATCH);
    len = 0u;
  }
  for (uint32_t i = tid; i < len; i += 256u) { s_col[i] = col[r0 + i]; s_met[i] = metric[r0 + i]; }
  if (tid < 16u) s_odst[tid] = 0u;
  __syncthreads();
  const uint32_t af = vflags[a];
  const bool expand_a = !(af & HSPF_VF_NO_EXPAND);
  const bool row_net = (af & HSPF_VF_NETWORK) != 0u;
  // ---- the links 2 tid, 2 tid + 1 of the row: two-way check (kb_links), and the kept links the other way = the in-row
  bool kp[2] = {false, false};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t i = 2u * tid + (uint32_t)q;
    if (i >= len) continue;
    const uint32_t t = s_col[i];
    bool first = true;                                          // the first link of the row to this target collects the in-links from it
    for (uint32_t i2 = 0; i2 < i && first; ++i2) first = s_col[i2] != t;
    const uint32_t tf = vflags[t];
    const bool collect = first && !(tf & HSPF_VF_NO_EXPAND);
    const uint32_t tsrc = t | (((tf & HSPF_VF_NO_TRANSIT) && !(tf & HSPF_VF_NETWORK)) ? SRC_NO_TRANSIT : 0u);   // (kb_scatter)
    const uint32_t tb0 = row_ptr[t], tb1 = row_ptr[t + 1];
    bool two = false;
    for (uint32_t k2 = tb0; k2 < tb1 && !(two && !collect); k2 += 4u) {       // four entries per step: one wait per four loads (kb_links)
      const uint32_t c[4] = {col[k2], k2 + 1u < tb1 ? col[k2 + 1u] : INF, k2 + 2u < tb1 ? col[k2 + 2u] : INF, k2 + 3u < tb1 ? col[k2 + 3u] : INF};
#pragma unroll
      for (uint32_t x = 0; x < 4u; ++x) {                                        // (a < n <= 2^24: INF never matches)
        if (c[x] != a) continue;
        two = true;
        if (!collect) continue;
        const uint32_t slot = atomicAdd(&s_cnt, 1u);
        if (slot < PA_IN_STRIDE) { s_isrc[slot] = tsrc; s_iw[slot] = metric[k2 + x]; s_ipos[slot] = k2 + x - tb0; }
      }
    }
    kp[q] = two && expand_a;
  }
  // ---- the kept out-row, in link order
  uint32_t total_out;
  const uint32_t mine = (kp[0] ? 1u : 0u) + (kp[1] ? 1u : 0u);
  uint32_t pos = gb_block_exclusive(mine, s_scan, total_out);
  uint32_t *so = st_out + (size_t)j * 3u * PA_OUT_STRIDE;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (!kp[q]) continue;
    const uint32_t i = 2u * tid + (uint32_t)q;
    so[pos] = s_col[i]; so[PA_OUT_STRIDE + pos] = s_met[i]; so[2u * PA_OUT_STRIDE + pos] = i;
    if (pos < 16u) s_odst[pos] = s_col[i];
    ++pos;
  }
  __syncthreads();                                             // s_cnt, the in-entries and s_odst are complete; s_col / s_met are free
  uint32_t d = s_cnt;
  if (d > PA_IN_STRIDE) {
    if (tid == 0) atomicOr(&info->err, GB_ERR_PATCH);
    d = 0u;
  }
  // ---- the in-row in its order: cost descending, source ascending, position ascending (kb_rank)
  uint32_t sraw = 0u, w = 0u, f = 0u, src = 0u, rank = 0u;
  if (tid < d) {
    sraw = s_isrc[tid]; w = s_iw[tid]; f = s_ipos[tid]; src = sraw & SRC_MASK;
    for (uint32_t k = 0; k < d; ++k) {
      const uint32_t wk = s_iw[k], sk = s_isrc[k] & SRC_MASK, fk = s_ipos[k];
      rank += (wk > w || (wk == w && (sk < src || (sk == src && fk < f)))) ? 1u : 0u;
    }
    uint32_t *si = st_in + (size_t)j * 3u * PA_IN_STRIDE;
    si[rank] = sraw; si[PA_IN_STRIDE + rank] = w; si[2u * PA_IN_STRIDE + rank] = f;
    s_col[rank] = sraw; s_met[rank] = w;
  }
  // ---- row flags and the hop-count facts (kb_rowflags)
  bool nt = false, zero = false, bad = false;
  if (tid < d) {
    nt = (sraw & SRC_NO_TRANSIT) != 0u;
    zero = w == 0u && src >= a;
    bad = row_net ? !(w == 0u && !(vflags[src] & HSPF_VF_NETWORK) && src > a) : w != 1u;
  }
  const int any_nt = __syncthreads_or(nt ? 1 : 0), any_zero = __syncthreads_or(zero ? 1 : 0), any_bad = __syncthreads_or(bad ? 1 : 0);
  // (the barriers above also order the ranked copy in s_col / s_met before the reads below)
  if (tid == 0) {
    rowflags[a] = (uint8_t)((d > 16u ? RF_MANY : 0u) | (d > giant_deg ? RF_GIANT : 0u) | (any_nt ? RF_NT : 0u) | (any_zero ? RF_ZERO : 0u));
    rowaux[a] = (uint8_t)((any_bad ? RA_BAD : 0u) | ((row_net && d > 0u) ? RA_NET_IN : 0u));
    const uint8_t lf = (d == 1u && (total_out == 0u || (total_out == 1u && s_odst[0] == (s_col[0] & SRC_MASK)))) ? 1u : 0u;   // (kb_leaf_mark)
    if (leaf[a] != lf) atomicAdd(&info->n_leaf, 1u);             // a flipped mark: kb_pa_shift refreshes SRC_LEAF on every in-link
    leaf[a] = lf;
    meta[4u * na1 + j] = d; meta[5u * na1 + j] = total_out;
  }
  // ---- the ELL record (kb_ell)
  if (tid < 16u) {
    uint32_t eso = n << 8, ew = 0u, eod = 0xFFFFFFFFu;
    if (d <= 16u && tid < d) { eso = (s_col[tid] & SRC_MASK) << 8; ew = s_met[tid]; }
    if (tid < total_out) eod = s_odst[tid] << 2;
    if (tid == 0u) eso |= (d <= 16u ? d : 0x1Fu) | (total_out > 16u ? 0x20u : 0u) | (row_net ? 0x80u : 0u);
    ell_so[(size_t)a * 16u + tid] = eso; ell_w[(size_t)a * 16u + tid] = ew; ell_od[(size_t)a * 16u + tid] = eod;
  }
  // ---- the workgroup that finishes last turns the rows' length changes into shifts (no wait: the canonical "last block"
  // hand-over — every workgroup publishes its metadata, fences at agent scope and counts itself; whoever counts na - 1 others
  // fences again and reads them all)
  __shared__ uint32_t s_last;
  __syncthreads();                                             // thread 0's metadata stores are issued
  if (tid == 0) {
    __threadfence();
    s_last = (atomicAdd(done, 1u) == na - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  pa_scan_body(na, meta, kept_old, info, s_scan);
}

// one workgroup: the rows' length changes -> shifts (exclusive prefix sums, modulo 2^32), the new number of kept links
__device__ __forceinline__ void pa_scan_body(uint32_t na, uint32_t *meta, uint32_t kept_old, BuildInfo *info, uint32_t *sh) {
  const uint32_t na1 = na + 1u, tid = threadIdx.x;
  const uint32_t ipt = (na + GB_BLOCK - 1u) / GB_BLOCK;                       // <= PA_MAX_ROWS / 256
  const uint32_t j0 = min(tid * ipt, na), j1 = min(j0 + ipt, na);
  uint32_t tot[2];
  for (uint32_t which = 0; which < 2u; ++which) {
    const volatile uint32_t *oldl = meta + (1u + 2u * which) * na1, *newl = meta + (4u + which) * na1;
    uint32_t *shift = meta + (6u + which) * na1;
    uint32_t s = 0u;
    for (uint32_t j = j0; j < j1; ++j) s += newl[j] - oldl[j];
    uint32_t total;
    uint32_t run = gb_block_exclusive(s, sh, total);
    for (uint32_t j = j0; j < j1; ++j) { shift[j] = run; run += newl[j] - oldl[j]; }
    if (tid == 0) shift[na] = total;
    tot[which] = total;
    __syncthreads();
  }
  if (tid == 0) {
    info->kept = kept_old + tot[0];
    if (tot[0] != tot[1]) atomicOr(&info->err, GB_ERR_PATCH);              // every kept link is in one in-row and one out-row
  }
}

// Entry k of a shifted array: from the staging area when it lies in an affected row, else the old entry k - (length changes
// of the affected rows in front of it).  J = affected rows whose NEW start (old start + shift) is at or before k.
struct PaWhere { bool staged; uint32_t j, off, old_k; };
__device__ __forceinline__ PaWhere pa_locate(uint32_t k, uint32_t na, const uint32_t *__restrict__ old0, const uint32_t *__restrict__ newl,
                                             const uint32_t *__restrict__ shift) {
  uint32_t lo = 0, hi = na;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (old0[mid] + shift[mid] <= k) lo = mid + 1; else hi = mid;
  }
  PaWhere w{false, 0u, 0u, k - shift[lo]};
  if (lo != 0u) {
    const uint32_t j = lo - 1u, off = k - (old0[j] + shift[j]);
    if (off < newl[j]) { w.staged = true; w.j = j; w.off = off; }
  }
  return w;
}

__global__ void __launch_bounds__(GB_BLOCK)
kb_pa_shift(uint32_t n, uint32_t na, const uint32_t *__restrict__ A, const uint32_t *meta, const uint32_t *__restrict__ st_in,
            const uint32_t *__restrict__ st_out, const uint32_t *__restrict__ o_in_src, const uint32_t *__restrict__ o_in_w,
            const uint32_t *__restrict__ o_in_fpos, const uint32_t *__restrict__ o_out_dst, const uint32_t *__restrict__ o_out_w,
            const uint32_t *__restrict__ o_out_fpos, uint32_t *__restrict__ n_in_src, uint32_t *__restrict__ n_in_w,
            uint32_t *__restrict__ n_in_fpos, uint32_t *__restrict__ n_out_dst, uint32_t *__restrict__ n_out_w, uint32_t *__restrict__ n_out_fpos,
            uint32_t *in_ptr, uint32_t *out_ptr, const uint8_t *__restrict__ leaf, BuildInfo *__restrict__ info) {
  if (*(volatile uint32_t *)&info->err & GB_ERR_PATCH) return;            // (block-uniform: written by earlier launches)
  const uint32_t k = blockIdx.x * GB_BLOCK + threadIdx.x;
  const uint32_t na1 = na + 1u, kept = info->kept;
  // the tables of the searches in LDS when they fit (a patch of a few rows has a few dozen affected rows): a search is then
  // ~5 LDS reads instead of ~5 dependent trips to the L2, twice per thread
  __shared__ uint32_t s_meta[PA_META * (PA_LDS_ROWS + 1u)];
  if (na <= PA_LDS_ROWS) {
    for (uint32_t i = threadIdx.x; i < PA_META * na1; i += GB_BLOCK) s_meta[i] = meta[i];
    __syncthreads();
    meta = s_meta;
  }
  uint32_t wmax = 0u;
  const bool leaf_flips = info->n_leaf != 0u;                    // (kb_pa_rows counts the flipped marks there; kb_pa_summary sets the real count later)
  if (k < kept) {
    {
      const PaWhere p = pa_locate(k, na, meta, meta + 4u * na1, meta + 6u * na1);
      uint32_t s, w, f;
      if (p.staged) {
        const uint32_t *si = st_in + (size_t)p.j * 3u * PA_IN_STRIDE;
        s = si[p.off]; w = si[PA_IN_STRIDE + p.off]; f = si[2u * PA_IN_STRIDE + p.off];
      } else {
        s = o_in_src[p.old_k]; w = o_in_w[p.old_k]; f = o_in_fpos[p.old_k];
      }
      // SRC_LEAF (kb_leaf_links) from the marks kb_pa_rows left: staged entries carry none yet; the others keep theirs unless
      // some affected row's mark flipped (then every in-link looks its source up: a random byte per link, rarely needed)
      if (p.staged || leaf_flips) s = (s & ~SRC_LEAF) | (leaf[s & SRC_MASK] ? SRC_LEAF : 0u);
      n_in_src[k] = s; n_in_w[k] = w; n_in_fpos[k] = f;
    }
    {
      const PaWhere p = pa_locate(k, na, meta + 2u * na1, meta + 5u * na1, meta + 7u * na1);
      uint32_t t, w, f;
      if (p.staged) {
        const uint32_t *so = st_out + (size_t)p.j * 3u * PA_OUT_STRIDE;
        t = so[p.off]; w = so[PA_OUT_STRIDE + p.off]; f = so[2u * PA_OUT_STRIDE + p.off];
      } else {
        t = o_out_dst[p.old_k]; w = o_out_w[p.old_k]; f = o_out_fpos[p.old_k];
      }
      n_out_dst[k] = t; n_out_w[k] = w; n_out_fpos[k] = f;
      wmax = w;
    }
  }
  for (int o = 32; o; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o));
  if ((threadIdx.x & 63u) == 0u) gb_raise_wmax((BuildInfo *)info, wmax);   // (kb_scatter)
  if (k <= n) {                                                  // row bounds, in place: affected rows in front of row k
    uint32_t lo = 0, hi = na;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (A[mid] < k) lo = mid + 1; else hi = mid;
    }
    in_ptr[k] += meta[6u * na1 + lo];
    out_ptr[k] += meta[7u * na1 + lo];
  }
}

// BuildInfo's per-row summary from the per-row arrays (what kb_rowflags / kb_leaf_mark reduce on the fly during a build), and
// the heavy flag of every 16-vertex chunk (kb_unit_count)
__global__ void __launch_bounds__(GB_BLOCK)
kb_pa_summary(uint32_t n, const uint32_t *__restrict__ in_ptr, const uint8_t *__restrict__ rowflags, const uint8_t *__restrict__ rowaux,
              const uint8_t *__restrict__ leaf, uint32_t *__restrict__ hf, uint32_t heavy_deg, BuildInfo *__restrict__ info) {
  if (*(volatile uint32_t *)&info->err & GB_ERR_PATCH) return;
  const uint32_t v = blockIdx.x * GB_BLOCK + threadIdx.x, lane = threadIdx.x & 63u;
  const bool valid = v < n;
  const uint32_t deg = valid ? in_ptr[v + 1] - in_ptr[v] : 0u;
  const uint32_t rf = valid ? rowflags[v] : 0u, aux = valid ? rowaux[v] : 0u, lf = valid ? leaf[v] : 0u;
  uint32_t heavy = deg > heavy_deg ? 1u : 0u;
  for (int o = 8; o; o >>= 1) heavy |= (uint32_t)__shfl_xor((int)heavy, o);           // the chunk's sixteen lanes
  if (valid && (threadIdx.x & 15u) == 0u) hf[v >> 4] = heavy;
  uint32_t mx = deg, fo = rf;
  for (int o = 32; o; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); fo |= (uint32_t)__shfl_xor((int)fo, o); }
  const uint64_t m_bad = __ballot((aux & RA_BAD) != 0u), m_zero = __ballot((rf & RF_ZERO) != 0u), m_net = __ballot((aux & RA_NET_IN) != 0u),
                 m_leaf = __ballot(lf != 0u);
  __shared__ uint32_t red[GB_BLOCK / 64][6];
  if (lane == 0u) {
    uint32_t *r = red[threadIdx.x >> 6];
    r[0] = mx; r[1] = fo; r[2] = (uint32_t)__popcll(m_bad); r[3] = (uint32_t)__popcll(m_zero); r[4] = m_net ? 1u : 0u; r[5] = (uint32_t)__popcll(m_leaf);
  }
  __syncthreads();
  if (threadIdx.x != 0u) return;
  uint32_t bmx = 0, bfo = 0, nbad = 0, nzero = 0, bnet = 0, nleaf = 0;
  for (int w = 0; w < GB_BLOCK / 64; ++w) { bmx = max(bmx, red[w][0]); bfo |= red[w][1]; nbad += red[w][2]; nzero += red[w][3]; bnet |= red[w][4]; nleaf += red[w][5]; }
  if (bmx > *(volatile uint32_t *)&info->max_in_deg) atomicMax(&info->max_in_deg, bmx);
  if (bfo & ~*(volatile uint32_t *)&info->any_rowflags) atomicOr(&info->any_rowflags, bfo);
  if (nbad) atomicAdd(gb_spread(info, GB_SC_BAD), nbad);
  if (nzero) atomicAdd(gb_spread(info, GB_SC_ZERO), nzero);
  if (nleaf) atomicAdd(gb_spread(info, GB_SC_LEAF), nleaf);
  if (bnet && !*(volatile uint32_t *)&info->hc_net) info->hc_net = 1u;
}

}  // namespace hspf
