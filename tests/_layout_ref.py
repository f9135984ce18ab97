"""CPU restatement (numpy + plain loops) of the device graph layout that hspf_graph_upload / hspf_graph_patch
build with the kernels of holo_amd/csrc/graph_build.hip.h.  TEST INFRASTRUCTURE ONLY.

Semantics restated (include/holo_spf_hip.h, hspf_graph_export):
  two-way check   link u->t usable iff row t lists u, cost not compared
                  (holo-ospf/src/spf.rs:654-664, holo-isis/src/spf.rs:616-627)
  kept            two-way and the source may be expanded (not HSPF_VF_NO_EXPAND, holo-isis/src/spf.rs:557-604)
  out-rows        kept links in the caller's order
  in-rows         kept links into a vertex by (cost descending, source ascending, position ascending);
                  bit 31 of the source = HSPF_VF_NO_TRANSIT of the source (routers only: ignored on network vertices)
  leaves          exactly one kept in-link, and the kept out-links (at most one) lead back to its source; bit 30 of the
                  source of every in-row entry = the source is a leaf
"""
import numpy as np

VF_NETWORK, VF_NO_TRANSIT, VF_NO_EXPAND = 1, 2, 4
RF_MANY, RF_NT, RF_ZERO, RF_GIANT = 1, 2, 4, 16


def layout(row_ptr, col, metric, vflags):
    n = len(row_ptr) - 1
    rp = row_ptr.astype(np.int64)
    src = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    pos = np.arange(len(col), dtype=np.int64) - rp[src]
    rows = [set(col[rp[u]:rp[u + 1]].tolist()) for u in range(n)]
    twoway = np.array([int(src[k]) in rows[int(col[k])] for k in range(len(col))], dtype=np.uint8)
    keep = (twoway == 1) & ((vflags[src] & VF_NO_EXPAND) == 0) if len(col) else np.zeros(0, bool)
    ks, kt, kw, kp = src[keep], col[keep].astype(np.int64), metric[keep].astype(np.int64), pos[keep]
    out_ptr = np.zeros(n + 1, np.int64)
    np.add.at(out_ptr, ks + 1, 1)
    out_ptr = np.cumsum(out_ptr)
    order = np.lexsort((kp, ks, -kw, kt))                 # target, cost desc, source, position
    in_ptr = np.zeros(n + 1, np.int64)
    np.add.at(in_ptr, kt + 1, 1)
    in_ptr = np.cumsum(in_ptr)
    nt = ((vflags[ks[order]] & VF_NO_TRANSIT) != 0) & ((vflags[ks[order]] & VF_NETWORK) == 0)   # routers only
    deg_in, deg_out = np.diff(in_ptr), np.diff(out_ptr)
    leaf = np.zeros(n, np.uint8)
    for v in range(n):
        if deg_in[v] == 1 and (deg_out[v] == 0 or (deg_out[v] == 1 and kt[out_ptr[v]] == ks[order][in_ptr[v]])):
            leaf[v] = 1
    in_src = (ks[order] | (nt.astype(np.int64) << 31) | (leaf[ks[order]].astype(np.int64) << 30)).astype(np.uint32)
    rowflags = np.zeros(n, np.uint8)
    t_o, w_o, s_o = kt[order], kw[order], ks[order]
    for t in range(n):
        a, b = in_ptr[t], in_ptr[t + 1]
        f = (RF_MANY if b - a > 16 else 0) | (RF_GIANT if b - a > 256 else 0)
        if nt[a:b].any():
            f |= RF_NT
        if ((w_o[a:b] == 0) & (s_o[a:b] >= t)).any():
            f |= RF_ZERO
        rowflags[t] = f
    # work units of the sweep kernels: a 16-vertex chunk with a row of more than 32 in-links = 4 units of one row per wave
    deg = np.diff(in_ptr)
    nb = (n + 15) // 16
    heavy = np.array([bool((deg[c * 16:c * 16 + 16] > 32).any()) for c in range(nb)], dtype=bool)
    units = np.zeros(0, np.uint32)
    if heavy.any():
        hv = [min(c * 16 + 4 * k, n) | 0x80000000 for c in np.nonzero(heavy)[0] for k in range(4)]
        nv = [c * 16 for c in np.nonzero(~heavy)[0]]
        units = np.asarray(hv + nv, np.uint32)
    return {
        "twoway": twoway, "units": units,
        "in_ptr": in_ptr.astype(np.uint32), "in_src": in_src, "in_cost": w_o.astype(np.uint32),
        "in_pos": kp[order].astype(np.uint32),
        "out_ptr": out_ptr.astype(np.uint32), "out_dst": kt.astype(np.uint32), "out_cost": kw.astype(np.uint32),
        "out_pos": kp.astype(np.uint32), "rowflags": rowflags, "leaf": leaf,
    }
