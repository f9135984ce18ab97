"""CPU model of fixed-point schedules for the lane = root layout (one 64-root batch, distances only): how many
(vertex, batch) rows each schedule evaluates and in how many sweeps.  Numbers quoted in DESIGN.md §4 / profiles/r02_notes.md.

    python tools/sim_schedules.py jacobi  [spread|cluster]        per-sweep activity of the synchronous pull sweep
    python tools/sim_schedules.py gated   [spread|cluster] DELTA [K] [R]
                                                                  threshold-gated relaxation: a lane's tentative distance is
                                                                  visible to its neighbours only when <= T; T += DELTA when a
                                                                  bucket is stable (K = 0, strict delta-stepping) or every K sweeps
    python tools/sim_schedules.py precise [spread|cluster]        row loads with precise wake-ups in the correction tail

The reference settles every vertex once per root (holo-isis/src/spf.rs:552-556); a row of the lane = root layout carries
64 roots, so 1 x N rows per batch would be the same amount of work.  Results on isis-100k (N = 100 000, 64 roots):
    jacobi   spread 21.1 x N rows in 34 sweeps; 64 ADJACENT roots 20.6 x N in 34 (the chords make every root's hop
             wavefront cover the graph in 5 sweeps; what follows is 28 sweeps of corrections)
    gated    spread, strict: DELTA 10 / 25 / 50 -> 34.9 / 25.7 / 23.0 x N in 127 / 87 / 67 sweeps; raised every sweep
             (K = 1): DELTA 8 / 10 / 12 / 15 -> 19.8 / 18.3 / 18.5 / 18.9 x N in 37 / 35 / 35 / 34 sweeps;
             ONE root, strict, DELTA 10 / 25: 6.6 / 5.7 x N in 85 / 53 sweeps
    precise  21.1 M -> 19.6 M row loads (threshold 35 %)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402

INF = 1 << 29


def load(mode, R):
    g = synth.isis_100k()
    n = g.n
    rp = g.row_ptr.astype(np.int64); col = g.col.astype(np.int64); met = g.metric.astype(np.int64)
    src = np.repeat(np.arange(n), np.diff(rp))
    order = np.argsort(col, kind="stable")
    in_dst = col[order]; in_src = src[order]; in_w = met[order].astype(np.int32)
    in_ptr = np.zeros(n + 1, np.int64); np.add.at(in_ptr, in_dst + 1, 1); in_ptr = np.cumsum(in_ptr)
    roots = (np.arange(R) * n) // R if mode == "spread" else np.arange(R) * 3 + n // 2
    return n, in_ptr, in_src, in_dst, in_w, roots


def jacobi(mode, precise=False):
    R = 64
    n, in_ptr, in_src, in_dst, in_w, roots = load(mode, R)
    deg = np.diff(in_ptr)
    dist = np.full((n, R), INF, np.int32); dist[roots, np.arange(R)] = 0
    chl = np.zeros((n, R), bool); chl[roots, np.arange(R)] = True
    ch_rows = chl.any(axis=1)
    tot = base = prec = 0
    seen_dense = False
    for s in range(200):
        act = np.add.reduceat(ch_rows[in_src].astype(np.int32), in_ptr[:-1]) > 0
        if not act.any():
            break
        cand = dist[in_src] + in_w[:, None]
        hit = (chl[in_src] & (cand <= dist[in_dst])).any(axis=1)
        aff = np.add.reduceat(hit.astype(np.int32), in_ptr[:-1]) > 0
        c = int(ch_rows.sum())
        seen_dense = seen_dense or c > n // 2
        mode_p = precise and seen_dense and c < 0.35 * n
        new = np.minimum(np.minimum.reduceat(cand, in_ptr[:-1], axis=0), dist)
        new[~act] = dist[~act]
        lb = int(deg[act].sum())
        lp = int(deg[aff].sum()) + int(deg[ch_rows].sum()) if mode_p else lb
        chl = new != dist; ch_rows = chl.any(axis=1)
        assert not (ch_rows & ~aff).any()
        print(f"sweep {s:2d} changed_before {c:6d} active {int(act.sum()):6d} affected {int(aff.sum()):6d} changed {int(ch_rows.sum()):6d} "
              f"lanes_changed {int(chl.sum()):8d} row_loads {lb:8d}" + (f" -> {lp:8d}" if precise else ""))
        tot += int(act.sum()); base += lb; prec += lp
        dist = new
    print(f"{mode}: rows evaluated {tot} = {tot / n:.2f} x N, row loads {base}" + (f" -> {prec} with precise wake-ups" if precise else ""),
          "; max dist", int(dist.max()))


def gated(mode, delta, K, R):
    n, in_ptr, in_src, in_dst, in_w, roots = load(mode, R)
    t = np.full((n, R), INF, np.int32); t[roots, np.arange(R)] = 0
    T = 0
    vis = np.where(t <= T, t, INF)
    chg = (vis != INF).any(axis=1)
    tot = sweeps = since = 0
    while True:
        act = np.add.reduceat(chg[in_src].astype(np.int32), in_ptr[:-1]) > 0
        if not act.any():
            if T >= INF:
                break
            pend = t[(t > T) & (t < INF)]
            T = INF if pend.size == 0 else max(T + delta, int(pend.min()))
            vis_new = np.where(t <= T, t, INF)
            chg = (vis_new != vis).any(axis=1); vis = vis_new
            if not chg.any() and T >= INF:
                break
            continue
        new = np.minimum(np.minimum.reduceat(vis[in_src] + in_w[:, None], in_ptr[:-1], axis=0), t)
        new[~act] = t[~act]
        t = new
        vis_new = np.where(t <= T, t, INF)
        chg = (vis_new != vis).any(axis=1); vis = vis_new
        tot += int(act.sum()); sweeps += 1; since += 1
        if K and since >= K and T < INF:
            T += delta; since = 0
            vis_new = np.where(t <= T, t, INF); chg |= (vis_new != vis).any(axis=1); vis = vis_new
    print(f"{mode} R={R} delta={delta} K={K}: {sweeps} sweeps, rows evaluated {tot} = {tot / n:.2f} x N")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "jacobi"
    mode = sys.argv[2] if len(sys.argv) > 2 else "spread"
    if what == "jacobi":
        jacobi(mode)
    elif what == "precise":
        jacobi(mode, True)
    else:
        gated(mode, int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0, int(sys.argv[5]) if len(sys.argv) > 5 else 64)
