"""Where a structural one-row hspf_graph_patch spends its time (isis-100k): wall per call of the C entry point, and with
HSPF_PATCH_TIMING=1 the laps inside it.   python tools/gpu_patch_probe.py [n_patches]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E, _lib as L     # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    g = synth.isis_100k()
    ctx = E.SpfContext(0)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    rng = np.random.default_rng(5)
    walls = {"drop": [], "restore": []}
    for i in range(reps):
        v = int(rng.integers(0, g.n))
        a, b = int(G.row_ptr[v]), int(G.row_ptr[v + 1])
        if b - a < 2:
            continue
        full = (G.col[a:b].copy(), G.metric[a:b].copy())
        for what, row in (("drop", (full[0][:-1], full[1][:-1])), ("restore", full)):
            vs = np.array([v], np.uint32); rp = np.array([0, len(row[0])], np.uint32)
            dcol = np.ascontiguousarray(row[0]); dmet = np.ascontiguousarray(row[1]); nf = np.array([G.vflags[v]], np.uint8)
            r = L.HspfRows(1, E._u32(vs), E._u32(rp), E._u32(dcol), E._u32(dmet), nf.ctypes.data_as(L.u8p))
            t0 = time.perf_counter()
            rc = ctx.lib.hspf_graph_patch(ctx.handle, G.handle, ctypes.byref(r))
            walls[what].append((time.perf_counter() - t0) * 1e3)
            assert rc == 0, ctx.last_error()
            # keep the numpy mirrors of the Graph object in step (outside the timed call)
            G._pending.append((vs, [dcol], [dmet], nf))          # (what G.patch records for its numpy mirrors)
    for k, w in walls.items():
        w = np.array(w[2:])
        print(f"{k}: median {np.median(w):.3f} ms  min {w.min():.3f}  max {w.max():.3f}  ({len(w)} calls, n {g.n}, e {len(g.col)})", flush=True)
    # the run that follows is bit-exact (a fresh upload of the mirrors gives the same tables)
    roots = np.arange(64, dtype=np.uint32) * 997
    r1 = ctx.run(G, roots, 0)
    G2 = ctx.upload(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
    r2 = ctx.run(G2, roots, 0)
    assert np.array_equal(r1.dist, r2.dist) and np.array_equal(r1.first_hop_mask, r2.first_hop_mask)
    print("patched graph == fresh upload on 64 roots", flush=True)


if __name__ == "__main__":
    main()
