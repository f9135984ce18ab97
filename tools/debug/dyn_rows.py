"""Where the sweep time of a graph with zero-cost links goes: rows evaluated (HSPF_RUN_COUNT_ROWS), passes, time — for the
plain graph, zero-cost links from LOWER-numbered sources only (no RF_ZERO rows), from HIGHER-numbered only, and both."""
import os, sys, json, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tools"))
import torch
from holo_amd import engine as E, synth
import gpu_dynamic_probe as P

def variant(g, share, seed, which):
    rng = np.random.default_rng(seed)
    m = g.metric.copy()
    src = np.repeat(np.arange(g.n, dtype=np.int64), np.diff(g.row_ptr.astype(np.int64)))
    pick = rng.random(len(m)) < share
    if which == "asc": pick &= src < g.col
    if which == "desc": pick &= src > g.col
    m[pick] = 0
    return synth.CsrGraph(g.row_ptr, g.col, m, g.vflags, g.max_path_metric, g.name)

ctx = E.SpfContext(0)
g0 = synth.isis_100k()
roots = (np.arange(64, dtype=np.uint64) * g0.n // 64).astype(np.uint32)
for which in ("none", "asc", "desc", "both"):
    g = g0 if which == "none" else variant(g0, 0.01, 11, which)
    ms, st = P.timed(ctx, g, roots, reps=5)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    dev = torch.device("cuda:0"); R, n = 64, g.n
    t = dict(dist=torch.empty((R, n), dtype=torch.int32, device=dev), hops=torch.empty((R, n), dtype=torch.int16, device=dev),
             flags=torch.empty((R, n), dtype=torch.int16, device=dev), mask=torch.empty((R, n, 1), dtype=torch.int64, device=dev))
    st2 = ctx.run_device(G, roots, E.RUN_COUNT_ROWS, dist_ptr=t["dist"].data_ptr(), hops_ptr=t["hops"].data_ptr(), flags_ptr=t["flags"].data_ptr(), mask_ptr=t["mask"].data_ptr(), mask_words=1)
    G.free()
    print(json.dumps({"zero_links": which, "ms_call": round(ms, 3), "ms_relax": round(st["ms_relax"], 3), "ms_repair": round(st["ms_repair"], 3), "launches": st["n_relax_launches"],
                      "dbg1": hex(st["dbg"][1]), "rows_x_N": round(st2["rows_recomputed"] / g.n, 2), "repaired": st["n_repaired_roots"], "state_bytes": st["state_bytes"]}), flush=True)
