"""k_lv on the large graphs it is chosen for: isis-100k, one and two random roots, against the oracle; with row patches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from holo_amd import engine as E, synth
from oracle import graph_oracle as go
g = synth.isis_100k()
ctx = E.SpfContext(0)
G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
rng = np.random.default_rng(11)
ok = runs = 0
for it in range(24):
    k = 1 + it % 2
    roots = rng.choice(g.n, size=k, replace=False).astype(np.uint32)
    res = ctx.run(G, roots, 0)
    ref = go.run(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=res.first_hop_mask.shape[2])
    good = (np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops) and np.array_equal(res.flags & 1, ref.flags)
            and np.array_equal(res.first_hop_mask, ref.mask) and res.stats["lane_vertex"] == 1)
    ok += good; runs += 1
    if not good: print("MISMATCH", it, roots, res.stats, flush=True)
    if it % 3 == 2:                                  # re-originate a few rows: costs only, then a link less
        vs = np.sort(rng.choice(g.n, size=3, replace=False))
        rows, fl = [], []
        for v in vs.tolist():
            c = G.col[G.row_ptr[v]:G.row_ptr[v + 1]]; m = G.metric[G.row_ptr[v]:G.row_ptr[v + 1]].copy()
            if it % 6 == 5 and len(c) > 1: c, m = c[1:], m[1:]
            else: m[:] = rng.integers(1, 60, size=len(m))
            rows.append((c, m)); fl.append(int(G.vflags[v]))
        G.patch(vs, rows, fl)
print(f"k_lv on isis-100k: {ok}/{runs} runs bit-exact (one and two roots, patches in between)")
