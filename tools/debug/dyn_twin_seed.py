"""Debug: one random IS-IS instance with zero metrics — engine tables (with and without pop ranks) against the oracle, per graph."""
import os, sys, json, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from holo_amd import engine as E, isis as H
from oracle import graph_oracle as go, isis_ref as R
from _random_isis import make
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5058
vec = make(seed, zero=True)
inst = H.Instance.from_vector(vec)
ctx = E.SpfContext(0)
got, want = H.compute_spf(H.Instance.from_vector(vec), ctx), R.local_rib(vec)
print("rib equal", got == want)
for a, b in zip(got, want):
    if a != b: print(" got", a, "\n want", b)
for level in inst.config.levels():
    for mt, hop in ((0, False), (2, False), (None, True)):
        if mt == 2 and not inst.config.mt_ipv6_unicast: continue
        g = H.LevelGraph(inst, level, mt, hop)
        if g.n == 0: continue
        roots = np.arange(g.n, dtype=np.uint32)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for fl in (g.run_flags, g.run_flags | E.RUN_POP_RANK):
            res = ctx.run(G, roots, fl)
            ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, fl & 3, go.MAP, mask_words_=res.first_hop_mask.shape[2])
            bad = {k: bool(not np.array_equal(x, y)) for k, x, y in (("dist", res.dist, ref.dist), ("hops", res.hops, ref.hops), ("mask", res.first_hop_mask, ref.mask), ("in", res.flags & 1, ref.flags))}
            if res.pop_rank is not None: bad["rank"] = bool(not np.array_equal(res.pop_rank, ref.pop_rank))
            print("level", level, "mt", mt, "hop", hop, "n", g.n, "flags", fl, "bad", bad, "stats", {k: res.stats[k] for k in ("n_exact_roots", "n_repaired_roots", "single_wg", "state_bytes")})
            if any(bad.values()):
                for k in bad:
                    if not bad[k]: continue
                    x = {"dist": res.dist, "hops": res.hops, "mask": res.first_hop_mask[..., 0], "in": res.flags & 1, "rank": res.pop_rank}[k]
                    y = {"dist": ref.dist, "hops": ref.hops, "mask": ref.mask[..., 0], "in": ref.flags, "rank": ref.pop_rank}[k]
                    r, v = np.argwhere(x != y)[0][:2]
                    print("   first", k, "root", r, "vertex", v, "got", x[r, v], "want", y[r, v])
                    print("   row_ptr", g.row_ptr.tolist(), "\n   col", g.col.tolist(), "\n   metric", g.metric.tolist(), "\n   vflags", g.vflags.tolist(), "maxpath", g.max_path_metric)
                    break
        G.free()
