"""Debug: the first run of one seed of tools/gpu_fuzz.py's fuzz() (same graph, same roots, same flags), repeated; where the tables differ from the oracle's."""
import os, sys, numpy as np
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R)
from holo_amd import synth, engine as E
from oracle import graph_oracle as go
seed = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(10_000 + seed)
nr = int(rng.integers(5, 260)) if rng.random() > 0.04 else int(rng.integers(800, 3000))
nn = int(rng.integers(0, 14)); hop = rng.random() < 0.2; zero = os.environ.get("FUZZ_ZERO") is not None
g = synth.random_lsdb(nr, nn, float(rng.uniform(1.2, 4.5)), 50_000 + seed, metric_lo=1, metric_hi=int(rng.integers(1, 5 if zero else 40)),
                      max_path=(1023 if rng.random() < 0.15 else (0xFFFFFFFF if rng.random() < 0.3 else synth.MAX_PATH_METRIC_WIDE)),
                      p_oneway=float(rng.choice([0.0, 0.03, 0.3])), p_parallel=float(rng.choice([0.0, 0.05, 0.4])),
                      p_overload=float(rng.choice([0.0, 0.03, 0.3])), p_noexpand=float(rng.choice([0.0, 0.02, 0.2])),
                      zero_cost_router_links=bool(rng.random() < 0.15) or zero, lan_size=int(rng.choice([2, 3, 5, 8, 14, 20, 30, 45, 70, 140])), hopcount=hop)
if rng.random() < 0.15 and not hop:
    g.metric = (g.metric.astype(np.uint64) << int(rng.integers(8, 25))).clip(0, 0xFFFFFFFE).astype(np.uint32)
k = int(rng.integers(1, min(g.n, 200 if g.n < 800 else 700) + 1))
roots = rng.choice(g.n, size=k, replace=rng.random() < 0.2).astype(np.uint32)
if k > 3 and rng.random() < 0.3:
    roots[int(rng.integers(0, k))] = E.NO_ROOT
flags = int(rng.choice([0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD, E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD]))
if hop: flags |= E.RUN_IGNORE_OVERLOAD
if rng.random() < (0.3 if zero else 0.08): flags |= E.RUN_POP_RANK
print("n", g.n, "links", len(g.col), "roots", k, "flags", flags, "hop", hop, "maxpath", hex(g.max_path_metric), "wmax", int(g.metric.max()), flush=True)
ctx = E.SpfContext(0)
G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
ref = None
for it in range(reps):
    res = ctx.run(G, roots, flags)
    if ref is None:
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, flags & 3, go.MAP, mask_words_=res.first_hop_mask.shape[2])
    bad = {"dist": res.dist != ref.dist, "hops": res.hops != ref.hops, "in": (res.flags & 1) != ref.flags, "mask": (res.first_hop_mask != ref.mask).any(axis=2)}
    line = {kk: int(v.sum()) for kk, v in bad.items()}
    st = res.stats
    print("run", it, line, {x: st[x] for x in ("n_exact_roots", "n_repaired_roots", "repair_sweeps", "repair_evals", "state_bytes", "single_wg", "lane_vertex")}, flush=True)
    for kk, v in bad.items():
        if v.any():
            rr, vv = np.argwhere(v)[0][:2]
            got = {"dist": res.dist, "hops": res.hops, "in": res.flags & 1, "mask": res.first_hop_mask[..., 0]}[kk]
            want = {"dist": ref.dist, "hops": ref.hops, "in": ref.flags, "mask": ref.mask[..., 0]}[kk]
            print("   first", kk, "root slot", int(rr), "root", int(roots[rr]), "vertex", int(vv), "got", int(got[rr, vv]), "want", int(want[rr, vv]),
                  "dist", int(ref.dist[rr, vv]), "exact flag", int(res.flags[rr, roots[rr]] & 2) if roots[rr] != E.NO_ROOT else None,
                  "bad roots", sorted(set(np.argwhere(v)[:, 0].tolist()))[:12], "bad vertices of that root", np.argwhere(v[rr])[:8, 0].tolist(), flush=True)
            a, b = int(g.row_ptr[vv]), int(g.row_ptr[vv + 1])
            print("   row of the vertex: targets", g.col[a:b].tolist(), "costs", g.metric[a:b].tolist(), "vflags", int(g.vflags[vv]))
            ins = [(int(u), int(g.metric[kx])) for u in range(g.n) for kx in range(int(g.row_ptr[u]), int(g.row_ptr[u + 1])) if g.col[kx] == vv]
            print("   links into it (source, cost, dist of source, hops of source, ref):", [(u, c, int(ref.dist[rr, u]), int(ref.hops[rr, u]), int(res.hops[rr, u])) for u, c in ins][:24])
            break
    if any(v.any() for v in bad.values()) and os.environ.get("STOP_AT_MISMATCH"):
        sys.exit(3)
