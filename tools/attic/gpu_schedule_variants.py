"""Schedule variants of the fused fixed point on the headline graph: rows recomputed, launches, device time
(VERDICT r01 item 1).  One context per HSPF_VARIANT value (the switch is read at hspf_init).  The r02a experiment
(profiles/r02a_schedule_variants.jsonl, profiles/r02_notes.md) had three more switches that were measured and removed:

    bit2 (4)   no precise wake-ups (the r01 schedule)
    bit3 (8)   alternating sweep directions
    bit4 (16)  precise wake-ups forced on in every sweep (a correctness mode, not a fast one)

What is left is the row counter (HSPF_RUN_COUNT_ROWS): each root set is timed without it and counted in one extra run.

Root sets: spread (the headline's floor(i*N/64)), clustered (a router and its 2-hop neighbourhood: LFA / MANET shape,
holo-isis/src/flooding/manet.rs:47-69), one root.  Every variant's results are compared bit for bit with the first
one's, and the first one's with the CPU oracle for all roots.

    python tools/gpu_schedule_variants.py [--graph isis-100k] [--reps 9]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def clustered_roots(g, centre: int, k: int) -> np.ndarray:
    seen, order = {centre}, [centre]
    i = 0
    while len(order) < k and i < len(order):
        u = order[i]; i += 1
        for v in g.col[g.row_ptr[u]:g.row_ptr[u + 1]]:
            if int(v) not in seen:
                seen.add(int(v)); order.append(int(v))
    return np.array(order[:k], np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="isis-100k")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--variants", default="0")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated root-set names (default: all)")
    a = ap.parse_args()
    import torch
    g = {"isis-100k": synth.isis_100k, "ospf-10k": synth.ospf_10k, "ospf-500": synth.ospf_500}[a.graph]()
    n = g.n
    dev = torch.device("cuda:0")
    root_sets = {"spread64": ((np.arange(64, dtype=np.uint64) * n) // 64).astype(np.uint32),
                 "clustered64": clustered_roots(g, n // 2 + 137, 64),
                 "one": np.array([n // 2 + 137], np.uint32)}
    ref = {}
    for var in [int(x) for x in a.variants.split(",")]:
        os.environ["HSPF_VARIANT"] = str(var)
        ctx = E.SpfContext(0)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for name, roots in root_sets.items():
            if a.only and name not in a.only.split(","):
                continue
            R = len(roots)
            W = G.mask_words(roots)
            dist = torch.empty((R, n), dtype=torch.int32, device=dev); hops = torch.empty((R, n), dtype=torch.int16, device=dev)
            flags = torch.empty((R, n), dtype=torch.int16, device=dev); mask = torch.empty((R, n, W), dtype=torch.int64, device=dev)
            ms, launches = [], 0
            for it in range(a.reps):
                st = ctx.run_device(G, roots, 0, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                                    mask_ptr=mask.data_ptr(), mask_words=W)
                ms.append(st["ms_total"]); launches = st["n_relax_launches"]
            stc = ctx.run_device(G, roots, E.RUN_COUNT_ROWS, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                                 mask_ptr=mask.data_ptr(), mask_words=W)
            rows = stc["rows_recomputed"]
            torch.cuda.synchronize()
            out = (dist.cpu().numpy().view(np.uint32), hops.cpu().numpy().view(np.uint16), flags.cpu().numpy().view(np.uint16),
                   mask.cpu().numpy().view(np.uint64))
            same = None
            if name in ref:
                same = all(np.array_equal(x, y) for x, y in zip(out, ref[name]))
            else:
                ref[name] = out
                if not a.no_oracle:
                    from oracle import graph_oracle as go
                    o = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=W)
                    same = bool(np.array_equal(out[0], o.dist) and np.array_equal(out[1], o.hops)
                                and np.array_equal(out[3], o.mask))
            print(json.dumps({"graph": a.graph, "variant": var, "roots": name, "device_ms_median": round(float(np.median(ms[2:])), 4),
                              "device_ms_min": round(float(min(ms[2:])), 4), "runs_per_s": round(R / (np.median(ms[2:]) * 1e-3)),
                              "launches": launches, "rows_recomputed": int(rows), "rows_x_N": round(rows / n, 2),
                              "counted_run_ms": round(stc["ms_total"], 4), "state_bytes": st["state_bytes"],
                              "bit_identical_to": "first variant" if var != int(a.variants.split(",")[0]) else "cpu oracle", "identical": same}),
                  flush=True)
            del dist, hops, flags, mask
        G.free()
        del ctx


if __name__ == "__main__":
    main()
