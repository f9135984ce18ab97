"""More than 24 first-hop slots: the fused W-word fixed point (k_fw) against the older two-phase path (k_relax + k_dag,
HSPF_VARIANT bit6) on two shapes — a sparse graph whose roots sit on a big LAN (isis-100k plus one 48-router
pseudonode: 50-60 slots, one mask word) and the fat-tree of BASELINE configs[4] (100-link switch rows, two mask words).
Results of the two paths are compared with each other and, for a sample of roots, with the CPU oracle.

    python tools/gpu_wide_mask_timing.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def isis_100k_with_lan(members: int = 48):
    """isis-100k with every router shifted by one and vertex 0 = a pseudonode joining `members` routers (network
    vertices sort before routers in VertexId order: a zero-cost link out of the LAN then comes from a LOWER index and
    the static pop order holds)."""
    g = synth.isis_100k()
    n = g.n
    rp = g.row_ptr.astype(np.int64); col = g.col.astype(np.int64) + 1; met = g.metric.astype(np.int64)
    src = np.repeat(np.arange(n), np.diff(rp)) + 1
    mem = (np.arange(members, dtype=np.int64) * 2083 + 17) % n + 1        # the LAN's routers, spread over the graph
    s2 = np.concatenate([src, mem, np.zeros(members, np.int64)])
    d2 = np.concatenate([col, np.zeros(members, np.int64), mem])
    m2 = np.concatenate([met, np.full(members, 10), np.zeros(members, np.int64)])   # router -> LAN 10, LAN -> router 0
    row_ptr, c, mm = synth._csr_from_links(n + 1, s2, d2, m2)
    vf = np.zeros(n + 1, np.uint8); vf[0] = synth.VF_NETWORK
    return synth.CsrGraph(row_ptr, c, mm, vf, g.max_path_metric, "isis-100k+lan48", {}), mem.astype(np.uint32)


def main():
    import torch
    from oracle import graph_oracle as go
    dev = torch.device("cuda:0")
    cases = []
    g, mem = isis_100k_with_lan()
    cases.append((g, np.concatenate([mem, (np.arange(16) * 6007 % (g.n - 1) + 1).astype(np.uint32)])[:64], 0))
    if "--lan-only" not in sys.argv:
        ft = synth.isis_fattree(100)
        cases.append((ft, np.asarray(ft.meta["roots"], np.uint32), 0))
    for g, roots, flags in cases:
        outs = {}
        for mode, var in (("k_fw", "0"), ("two_phase", "64")):
            os.environ["HSPF_VARIANT"] = var
            ctx = E.SpfContext(0)
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            W = G.mask_words(roots)
            R, n = len(roots), g.n
            d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
            f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
            ms = []
            for it in range(6):
                st = ctx.run_device(G, roots, flags, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                    mask_ptr=m.data_ptr(), mask_words=W)
                ms.append(st["ms_total"])
            outs[mode] = (d.cpu().numpy().view(np.uint32), h.cpu().numpy().view(np.uint16), m.cpu().numpy().view(np.uint64))
            rec = {"graph": g.name, "roots": R, "mask_words": W, "path": mode, "device_ms": round(float(np.median(ms[2:])), 3),
                   "launches": st["n_relax_launches"] + st["n_dag_launches"], "exact_roots": st["n_exact_roots"]}
            if mode == "two_phase":
                rec["identical_to_k_fw"] = all(np.array_equal(a, b) for a, b in zip(outs["k_fw"], outs["two_phase"]))
                sample = list(range(0, R, max(1, R // 6)))
                o = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots[sample], flags, go.HEAP, mask_words_=W, threads=8)
                rec["identical_to_oracle_sample"] = bool(np.array_equal(outs[mode][0][sample], o.dist) and np.array_equal(outs[mode][1][sample], o.hops)
                                                         and np.array_equal(outs[mode][2][sample], o.mask))
            print(json.dumps(rec), flush=True)
            G.free(); ctx.close()


if __name__ == "__main__":
    main()
