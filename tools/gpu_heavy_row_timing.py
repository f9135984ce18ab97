"""What one big LAN costs: isis-100k plus ONE pseudonode with D member routers (D in-links on the pseudonode's row, one
zero-cost in-link more on every member), the bench's 64 roots, device time per run.

    python tools/gpu_heavy_row_timing.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402
from oracle import graph_oracle as go          # noqa: E402


def with_lan(g, D, seed=5):
    """Vertex 0 becomes a pseudonode (network vertices sort first in the reference): every router index shifts by one."""
    n = g.n + 1
    src = np.repeat(np.arange(g.n, dtype=np.int64), np.diff(g.row_ptr.astype(np.int64))) + 1
    dst = g.col.astype(np.int64) + 1
    met = g.metric.astype(np.int64)
    rng = np.random.default_rng(seed)
    roots = (np.arange(64, dtype=np.int64) * (n - 1)) // 64 + 1        # main()'s roots stay off the LAN (a member root of a
    mem = rng.choice(np.setdiff1d(np.arange(1, n), roots), size=D, replace=False)   # 1 000-router LAN has > 1 024 slots)
    src = np.concatenate([src, mem, np.zeros(D, np.int64)])
    dst = np.concatenate([dst, np.zeros(D, np.int64), mem])
    met = np.concatenate([met, rng.integers(1, 101, D), np.zeros(D, np.int64)])
    row_ptr, col, metric = synth._csr_from_links(n, src, dst, met)
    vf = np.zeros(n, np.uint8); vf[0] = synth.VF_NETWORK
    return synth.CsrGraph(row_ptr, col, metric, vf, g.max_path_metric, f"isis-100k+lan{D}", {})


def main():
    import torch
    dev = torch.device("cuda:0")
    ctx = E.SpfContext(0)
    base = synth.isis_100k()
    for D in (0, 100, 900, 1000, 5000, 20000):
        g = with_lan(base, D) if D else base
        n = g.n
        roots = ((np.arange(64, dtype=np.int64) * (n - 1)) // 64 + 1).astype(np.uint32)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        W = G.mask_words(roots)
        d = torch.empty((64, n), dtype=torch.int32, device=dev); h = torch.empty((64, n), dtype=torch.int16, device=dev)
        f = torch.empty((64, n), dtype=torch.int16, device=dev); m = torch.empty((64, n, W), dtype=torch.int64, device=dev)
        ms = []
        for _ in range(6):
            st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                mask_ptr=m.data_ptr(), mask_words=W)
            ms.append(st["ms_total"])
        rec = {"graph": g.name, "lan_members": D, "mask_words": W, "device_ms": round(float(np.median(ms[2:])), 3),
               "launches": st["n_relax_launches"], "state_bytes": st["state_bytes"], "exact_roots": st["n_exact_roots"]}
        if D in (1000,):
            ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots[:8], 0, go.HEAP, mask_words_=W)
            rec["verified_8_roots"] = bool(np.array_equal(d[:8].cpu().numpy().view(np.uint32), ref.dist)
                                           and np.array_equal(m[:8].cpu().numpy().view(np.uint64), ref.mask))
        # one root (k_lv on a graph of this size) and, for a LAN the mask words can hold, 64 roots that are MEMBERS (k_fw)
        r1 = roots[:1]
        ms1 = []
        for _ in range(5):
            st = ctx.run_device(G, r1, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                mask_ptr=m.data_ptr(), mask_words=W)
            ms1.append(st["ms_total"])
        rec["one_root_ms"] = round(float(np.median(ms1[2:])), 3); rec["one_root_lv"] = st["lane_vertex"]
        if 0 < D <= 900:
            mem = g.col[g.row_ptr[0]:g.row_ptr[1]][:64].astype(np.uint32)
            Wm = G.mask_words(mem)
            mm = torch.empty((64, n, Wm), dtype=torch.int64, device=dev)
            msm = []
            for _ in range(5):
                st = ctx.run_device(G, mem, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                    mask_ptr=mm.data_ptr(), mask_words=Wm)
                msm.append(st["ms_total"])
            rec["member_roots_ms"] = round(float(np.median(msm[2:])), 3); rec["member_roots_words"] = Wm
        print(json.dumps(rec), flush=True)
        G.free()


if __name__ == "__main__":
    main()
