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


def with_lan(g, D, seed=5, second=0):
    """Vertex 0 becomes a pseudonode with D members (network vertices sort first in the reference): every router index
    shifts.  `second`: one more pseudonode whose members are main()'s first `second` roots (roots with many slots)."""
    k = 2 if second else 1
    n = g.n + k
    src = np.repeat(np.arange(g.n, dtype=np.int64), np.diff(g.row_ptr.astype(np.int64))) + k
    dst = g.col.astype(np.int64) + k
    met = g.metric.astype(np.int64)
    rng = np.random.default_rng(seed)
    roots = (np.arange(64, dtype=np.int64) * (n - k)) // 64 + k          # main()'s roots stay off the big LAN (a member root
    mem = rng.choice(np.setdiff1d(np.arange(k, n), roots), size=D, replace=False)   # of a 1 000-router LAN has > 1 024 slots)
    src = np.concatenate([src, mem, np.zeros(D, np.int64)])
    dst = np.concatenate([dst, np.zeros(D, np.int64), mem])
    met = np.concatenate([met, rng.integers(1, 101, D), np.zeros(D, np.int64)])
    if second:
        m2 = roots[:second]
        src = np.concatenate([src, m2, np.ones(second, np.int64)])
        dst = np.concatenate([dst, np.ones(second, np.int64), m2])
        met = np.concatenate([met, rng.integers(1, 101, second), np.zeros(second, np.int64)])
    row_ptr, col, metric = synth._csr_from_links(n, src, dst, met)
    vf = np.zeros(n, np.uint8); vf[:k] = synth.VF_NETWORK
    return synth.CsrGraph(row_ptr, col, metric, vf, g.max_path_metric, f"isis-100k+lan{D}" + (f"+lan{second}" if second else ""), {"k": k})


def main():
    import torch
    dev = torch.device("cuda:0")
    ctx = E.SpfContext(0)
    os.environ["HSPF_VARIANT"] = "8192"                  # the same engine with giant rows walked whole (no slices)
    ctx_whole = E.SpfContext(0)
    del os.environ["HSPF_VARIANT"]
    base = synth.isis_100k()
    for D, second in ((0, 0), (100, 0), (900, 0), (1000, 0), (5000, 0), (20000, 0), (1000, 40), (5000, 40)):
        g = with_lan(base, D, second=second) if D else base
        n = g.n
        k = g.meta.get("k", 1)
        roots = ((np.arange(64, dtype=np.int64) * (n - k)) // 64 + k).astype(np.uint32)   # `second`: the first 40 share a LAN
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
        if 0 < D <= 900 and not second:
            mem = g.col[g.row_ptr[0]:g.row_ptr[1]][:64].astype(np.uint32)
            Wm = G.mask_words(mem)
            mm = torch.empty((64, n, Wm), dtype=torch.int64, device=dev)
            msm = []
            for _ in range(5):
                st = ctx.run_device(G, mem, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                    mask_ptr=mm.data_ptr(), mask_words=Wm)
                msm.append(st["ms_total"])
            rec["member_roots_ms"] = round(float(np.median(msm[2:])), 3); rec["member_roots_words"] = Wm
        if second or D >= 900:
            Gw = ctx_whole.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            Ww = Gw.mask_words(roots)
            mw = torch.empty((64, n, Ww), dtype=torch.int64, device=dev)
            msw = []
            for _ in range(4):
                st = ctx_whole.run_device(Gw, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                          mask_ptr=mw.data_ptr(), mask_words=Ww)
                msw.append(st["ms_total"])
            rec["rows_walked_whole_ms"] = round(float(np.median(msw[1:])), 3)
            Gw.free()
        print(json.dumps(rec), flush=True)
        G.free()


if __name__ == "__main__":
    main()
