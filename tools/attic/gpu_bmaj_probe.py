"""Batch-major XCD placement of the dense stretch (k_fused_lean, calls of >= 8 batches): device ms of ospf-10k x 1024 roots and
of the ten areas of configs[3] (5 000 routers x 1 000 roots each), every root compared with the oracle.
    python tools/gpu_bmaj_probe.py            HSPF_VARIANT=16777216 python tools/gpu_bmaj_probe.py   (row-major, as before)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E     # noqa: E402
from oracle import graph_oracle as go       # noqa: E402


def one(ctx, g, roots, fl, check=True):
    dev = torch.device("cuda:0")
    roots = np.asarray(roots, np.uint32)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    W = G.mask_words(roots)
    R, n = len(roots), g.n
    d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
    f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
    ms = []
    for _ in range(7):
        st = ctx.run_device(G, roots, fl, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
        ms.append(st["ms_total"])
    ok = None
    if check:
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, fl & 3, go.HEAP, mask_words_=W, threads=min(64, os.cpu_count() or 1))
        ok = bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref.hops)
                  and np.array_equal(m.cpu().numpy().view(np.uint64), ref.mask))
    G.free()
    return float(np.median(ms[2:])), st, ok


def main():
    ctx = E.SpfContext(0)
    g = synth.ospf_10k()
    roots = (np.arange(1024, dtype=np.int64) * g.n // 1024).astype(np.uint32)
    ms, st, ok = one(ctx, g, roots, E.RUN_NET_NEXTHOPS)
    print(f"ospf-10k x 1024 roots: {ms:.3f} ms  launches {st['n_relax_launches']}  dense passes {st['dbg'][1] & 0xFF}  identical {ok}", flush=True)
    tot, allok = 0.0, True
    for i, a in enumerate(synth.ospf_multi_area()):
        ms, st, ok = one(ctx, a, a.meta["roots"], E.RUN_NET_NEXTHOPS, check=i < 3)
        tot += ms; allok = allok and (ok is not False)
        if i == 0:
            print(f"area 0: {ms:.3f} ms  launches {st['n_relax_launches']}  dense passes {st['dbg'][1] & 0xFF}", flush=True)
    print(f"ten areas x 1000 roots: {tot:.3f} ms  identical (first three areas) {allok}", flush=True)


if __name__ == "__main__":
    main()
