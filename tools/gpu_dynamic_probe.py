#!/usr/bin/env python3
"""Timing of runs whose roots have a dynamic pop order (zero-cost links): isis-100k / ospf-10k with a share of their link
entries at metric 0, results in HBM (hspf_run_device), against the same graph without them and against the CPU heap.
    python tools/gpu_dynamic_probe.py [--seq]      (--seq: also the sequential kernel, HSPF_VARIANT bit 27 — slow)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import engine as E, synth        # noqa: E402


def zero_links(g, share, seed):
    rng = np.random.default_rng(seed)
    m = g.metric.copy()
    m[rng.random(len(m)) < share] = 0
    return synth.CsrGraph(g.row_ptr, g.col, m, g.vflags, g.max_path_metric, g.name)


def timed(ctx, g, roots, reps=10):
    dev = torch.device("cuda:0")
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    R, n = len(roots), g.n
    W = G.mask_words(roots)
    t = dict(dist=torch.empty((R, n), dtype=torch.int32, device=dev), hops=torch.empty((R, n), dtype=torch.int16, device=dev),
             flags=torch.empty((R, n), dtype=torch.int16, device=dev), mask=torch.empty((R, n, W), dtype=torch.int64, device=dev))
    kw = dict(dist_ptr=t["dist"].data_ptr(), hops_ptr=t["hops"].data_ptr(), flags_ptr=t["flags"].data_ptr(), mask_ptr=t["mask"].data_ptr(), mask_words=W)
    for _ in range(3):
        st = ctx.run_device(G, roots, 0, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        st = ctx.run_device(G, roots, 0, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    G.free()
    return ms, st


def main():
    ctx = E.SpfContext(0)
    out = []
    for name, g0, rootsets in (("isis-100k", synth.isis_100k(), {"64": None, "1": [0]}), ("ospf-10k", synth.ospf_10k(), {"64": None, "1": [0]})):
        for share in ((0.0, 0.01, 0.05) if "--quick" in sys.argv else (0.0, 0.001, 0.01, 0.05)):
            g = zero_links(g0, share, 11) if share else g0
            for rn, roots in rootsets.items():
                roots = (np.arange(64, dtype=np.uint64) * g.n // 64).astype(np.uint32) if roots is None else np.asarray(roots, np.uint32)
                ms, st = timed(ctx, g, roots)
                rec = {"graph": name, "zero_share": share, "roots": len(roots), "ms_per_run_call": round(ms, 4), "runs_per_s": round(len(roots) / ms * 1e3, 1),
                       "n_repaired_roots": st["n_repaired_roots"], "n_exact_roots": st["n_exact_roots"], "ms_repair": round(st["ms_repair"], 4),
                       "repair_sweeps": st["repair_sweeps"], "repair_evals": st["repair_evals"], "repair_groups": st["repair_groups"],
                       "ms_total_device": round(st["ms_total"], 4)}
                out.append(rec)
                print(json.dumps(rec), flush=True)
    ctx.close()
    if "--seq" in sys.argv:
        os.environ["HSPF_VARIANT"] = str(1 << 27)
        ctx = E.SpfContext(0)
        g = zero_links(synth.ospf_10k(), 0.01, 11)
        for roots in ([0], list(range(0, 10000, 1250))):
            t0 = time.perf_counter()
            ms, st = timed(ctx, g, np.asarray(roots, np.uint32), reps=1)
            print(json.dumps({"graph": "ospf-10k", "zero_share": 0.01, "roots": len(roots), "sequential_kernel_ms": round(ms, 2), "n_exact_roots": st["n_exact_roots"]}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
