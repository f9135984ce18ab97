"""Where the lane = vertex kernels pay: the BASELINE graphs with 1 .. 1024 roots, device and wall time of
hspf_run_device with k_lv forced on (HSPF_LV_MAX_ROOTS=64, HSPF_LV_MIN_N=0) and off (the lane = root sweep engine),
k_single off in both.  Sets the defaults of ctx->lv_max_roots / lv_min_n.

    python tools/gpu_lv_threshold.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def main():
    import torch
    dev = torch.device("cuda:0")
    ctxs = {}
    envs = {"lv": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "64", "HSPF_LV_MIN_N": "0"},
            "sweeps": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "0"}}
    for mode, env in envs.items():
        os.environ.update(env)
        ctxs[mode] = E.SpfContext(0)
    def grid(rows, cols, links, name):
        n = rows * cols
        return synth._routers_only(n, synth._add_chords(n, synth._grid8_links(rows, cols), links, synth.SEED), synth.SEED, 1, 100,
                                   synth.MAX_PATH_METRIC_WIDE, name, {})
    graphs = [synth.ospf_500(), synth.ospf_10k(), grid(125, 200, 125000, "grid-25k"), grid(200, 250, 250000, "grid-50k"),
              synth.isis_100k(), grid(500, 800, 2000000, "grid-400k")]
    if "--fattree" in sys.argv:
        graphs.append(synth.isis_fattree())
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--graph=")]
    rlist = [int(a.split("=", 1)[1]) for a in sys.argv if a.startswith("--roots=")] or [1, 2, 4, 8, 64]
    for g in graphs:
        if only and g.name not in only:
            continue
        n = g.n
        fl = 1 if g.name.startswith("ospf") else 0
        for R in rlist:
            roots = ((np.arange(R, dtype=np.uint64) * n) // R).astype(np.uint32)
            if g.name.startswith("isis-fattree"):
                roots = np.arange(R, dtype=np.uint32) + np.uint32(n - 1000)       # hosts: one first-hop slot each
            rec = {"graph": g.name, "n": n, "roots": R}
            outs = {}
            for mode, ctx in ctxs.items():
                G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
                W = G.mask_words(roots)
                d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
                f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
                wall, dv = [], []
                for it in range(12):
                    t0 = time.perf_counter()
                    st = ctx.run_device(G, roots, fl, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                        mask_ptr=m.data_ptr(), mask_words=W)
                    wall.append((time.perf_counter() - t0) * 1e3); dv.append(st["ms_total"])
                st2 = ctx.run_device(G, roots, fl | E.RUN_COUNT_ROWS, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(),
                                     flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
                rec[mode] = {"wall_ms": round(float(np.median(wall[3:])), 4), "device_ms": round(float(np.median(dv[3:])), 4),
                             "lane_vertex": st["lane_vertex"], "single_wg": st["single_wg"], "launches": st["n_relax_launches"],
                             "exact_roots": st["n_exact_roots"], "rows": int(st2["rows_recomputed"]), "sweeps": int(st2["dbg"][0])}
                outs[mode] = (d.cpu().numpy().copy(), h.cpu().numpy().copy(), m.cpu().numpy().copy())
                G.free()
            rec["identical"] = all(all(np.array_equal(a, b) for a, b in zip(outs[m], outs["sweeps"])) for m in outs)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
