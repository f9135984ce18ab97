"""Where the one-workgroup-per-root kernel (k_single) pays: 4-neighbour router grids of growing size, 1 / 64 / 1024 roots,
wall time of hspf_run_device with the kernel on (HSPF_SINGLE_MAX_N=8192) and off (=0).  Sets ctx->single_max_n's default.

    python tools/gpu_single_threshold.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def grid(rows, cols, chords=0):
    n = rows * cols
    links = synth._grid4_links(rows, cols)
    if chords:
        links = synth._add_chords(n, links, len(links) + chords, synth.SEED)
    return synth._routers_only(n, links, synth.SEED, 1, 100, synth.MAX_PATH_METRIC_OSPF, f"grid-{n}", {})


def main():
    import torch
    dev = torch.device("cuda:0")
    ctxs = {}
    for mode, v in (("single", "8192"), ("sweeps", "0")):
        os.environ["HSPF_SINGLE_MAX_N"] = v
        ctxs[mode] = E.SpfContext(0)
    shapes = [(20, 25, 0), (32, 32, 0), (32, 64, 0), (64, 64, 0), (64, 64, 400), (64, 128, 0), (64, 128, 800)]
    for rows, cols, ch in shapes:
        g = grid(rows, cols, ch)
        n = g.n
        for R in (1, 64, 1024):
            roots = ((np.arange(R, dtype=np.uint64) * n) // R).astype(np.uint32)
            rec = {"n": n, "chords": ch, "roots": R}
            outs = {}
            for mode, ctx in ctxs.items():
                G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
                W = G.mask_words(roots)
                d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
                f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
                wall, dv = [], []
                for it in range(12):
                    t0 = time.perf_counter()
                    st = ctx.run_device(G, roots, 1, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                        mask_ptr=m.data_ptr(), mask_words=W)
                    wall.append((time.perf_counter() - t0) * 1e3); dv.append(st["ms_total"])
                rec[mode] = {"wall_ms": round(float(np.median(wall[3:])), 4), "device_ms": round(float(np.median(dv[3:])), 4),
                             "single_wg": st["single_wg"], "launches": st["n_relax_launches"]}
                outs[mode] = (d.cpu().numpy().copy(), h.cpu().numpy().copy(), m.cpu().numpy().copy())
                G.free()
            rec["identical"] = all(np.array_equal(a, b) for a, b in zip(outs["single"], outs["sweeps"]))
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
