"""Device time of the other BASELINE configs (parity-test cases, not bench lines): configs[3] multi-area (10 areas x
1000 roots) and configs[4] fat-tree (101 roots, two mask words).  One JSON line each."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def run(ctx, g, roots, flags, reps=4):
    import torch
    dev = torch.device("cuda:0")
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.asarray(roots, np.uint32)
    W = G.mask_words(roots)
    R, n = len(roots), g.n
    dist = torch.empty((R, n), dtype=torch.int32, device=dev); hops = torch.empty((R, n), dtype=torch.int16, device=dev)
    fl = torch.empty((R, n), dtype=torch.int16, device=dev); mask = torch.empty((R, n, W), dtype=torch.int64, device=dev)
    ms = []
    for _ in range(reps):
        st = ctx.run_device(G, roots, flags, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=fl.data_ptr(),
                            mask_ptr=mask.data_ptr(), mask_words=W)
        ms.append(st["ms_total"])
    G.free()
    return float(np.median(ms[1:])), st, W


def main():
    ctx = E.SpfContext(0)
    tot, roots_n = 0.0, 0
    for g in synth.ospf_multi_area():
        ms, st, W = run(ctx, g, g.meta["roots"], E.RUN_NET_NEXTHOPS)
        tot += ms; roots_n += len(g.meta["roots"])
    print(json.dumps({"config": "ospf multi-area 10 x 5000 routers, 1000 roots per area", "roots": roots_n,
                      "device_ms_total": round(tot, 2), "runs_per_s": round(roots_n / tot * 1e3), "state_bytes": st["state_bytes"]}))
    # the same ten areas, each on its own context and host thread (the reference runs one OS thread per protocol instance,
    # holo-protocol/src/lib.rs:427-430; SURVEY.md §8b: distinct contexts work concurrently): wall time of the whole set
    import threading
    import time
    import torch
    areas = synth.ospf_multi_area()
    ctxs = [E.SpfContext(0) for _ in areas]
    dev = torch.device("cuda:0")
    jobs = []
    for c, g in zip(ctxs, areas):
        G = c.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        roots = np.asarray(g.meta["roots"], np.uint32)
        W = G.mask_words(roots)
        R, n = len(roots), g.n
        bufs = (torch.empty((R, n), dtype=torch.int32, device=dev), torch.empty((R, n), dtype=torch.int16, device=dev),
                torch.empty((R, n), dtype=torch.int16, device=dev), torch.empty((R, n, W), dtype=torch.int64, device=dev))
        jobs.append((c, G, roots, W, bufs))

    def work(j):
        c, G, roots, W, b = j
        c.run_device(G, roots, E.RUN_NET_NEXTHOPS, dist_ptr=b[0].data_ptr(), hops_ptr=b[1].data_ptr(), flags_ptr=b[2].data_ptr(),
                     mask_ptr=b[3].data_ptr(), mask_words=W)
    walls = {}
    for mode in ("sequential", "threads"):
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "sequential":
                for j in jobs:
                    work(j)
            else:
                th = [threading.Thread(target=work, args=(j,)) for j in jobs]
                for t in th: t.start()
                for t in th: t.join()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        walls[mode] = round(float(np.median(ts[1:])), 2)
    print(json.dumps({"config": "ospf multi-area, wall time of all 10 areas x 1000 roots", "one_context_after_the_other_ms": walls["sequential"],
                      "ten_contexts_on_ten_threads_ms": walls["threads"], "runs_per_s_threads": round(roots_n / walls["threads"] * 1e3)}))
    for c, G, *_ in jobs:
        G.free(); c.close()
    g = synth.isis_fattree(100)
    ms, st, W = run(ctx, g, g.meta["roots"], 0)
    print(json.dumps({"config": "isis fat-tree k=100 (262 500 vertices / 1.5 M entries), 101 roots", "mask_words": W,
                      "device_ms": round(ms, 2), "runs_per_s": round(101 / ms * 1e3), "relax_launches": st["n_relax_launches"],
                      "dag_launches": st["n_dag_launches"], "state_bytes": st["state_bytes"]}))


if __name__ == "__main__":
    main()
