"""Mid-size graphs (ospf-10k, the multi-area config): device ms of 1 / 64 / 1024 roots for a few HSPF_DENSE_MIN_WGS /
HSPF_DENSE_STAY_PCT settings; every result compared with the oracle."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E
from oracle import graph_oracle as go
import torch
dev = torch.device("cuda:0")
g = synth.ospf_10k()
n = g.n
for env in ({}, {"HSPF_DENSE_PERSIST": "0"}, {"HSPF_DENSE_STAY_PCT": "3"}, {"HSPF_LEAN_HEAD": "2", "HSPF_DENSE_STAY_PCT": "3"}):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ctx = E.SpfContext(0)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    for R in (1, 64, 1024):
        roots = ((np.arange(R, dtype=np.int64) * n) // R).astype(np.uint32)
        W = G.mask_words(roots)
        b = dict(dist=torch.empty((R, n), dtype=torch.int32, device=dev), hops=torch.empty((R, n), dtype=torch.int16, device=dev),
                 flags=torch.empty((R, n), dtype=torch.int16, device=dev), mask=torch.empty((R, n, W), dtype=torch.int64, device=dev))
        kw = dict(dist_ptr=b["dist"].data_ptr(), hops_ptr=b["hops"].data_ptr(), flags_ptr=b["flags"].data_ptr(), mask_ptr=b["mask"].data_ptr(), mask_words=W)
        ms = []
        for _ in range(16):
            st = ctx.run_device(G, roots, 1, **kw); ms.append(st["ms_total"])
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 1, go.HEAP, mask_words_=W, threads=64)
        ok = bool(np.array_equal(b["dist"].cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(b["hops"].cpu().numpy().view(np.uint16), ref.hops)
                  and np.array_equal(b["mask"].cpu().numpy().view(np.uint64), ref.mask))
        d = st["dbg"][1]
        print(json.dumps({"env": env, "roots": R, "device_ms": round(float(np.median(ms[8:])), 4), "launches": st["n_relax_launches"], "lean": st["dbg"][0],
                          "dense_used": d & 255, "head": (d >> 8) & 255, "planned": (d >> 16) & 255, "lv": st["lane_vertex"], "ok": ok}), flush=True)
    G.free(); ctx.close()
