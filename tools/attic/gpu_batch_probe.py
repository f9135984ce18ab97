import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
from holo_amd import synth, engine as E
from oracle import graph_oracle as go
import torch
dev = torch.device("cuda:0")
g = synth.isis_100k(); n = g.n
ctx = E.SpfContext(0)
G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
base = ((np.arange(64, dtype=np.int64) * n) // 64)
for K in (1, 2, 3, 4, 6, 8, 16):
    roots = np.concatenate([(base + 131 * k) % n for k in range(K)]).astype(np.uint32) if K > 1 else base.astype(np.uint32)
    if os.environ.get("SAME"): roots = np.tile(base, K).astype(np.uint32)
    W = G.mask_words(roots); R = len(roots)
    b = dict(dist=torch.empty((R, n), dtype=torch.int32, device=dev), hops=torch.empty((R, n), dtype=torch.int16, device=dev),
             flags=torch.empty((R, n), dtype=torch.int16, device=dev), mask=torch.empty((R, n, W), dtype=torch.int64, device=dev))
    kw = dict(dist_ptr=b["dist"].data_ptr(), hops_ptr=b["hops"].data_ptr(), flags_ptr=b["flags"].data_ptr(), mask_ptr=b["mask"].data_ptr(), mask_words=W)
    for _ in range(5): st = ctx.run_device(G, roots, 0, **kw)
    torch.cuda.synchronize(); reps = max(10, 200 // K); t0 = time.perf_counter(); dv = 0
    for _ in range(reps):
        st = ctx.run_device(G, roots, 0, **kw); dv += st["ms_total"]
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"K": K, "roots": R, "W": W, "wall_ms": round(dt * 1e3, 4), "device_ms": round(dv / reps, 4), "runs_per_s": round(R / dt), "launches": st["n_relax_launches"],
                      "state_bytes": st["state_bytes"], "lean": st["dbg"][0], "dense_used": st["dbg"][1] & 255}), flush=True)
