import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench
from holo_amd import engine as E, synth
def mk(env):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    c = E.SpfContext(0)
    for k, v in old.items():
        if v is None: del os.environ[k]
        else: os.environ[k] = v
    return c
ctxs = {"single": mk({"HSPF_XCD_MAX_ROOTS": "0", "HSPF_SINGLE_MAX_N": "8192"}), "xcd": mk({"HSPF_SINGLE_MAX_N": "0", "HSPF_XCD_ALWAYS": "1"}), "sweeps": mk({"HSPF_SINGLE_MAX_N": "0", "HSPF_XCD_MAX_ROOTS": "0"})}
dev = torch.device("cuda:0")
for nr, nn in ((300, 10), (600, 20), (1000, 30), (1500, 40), (2000, 60), (3000, 80), (4000, 100)):
    g = synth.random_lsdb(nr, nn, 3.0, 500 + nr, metric_hi=40, lan_size=5)
    for k in (1, 4):
        roots = (np.arange(k, dtype=np.uint32) * 37 + nn).astype(np.uint32)
        row = {"n": int(g.n), "roots": k}
        for name, ctx in ctxs.items():
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            W = G.mask_words(roots)
            d = torch.empty((k, g.n), dtype=torch.int32, device=dev); h = torch.empty((k, g.n), dtype=torch.int16, device=dev)
            f = torch.empty((k, g.n), dtype=torch.int16, device=dev); m = torch.empty((k, g.n, W), dtype=torch.int64, device=dev)
            wall, devms = [], []
            for it in range(14):
                t0 = time.perf_counter()
                st = ctx.run_device(G, roots, 1, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
                wall.append((time.perf_counter() - t0) * 1e3); devms.append(st["ms_total"])
            row[name] = [round(float(np.median(devms[3:])), 4), round(float(np.median(wall[3:])), 4), bench.path_of(st)[:8]]
            G.free()
        print(json.dumps(row), flush=True)
