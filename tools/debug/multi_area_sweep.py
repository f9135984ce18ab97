"""configs[3] (one of its ten areas: 5 000 routers, 1 000 roots = 16 batches): device time under the lean sweep's launch
plan switches, one context per setting.  Tuning probe: python tools/debug/multi_area_sweep.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from holo_amd import engine as E, synth
g = next(iter(synth.ospf_multi_area()))
roots = np.asarray(g.meta["roots"], np.uint32)
dev = torch.device("cuda:0")
R, n = len(roots), g.n
d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, 1), dtype=torch.int64, device=dev)
settings = [{}] + [{"HSPF_DENSE_PCT": v} for v in (5, 15, 50)] + [{"HSPF_LEAN_HEAD": v} for v in (1, 2, 8)] + \
           [{"HSPF_DENSE_PASSES": v} for v in (4, 8, 32)] + [{"HSPF_DENSE_MIN_WGS": v} for v in (512, 2048, 100000)] + \
           [{"HSPF_DENSE_STAY_PCT": v} for v in (2, 25)] + [{"HSPF_VARIANT": 1 << 19}]
for env in settings:
    env = {k: str(v) for k, v in env.items()}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ctx = E.SpfContext(0)
    for k, v in old.items():
        if v is None: del os.environ[k]
        else: os.environ[k] = v
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    ms = []
    for it in range(7):
        st = ctx.run_device(G, roots, 1, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=1)
        ms.append(st["ms_total"])
    dd = st["dbg"][1]
    print(json.dumps({"env": env, "device_ms": round(float(np.median(ms[2:])), 4), "launches": st["n_relax_launches"],
                      "dense_used": dd & 0xFF, "head_ran": (dd >> 8) & 0xFF, "dense_planned": (dd >> 16) & 0xFF}), flush=True)
    G.free(); ctx.close()
