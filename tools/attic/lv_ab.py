"""k_lv on isis-100k, one root: records / wake-up offsets from the ELL copy (HSPF_VARIANT bits 25 / 26) on ONE box."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from holo_amd import engine as E, synth
g = synth.isis_100k()
dev = torch.device("cuda:0")
roots = np.array([0], np.uint32)
d = torch.empty((1, g.n), dtype=torch.int32, device=dev); h = torch.empty((1, g.n), dtype=torch.int16, device=dev)
f = torch.empty((1, g.n), dtype=torch.int16, device=dev); m = torch.empty((1, g.n, 1), dtype=torch.int64, device=dev)
ctxs = {}
for name, var in (("ell + wake-ups", 0), ("ell records only", 1 << 25), ("round 4 form", 1 << 26)):
    os.environ["HSPF_VARIANT"] = str(var)
    ctxs[name] = E.SpfContext(0)
del os.environ["HSPF_VARIANT"]
for rep in range(3):
    for name, ctx in ctxs.items():
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        ms = []
        for it in range(20):
            st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=1)
            ms.append(st["ms_total"])
        print(json.dumps({"form": name, "device_ms": round(float(np.median(ms[4:])), 4), "launches": st["n_relax_launches"]}), flush=True)
        G.free()
