"""Where the wall time of a run goes on the host: hspf_stats.dbg[2] (entry -> first enqueue: slot tables, parameter
choice, upload block), dbg[3] (entry -> return), device time, wall time of the Python call.  isis-100k / 64 roots, ospf-500 / 1."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth
from holo_amd import engine as E
import torch
dev = torch.device("cuda:0")
ctx = E.SpfContext(0)
for g, roots in ((synth.isis_100k(), None), (synth.ospf_500(), [0])):
    n = g.n
    roots = np.asarray(roots if roots is not None else g.meta["roots"], np.uint32)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    R = len(roots); W = G.mask_words(roots)
    d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
    f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
    rec = []
    for it in range(40):
        t0 = time.perf_counter()
        st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
        w = (time.perf_counter() - t0) * 1e6
        rec.append((w, st["dbg"][2], st["dbg"][3], st["ms_total"] * 1e3))
    a = np.median(np.array(rec[10:]), axis=0)
    print(json.dumps({"graph": g.name, "roots": R, "wall_us": round(float(a[0]), 1), "host_before_first_enqueue_us": float(a[1]),
                      "host_entry_to_return_us": float(a[2]), "device_us": round(float(a[3]), 1)}))
    G.free()
