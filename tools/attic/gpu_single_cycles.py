"""Instrumented one-workgroup run (HSPF_RUN_COUNT_ROWS -> hspf_stats.dbg: sweeps, shader cycles, 100 MHz ticks, set-up
cycles of workgroup 0) on ospf-500, one root.  Run on the GPU box; HSPF_VARIANT is passed through."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E        # noqa: E402

if __name__ == "__main__":
    ctx = E.SpfContext(0)
    dev = torch.device("cuda:0")
    g = synth.ospf_500()
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.array([0], np.uint32)
    d = torch.empty((1, g.n), dtype=torch.int32, device=dev); h = torch.empty((1, g.n), dtype=torch.int16, device=dev)
    f = torch.empty((1, g.n), dtype=torch.int16, device=dev); m = torch.empty((1, g.n, 1), dtype=torch.int64, device=dev)
    for fl in (1, 1 | E.RUN_COUNT_ROWS, 1 | E.RUN_COUNT_ROWS, 1):
        ms = []
        for _ in range(10):
            st = ctx.run_device(G, roots, fl, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=1)
            ms.append(st["ms_total"])
        print(json.dumps({"flags": fl, "device_ms_median": round(float(np.median(ms)), 4), "dbg": list(st["dbg"]), "ms_relax": st["ms_relax"], "ms_finish": st["ms_finish"]}))
