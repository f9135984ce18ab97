"""Two (or more) independent SPF instances on ONE GPU — each its own hspf_ctx and stream, driven by its own host thread,
64 roots per run on isis-100k: aggregate runs/s against one instance alone.  The sparse head and tail of a run leave
most of the chip idle; a second instance's dense sweeps fill it.  Run on the GPU box."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E        # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = synth.isis_100k()
    n = g.n
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for inst in (1, 2, 3, 4):
        ctxs, graphs, bufs, roots = [], [], [], []
        for i in range(inst):
            c = E.SpfContext(0)
            ctxs.append(c)
            graphs.append(c.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric))
            r = ((np.arange(64, dtype=np.int64) * n) // 64 + 7 * i).astype(np.uint32) % n     # each instance its own root set
            roots.append(r)
            bufs.append((torch.empty((64, n), dtype=torch.int32, device=dev), torch.empty((64, n), dtype=torch.int16, device=dev),
                         torch.empty((64, n), dtype=torch.int16, device=dev), torch.empty((64, n, 1), dtype=torch.int64, device=dev)))

        def loop(i, k):
            d, h, f, m = bufs[i]
            for _ in range(k):
                ctxs[i].run_device(graphs[i], roots[i], 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                   mask_ptr=m.data_ptr(), mask_words=1)
        for i in range(inst):
            loop(i, 5)
        torch.cuda.synchronize()
        ts = [threading.Thread(target=loop, args=(i, K)) for i in range(inst)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"instances": inst, "runs_per_s": round(64 * K * inst / dt), "ms_per_64_root_run": round(dt / (K * inst) * 1e3, 4)}))
        for i in range(inst):
            graphs[i].free(); ctxs[i].close()


if __name__ == "__main__":
    main()
