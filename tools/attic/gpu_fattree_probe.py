"""configs[4] on one GPU (fat-tree k=100, 101 roots): device ms per run and the engine's own stats; under
rocprofv3 --kernel-trace --stats the per-kernel split.   python tools/gpu_fattree_probe.py [runs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth, engine as E     # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    g = synth.isis_fattree(100)
    roots = np.asarray(g.meta["roots"], np.uint32)
    ctx = E.SpfContext(0)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    W = G.mask_words(roots)
    R = len(roots)
    b = dict(dist=torch.empty((R, g.n), dtype=torch.int32, device=dev), hops=torch.empty((R, g.n), dtype=torch.int16, device=dev),
             flags=torch.empty((R, g.n), dtype=torch.int16, device=dev), mask=torch.empty((R, g.n, W), dtype=torch.int64, device=dev))
    kw = dict(dist_ptr=b["dist"].data_ptr(), hops_ptr=b["hops"].data_ptr(), flags_ptr=b["flags"].data_ptr(), mask_ptr=b["mask"].data_ptr(), mask_words=W)
    for i in range(runs):
        t0 = time.perf_counter()
        ctx.run_device(G, roots, 0, **kw)
        torch.cuda.synchronize()
        st = ctx.stats()
        print(f"run {i}: wall {(time.perf_counter() - t0) * 1e3:.3f} ms  stats {st}", flush=True)


if __name__ == "__main__":
    main()
