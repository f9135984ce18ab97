"""configs[4] alone (isis fat-tree k=100, 101 roots) through hspf_run_device: device ms per run and the stats of the
last run; meant to sit under `rocprofv3 --kernel-trace --stats` (run on the GPU box)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    ctx = E.SpfContext(0)
    dev = torch.device("cuda:0")
    g = synth.isis_fattree(100)
    roots = np.asarray(g.meta["roots"], np.uint32)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    W = G.mask_words(roots)
    R, n = len(roots), g.n
    d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
    f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
    ms = []
    for _ in range(reps):
        st = ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                            mask_ptr=m.data_ptr(), mask_words=W)
        ms.append(st["ms_total"])
    print(json.dumps({"device_ms": [round(x, 3) for x in ms], "W": W, "stats": {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in st.items()}}))
    if os.environ.get("HSPF_CHECK"):
        from oracle import graph_oracle as go
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=W, threads=32)
        ok = bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref.hops)
                  and np.array_equal(m.cpu().numpy().view(np.uint64), ref.mask))
        print(json.dumps({"identical_to_oracle": ok}))


if __name__ == "__main__":
    main()
