"""Device / wall time of hspf_run_device-style runs on the bench graphs for 1, 64 and 1024 roots.

    python tools/gpu_latency_table.py [path/to/alternative/libholo_spf_hip.so]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import _lib                      # noqa: E402
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def main():
    import torch
    ctx = E.SpfContext(0)
    dev = torch.device("cuda:0")
    for name, g in (("ospf-500", synth.ospf_500()), ("ospf-10k", synth.ospf_10k()), ("isis-100k", synth.isis_100k())):
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        row = {"graph": name}
        for R in (1, 64, 1024):
            roots = (np.arange(R, dtype=np.uint64) * g.n // R).astype(np.uint32)
            W = G.mask_words(roots)
            dist = torch.empty((R, g.n), dtype=torch.int32, device=dev); hops = torch.empty((R, g.n), dtype=torch.int16, device=dev)
            flags = torch.empty((R, g.n), dtype=torch.int16, device=dev); mask = torch.empty((R, g.n, W), dtype=torch.int64, device=dev)
            ms, wall, launches = [], [], 0
            for it in range(7):
                t0 = time.perf_counter()
                st = ctx.run_device(G, roots, 0, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                                    mask_ptr=mask.data_ptr(), mask_words=W)
                wall.append((time.perf_counter() - t0) * 1e3)
                ms.append(st["ms_total"]); launches = st["n_relax_launches"] + st["n_dag_launches"]
            row[f"r{R}"] = {"device_ms": round(float(np.median(ms[2:])), 3), "wall_ms": round(float(np.median(wall[2:])), 3),
                            "launches": launches, "state_bytes": st["state_bytes"]}
            del dist, hops, flags, mask
        G.free()
        print(json.dumps(row))


if __name__ == "__main__":
    main()
