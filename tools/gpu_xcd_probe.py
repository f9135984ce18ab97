"""k_xcd (one XCD per root, the state replicated in every CU's LDS, one launch) against the launch-per-sweep engine on
mid-size graphs: device and wall time of hspf_run_device for 1 / 2 / 4 / 8 roots, every result compared with the oracle;
"product" = the default context, which tries both once per graph and root count and then runs the faster one.
Run on the GPU box:   python tools/gpu_xcd_probe.py [> gpurun_out/xcd_probe.jsonl]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from holo_amd import engine as E, synth        # noqa: E402
from oracle import graph_oracle as go          # noqa: E402


def contexts():
    out = {}
    for name, env in (("xcd", {"HSPF_XCD_ALWAYS": "1"}), ("sweeps", {"HSPF_XCD_MAX_ROOTS": "0"}), ("product", {})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        out[name] = E.SpfContext(0)
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return out


def graphs():
    yield "ospf-10k", synth.ospf_10k(), 1
    yield "grid-2500", synth._routers_only(2500, synth._grid4_links(50, 50), 5, 1, 100, synth.MAX_PATH_METRIC_OSPF, "grid-2500", {}), 1
    yield "lsdb-5k+300net", synth.random_lsdb(5000, 300, 3.0, 77, metric_hi=60, lan_size=6), 1
    yield "lsdb-18k+800net", synth.random_lsdb(18000, 800, 3.2, 78, metric_hi=60, lan_size=8, p_overload=0.02), 1
    yield "isis-hop-6k", synth.random_lsdb(6000, 200, 3.0, 79, hopcount=True, lan_size=5), 2


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    ctxs = contexts()
    for name, g, fl in graphs():
        rng = np.random.default_rng(3)
        for k in (1, 2, 4, 8):
            roots = np.concatenate([[0 if not (g.vflags[0] & 1) else int(np.flatnonzero(~(g.vflags & 1).astype(bool))[0])],
                                    rng.choice(g.n, size=k - 1, replace=False)]).astype(np.uint32)
            ref = None
            row = {"graph": name, "n": int(g.n), "links": int(len(g.col)), "roots": k}
            for cname, ctx in ctxs.items():
                G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
                W = G.mask_words(roots)
                d = torch.empty((k, g.n), dtype=torch.int32, device=dev); h = torch.empty((k, g.n), dtype=torch.int16, device=dev)
                f = torch.empty((k, g.n), dtype=torch.int16, device=dev); m = torch.empty((k, g.n, W), dtype=torch.int64, device=dev)
                wall, devms = [], []
                for it in range(14):
                    t0 = time.perf_counter()
                    st = ctx.run_device(G, roots, fl, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                        mask_ptr=m.data_ptr(), mask_words=W)
                    wall.append((time.perf_counter() - t0) * 1e3); devms.append(st["ms_total"])
                if ref is None:
                    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, fl & 3, go.MAP, mask_words_=W)
                ok = bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref.hops)
                          and np.array_equal(f.cpu().numpy().view(np.uint16) & 1, ref.flags) and np.array_equal(m.cpu().numpy().view(np.uint64), ref.mask))
                row[cname] = {"device_ms": round(float(np.median(devms[3:])), 4), "wall_ms": round(float(np.median(wall[3:])), 4),
                              "launches": st["n_relax_launches"] + st["n_dag_launches"], "path": bench.path_of(st),
                              "sweeps": (st["dbg"][1] & 0xFFFF) if st["single_wg"] == 2 else None,
                              "off_xcd": bool(st["dbg"][1] >> 31) if st["single_wg"] == 2 else None,
                              "exact_roots": st["n_exact_roots"], "identical_to_oracle": ok}
                G.free()
            print(json.dumps(row), flush=True)
