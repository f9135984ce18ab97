"""Graph build (hspf_graph_upload) with per-link row scans against the sorted-key build (hub mode), on a star of routers
of growing size and on isis-100k: wall time of the upload call, median of 5.

    python tools/gpu_hub_build_timing.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def star(n_leaves, seed=3):
    rng = np.random.default_rng(seed)
    n = n_leaves + 1
    hub = n // 3
    leaves = np.concatenate([np.arange(0, hub), np.arange(hub + 1, n)])
    a = rng.choice(leaves, size=2 * n_leaves); b = rng.choice(leaves, size=2 * n_leaves)
    ok = a != b
    src = np.concatenate([np.full(n_leaves, hub), leaves, a[ok], b[ok]])
    dst = np.concatenate([leaves, np.full(n_leaves, hub), b[ok], a[ok]])
    met = rng.integers(1, 9, len(src))
    row_ptr, col, metric = synth._csr_from_links(n, src, dst, met)
    return synth.CsrGraph(row_ptr, col, metric, np.zeros(n, np.uint8), synth.MAX_PATH_METRIC_WIDE, f"star-{n_leaves}", {})


def main():
    ctxs = {}
    for mode, v in (("sorted", "0"), ("scan", str(1 << 30))):
        os.environ["HSPF_HUB_DEG"] = v
        ctxs[mode] = E.SpfContext(0)
    graphs = [synth.isis_100k(), star(2_000), star(20_000), star(100_000), star(400_000)]
    for g in graphs:
        rec = {"graph": g.name, "n": g.n, "links": int(len(g.col)), "max_row": int(np.diff(g.row_ptr).max())}
        exp = {}
        for mode, ctx in ctxs.items():
            t = []
            for _ in range(5):
                t0 = time.perf_counter()
                G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
                t.append((time.perf_counter() - t0) * 1e3)
                exp[mode] = [G.export(k) for k in ("in_src", "in_cost", "in_pos", "twoway")]
                assert int(G.export("build_mode")[0]) == (mode == "sorted")
                G.free()
            rec[mode + "_ms"] = round(float(np.median(t)), 3)
        rec["identical"] = all(np.array_equal(a, b) for a, b in zip(exp["scan"], exp["sorted"]))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
