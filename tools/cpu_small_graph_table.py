"""One root on small 4-neighbour grids, one host core: the reference-SHAPED loop (ordered map + linear candidate scan +
per-edge two-way rescan, oracle variant REF), the indexed map and the binary heap.  The numbers behind
hspf_recommend_cpu (include/holo_spf_hip.h) and INTEGRATION.md section 6; `overhead` = the same call on a 4-vertex
graph (ctypes + the oracle's thread start), subtracted.

    python tools/cpu_small_graph_table.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from oracle import graph_oracle as go          # noqa: E402


def time_one(g, variant, reps):
    R = go.Runner(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, 1, 1)
    roots = np.array([0], np.uint32)
    for _ in range(5):
        R.run(roots, 0, variant)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps):
            R.run(roots, 0, variant)
        best = min(best, (time.perf_counter() - t) / reps * 1e3)
    return best


def main():
    tiny = synth._routers_only(4, synth._grid4_links(2, 2), synth.SEED, 1, 100, synth.MAX_PATH_METRIC_OSPF, "g", {})
    over = {v: time_one(tiny, v, 400) for v in (go.REF, go.MAP, go.HEAP)}
    for side in (5, 7, 10, 12, 14, 16, 18, 20, 22, 25, 32):
        n = side * side
        g = synth._routers_only(n, synth._grid4_links(side, side), synth.SEED, 1, 100, synth.MAX_PATH_METRIC_OSPF, "g", {})
        row = {"n": n, "entries": int(g.e)}
        for name, v in (("ref_shaped_ms", go.REF), ("map_ms", go.MAP), ("heap_ms", go.HEAP)):
            row[name] = round(max(time_one(g, v, 200 if n < 400 else 60) - over[v], 0.0), 4)
        row["model_ms"] = round(2.4e-4 * n + 7e-7 * n * n, 4)
        print(json.dumps(row))


if __name__ == "__main__":
    main()
