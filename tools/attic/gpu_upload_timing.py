"""Wall-clock of hspf_graph_upload and hspf_graph_patch on the bench graphs (run on the GPU box).

    python tools/gpu_upload_timing.py            # prints one JSON line per graph
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402


def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def main():
    ctx = E.SpfContext(0)
    for name, g in (("ospf-500", synth.ospf_500()), ("ospf-10k", synth.ospf_10k()), ("isis-100k", synth.isis_100k()), ("isis-fattree-250k", synth.isis_fattree())):
        graphs = []

        def up():
            graphs.append(ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric))
        up()                                                    # warm-up: scratch allocation
        t_up = med(up, 7)
        G = graphs[-1]
        for x in graphs[:-1]:
            x.free()
        # link flap: the two rows of one link, metric changed and restored
        u = g.n // 3
        a, b = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
        rows = [(g.col[a:b], g.metric[a:b] + 1)], [(g.col[a:b], g.metric[a:b])]
        state = [0]

        def patch():
            G.patch([u], rows[state[0] & 1], [g.vflags[u]]); state[0] += 1
        patch(); patch()
        t_patch = med(patch, 20)
        # the C call alone (without the numpy mirror splice of SpfGraph.patch)
        import ctypes
        from holo_amd import _lib as L
        vs = np.array([u], np.uint32); rp = np.array([0, b - a], np.uint32)
        col = np.ascontiguousarray(g.col[a:b]); met = np.ascontiguousarray(g.metric[a:b]); vf = np.array([g.vflags[u]], np.uint8)
        r = L.HspfRows(1, vs.ctypes.data_as(L.u32p), rp.ctypes.data_as(L.u32p), col.ctypes.data_as(L.u32p),
                       met.ctypes.data_as(L.u32p), vf.ctypes.data_as(L.u8p))
        t_patch_c = med(lambda: ctx.lib.hspf_graph_patch(ctx.handle, G.handle, ctypes.byref(r)), 20)
        mode = int(G.export("build_mode")[0])                   # 2: costs changed in place, nothing rebuilt
        # a structural change: the row loses its last link and gets it back (splice + rebuild)
        srows = [(g.col[a:b - 1], g.metric[a:b - 1])], [(g.col[a:b], g.metric[a:b])]
        sstate = [0]

        def spatch():
            G.patch([u], srows[sstate[0] & 1], [g.vflags[u]]); sstate[0] += 1
        spatch(); spatch()
        t_spatch = med(spatch, 10)
        print(json.dumps({"graph": name, "n": g.n, "e": g.e, "upload_ms": round(t_up, 3),
                          "patch_1_row_ms": round(t_patch_c, 3), "patch_1_row_with_numpy_mirror_ms": round(t_patch, 3),
                          "patch_1_row_mode": mode, "patch_1_row_structural_with_numpy_mirror_ms": round(t_spatch, 3)}))
        G.free()


if __name__ == "__main__":
    main()
