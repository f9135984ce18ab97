"""What does a chord cost?  The fused sweep on three 100 000-router graphs with the same 8-neighbour 250 x 400 grid:
no chords, 101 948 chords confined to +-2 000 vertices (they stay inside an XCD's range), and isis-100k's uniform chords.
Prints device time per row evaluation for HSPF_VARIANT 32768 (k_fused) and 0 (default).  (profiles/r03_notes.md, r03c)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth
from holo_amd import engine as E


def local_chords(n, links, total, seed, span):
    need = total - len(links)
    r = synth.splitmix64(seed, 4 * need, 3)
    a = (r[0::2] % np.uint64(n)).astype(np.int64)
    d = (r[1::2] % np.uint64(2 * span)).astype(np.int64) - span
    b = np.clip(a + d, 0, n - 1)
    ok = np.abs(a - b) > 401
    a, b = a[ok][:need], b[ok][:need]
    return np.concatenate([links, np.stack([np.minimum(a, b), np.maximum(a, b)], axis=1)])


def main():
    import torch
    n = 100000
    grid = synth._grid8_links(250, 400)
    graphs = {
        "grid only": synth._routers_only(n, grid, synth.SEED, 1, 100, synth.MAX_PATH_METRIC_WIDE, "g0", {}),
        "grid + local chords": synth._routers_only(n, local_chords(n, grid, 500000, synth.SEED, 2000), synth.SEED, 1, 100, synth.MAX_PATH_METRIC_WIDE, "g1", {}),
        "isis-100k": synth.isis_100k(),
    }
    dev = torch.device("cuda:0")
    roots = ((np.arange(64, dtype=np.uint64) * n) // 64).astype(np.uint32)
    for var in (32768, 0):
        os.environ["HSPF_VARIANT"] = str(var)
        ctx = E.SpfContext(0)
        for name, g in graphs.items():
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            W = G.mask_words(roots)
            dist = torch.empty((64, n), dtype=torch.int32, device=dev); hops = torch.empty((64, n), dtype=torch.int16, device=dev)
            flags = torch.empty((64, n), dtype=torch.int16, device=dev); mask = torch.empty((64, n, W), dtype=torch.int64, device=dev)
            ms = []
            for it in range(8):
                st = ctx.run_device(G, roots, 0, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
                ms.append(st["ms_relax"])
            stc = ctx.run_device(G, roots, E.RUN_COUNT_ROWS, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
            rows = stc["rows_recomputed"]
            print(json.dumps({"variant": var, "graph": name, "entries": int(g.e), "sweep_ms": round(float(np.median(ms[2:])), 4), "launches": st["n_relax_launches"],
                              "rows_x_N": round(rows / n, 2), "ns_per_row": round(float(np.median(ms[2:])) * 1e6 / max(rows, 1), 2), "state_bytes": st["state_bytes"], "lean": st["dbg"][0] if "dbg" in st else None}), flush=True)
            G.free()
        del ctx


if __name__ == "__main__":
    main()
