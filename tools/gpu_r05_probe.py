"""Round-5 A/B probe on one MI355X: the headline run (isis-100k x 64 roots, results in HBM) under a list of environment
settings — one step at a time and four steps in flight on the lanes — each checked against the CPU oracle.

    python tools/gpu_r05_probe.py NAME=ENV1=v,ENV2=v [NAME2=...]     (a bare NAME = default settings)
e.g. python tools/gpu_r05_probe.py base inner2=HSPF_DENSE_INNER=2 inner3=HSPF_DENSE_INNER=3
One JSON line per setting; HSPF_LIB=<path> in a setting picks another build of the library (own process needed: it is
read at import time, so such settings are run through a child process).
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(name):
    import torch
    from holo_amd import synth
    from holo_amd import engine as E
    from oracle import graph_oracle as go
    dev = torch.device("cuda:0")
    g = synth.isis_100k()
    n, R = g.n, 64
    roots = ((np.arange(R, dtype=np.int64) * n) // R).astype(np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1, threads=min(64, os.cpu_count() or 1))
    ctx = E.SpfContext(0)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    W = 1

    def bufs():
        return dict(dist=torch.empty((R, n), dtype=torch.int32, device=dev), hops=torch.empty((R, n), dtype=torch.int16, device=dev),
                    flags=torch.empty((R, n), dtype=torch.int16, device=dev), mask=torch.empty((R, n, W), dtype=torch.int64, device=dev))

    def kw(b):
        return dict(dist_ptr=b["dist"].data_ptr(), hops_ptr=b["hops"].data_ptr(), flags_ptr=b["flags"].data_ptr(), mask_ptr=b["mask"].data_ptr(), mask_words=W)

    def same(b):
        return bool(np.array_equal(b["dist"].cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(b["hops"].cpu().numpy().view(np.uint16), ref.hops)
                    and np.array_equal(b["flags"].cpu().numpy().view(np.uint16) & 1, ref.flags) and np.array_equal(b["mask"].cpu().numpy().view(np.uint64), ref.mask))
    b = [bufs() for _ in range(4)]
    for _ in range(6):
        st = ctx.run_device(G, roots, 0, **kw(b[0]))
    ok = same(b[0])
    reps = 150
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    devms = 0.0
    launches = 0
    for _ in range(reps):
        st = ctx.run_device(G, roots, 0, **kw(b[0]))
        devms += st["ms_total"]; launches += st["n_relax_launches"]
    one = (time.perf_counter() - t0) / reps * 1e3
    ok = ok and same(b[0])
    stc = ctx.run_device(G, roots, E.RUN_COUNT_ROWS, **kw(b[0]))
    rows_x_n = stc["rows_recomputed"] / n
    d = st["dbg"][1]
    plan = {"dense_used": d & 0xFF, "head_ran": (d >> 8) & 0xFF, "dense_planned": (d >> 16) & 0xFF, "head_planned": (d >> 24) & 0x7F}

    def in_flight(K, depth):
        ts = []
        t0 = time.perf_counter()
        for i in range(K):
            if len(ts) >= depth:
                ctx.wait(ts.pop(0))
            ts.append(ctx.run_device_async(G, roots, 0, **kw(b[i % 4])))
        while ts:
            ctx.wait(ts.pop(0))
        return (time.perf_counter() - t0) / K * 1e3
    in_flight(12, 4)
    fl = in_flight(240, 4)
    ok = ok and all(same(x) for x in b)
    out = {"name": name, "identical_to_oracle": ok, "one_at_a_time_ms": round(one, 4), "one_at_a_time_runs_per_s": round(R / one * 1e3),
           "device_ms": round(devms / reps, 4), "launches": round(launches / reps, 2), "rows_x_N": round(rows_x_n, 2), "plan": plan,
           "in_flight4_ms": round(fl, 4), "in_flight4_runs_per_s": round(R / fl * 1e3)}
    G.free(); ctx.close()
    return out


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        print(json.dumps(measure(sys.argv[2])), flush=True)
        return
    for spec in sys.argv[1:] or ["base"]:
        name, _, envs = spec.partition("=")
        env = dict(os.environ)
        env.setdefault("GPU_MAX_HW_QUEUES", "8")
        if envs:
            for kv in envs.split(","):
                k, _, v = kv.partition("=")
                env[k] = v
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], env=env, capture_output=True, text=True, timeout=600)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else json.dumps({"name": name, "error": p.stderr[-400:]})
        print(line, flush=True)


if __name__ == "__main__":
    main()
