"""Round-4 probe on one MI355X: the device-side plan of the lean sweep (cold = steady?), asynchronous lanes with and
without the dense token, threshold sweep, tiny-graph latency.  Every result is compared with the CPU oracle.

    python tools/gpu_r04_probe.py [what ...]      what = plan lanes thresholds tiny (default: all)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402
from oracle import graph_oracle as go          # noqa: E402

import torch                                   # noqa: E402

DEV = torch.device("cuda:0")


def bufs(R, n, W):
    return dict(dist=torch.empty((R, n), dtype=torch.int32, device=DEV), hops=torch.empty((R, n), dtype=torch.int16, device=DEV),
                flags=torch.empty((R, n), dtype=torch.int16, device=DEV), mask=torch.empty((R, n, W), dtype=torch.int64, device=DEV))


def kw(b, W):
    return dict(dist_ptr=b["dist"].data_ptr(), hops_ptr=b["hops"].data_ptr(), flags_ptr=b["flags"].data_ptr(),
                mask_ptr=b["mask"].data_ptr(), mask_words=W)


def same(b, ref):
    return bool(np.array_equal(b["dist"].cpu().numpy().view(np.uint32), ref.dist) and
                np.array_equal(b["hops"].cpu().numpy().view(np.uint16), ref.hops) and
                np.array_equal(b["flags"].cpu().numpy().view(np.uint16) & 1, ref.flags) and
                np.array_equal(b["mask"].cpu().numpy().view(np.uint64), ref.mask))


def plan_of(st):
    d = st["dbg"][1]
    return {"dense_used": d & 0xFF, "head_ran": (d >> 8) & 0xFF, "dense_planned": (d >> 16) & 0xFF, "head_planned": (d >> 24) & 0x7F}


def env_ctx(**env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return E.SpfContext(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def sync_loop(ctx, G, roots, b, W, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev = 0.0
    for _ in range(reps):
        st = ctx.run_device(G, roots, 0, **kw(b, W))
        dev += st["ms_total"]
    return (time.perf_counter() - t0) / reps * 1e3, dev / reps, st


def main():
    what = set(sys.argv[1:]) or {"plan", "lanes", "thresholds", "tiny"}
    g = synth.isis_100k()
    n = g.n
    R = 64
    roots = ((np.arange(R, dtype=np.int64) * n) // R).astype(np.uint32)
    other = ((roots.astype(np.int64) + 777) % n).astype(np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1, threads=64)
    ref_other = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, other, 0, go.HEAP, mask_words_=1, threads=64)

    if "plan" in what:
        ctx = E.SpfContext(0)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        W = G.mask_words(roots)
        b = bufs(R, n, W)
        seq = []
        for i in range(6):
            t0 = time.perf_counter()
            st = ctx.run_device(G, roots, 0, **kw(b, W))
            wall = (time.perf_counter() - t0) * 1e3
            seq.append({"run": i, "wall_ms": round(wall, 4), "device_ms": round(st["ms_total"], 4), "launches": st["n_relax_launches"],
                        "plan": plan_of(st), "ok": same(b, ref)})
        wall, dev, st = sync_loop(ctx, G, roots, b, W, 200)
        out = {"probe": "plan", "first_runs": seq, "steady_sync": {"wall_ms": round(wall, 4), "device_ms": round(dev, 4), "runs_per_s": round(R / wall * 1e3), "plan": plan_of(st)}}
        # other roots, then back; then a structural one-row patch (a link removed on both sides), first run after it
        t0 = time.perf_counter(); st = ctx.run_device(G, other, 0, **kw(b, W)); w1 = (time.perf_counter() - t0) * 1e3
        out["other_roots_first"] = {"wall_ms": round(w1, 4), "device_ms": round(st["ms_total"], 4), "plan": plan_of(st), "ok": same(b, ref_other)}
        u = n // 3
        a0, b0 = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
        v = int(g.col[a0])
        c0, d0 = int(g.row_ptr[v]), int(g.row_ptr[v + 1])
        keep_u = np.arange(a0, b0)[1:]
        keep_v = np.array([k for k in range(c0, d0) if int(g.col[k]) != u])
        t0 = time.perf_counter()
        G.patch([u, v], [(g.col[keep_u], g.metric[keep_u]), (g.col[keep_v], g.metric[keep_v])], [g.vflags[u], g.vflags[v]])
        tp = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); st = ctx.run_device(G, roots, 0, **kw(b, W)); w2 = (time.perf_counter() - t0) * 1e3
        ref2 = go.run(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1, threads=64)
        out["after_structural_patch_first"] = {"patch_wall_ms": round(tp, 4), "wall_ms": round(w2, 4), "device_ms": round(st["ms_total"], 4), "plan": plan_of(st), "ok": same(b, ref2)}
        # a fresh context: its very first run (buffers cold), then its second
        ctx2 = E.SpfContext(0)
        G2 = ctx2.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        fr = []
        for i in range(3):
            t0 = time.perf_counter(); st = ctx2.run_device(G2, roots, 0, **kw(b, W)); w3 = (time.perf_counter() - t0) * 1e3
            fr.append({"run": i, "wall_ms": round(w3, 4), "device_ms": round(st["ms_total"], 4), "plan": plan_of(st), "ok": same(b, ref)})
        out["fresh_context"] = fr
        # stamped only (round 3's cold path) for comparison
        ctx3 = env_ctx(HSPF_VARIANT=524288)
        G3 = ctx3.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for _ in range(3):
            ctx3.run_device(G3, roots, 0, **kw(b, W))
        wall, dev, st = sync_loop(ctx3, G3, roots, b, W, 100)
        out["stamped_only"] = {"wall_ms": round(wall, 4), "device_ms": round(dev, 4), "launches": st["n_relax_launches"], "ok": same(b, ref)}
        print(json.dumps(out), flush=True)
        G.free(); G2.free(); G3.free(); ctx.close(); ctx2.close(); ctx3.close()

    if "thresholds" in what:
        W = 1
        b = bufs(R, n, W)
        for enter, stay, head in ((30, 10, 4), (30, 30, 4), (30, 45, 4), (30, 60, 4), (30, 75, 4), (30, 90, 4), (60, 45, 4), (60, 60, 5)):
            ctx = env_ctx(HSPF_DENSE_PCT=enter, HSPF_DENSE_STAY_PCT=stay, HSPF_LEAN_HEAD=head)
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            t0 = time.perf_counter(); st0 = ctx.run_device(G, roots, 0, **kw(b, W)); first = (time.perf_counter() - t0) * 1e3
            for _ in range(4):
                ctx.run_device(G, roots, 0, **kw(b, W))
            wall, dev, st = sync_loop(ctx, G, roots, b, W, 150)
            print(json.dumps({"probe": "thresholds", "enter_pct": enter, "stay_pct": stay, "head0": head, "first_wall_ms": round(first, 4), "first_plan": plan_of(st0),
                              "wall_ms": round(wall, 4), "device_ms": round(dev, 4), "runs_per_s": round(R / wall * 1e3), "launches": st["n_relax_launches"],
                              "plan": plan_of(st), "ok": same(b, ref)}), flush=True)
            G.free(); ctx.close()

    if "lanes" in what:
        W = 1
        for lanes, token, lds in [tuple(int(x) for x in c.split(',')) for c in os.environ.get('LANE_CONFIGS', '3,0,0;3,0,0;2,0,0;4,0,0').split(';')]:
            ctx = env_ctx(HSPF_ASYNC_LANES=lanes, HSPF_DENSE_STREAM=token, HSPF_DENSE_LDS_KB=lds)
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            nb = lanes + 1
            bb = [bufs(R, n, W) for _ in range(nb)]

            def pipeline(reps):
                tickets, last = [], None
                for i in range(reps):
                    tickets.append(ctx.run_device_async(G, roots, 0, **kw(bb[i % nb], W)))
                    if len(tickets) >= lanes:                       # `lanes` runs in flight while the host turns around
                        last = ctx.wait(tickets.pop(0))
                while tickets:
                    last = ctx.wait(tickets.pop(0))
                return last
            pipeline(12)
            torch.cuda.synchronize()
            reps = 300
            t0 = time.perf_counter()
            st = pipeline(reps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = all(same(x, ref) for x in bb)
            print(json.dumps({"probe": "lanes", "hwq": os.environ.get("GPU_MAX_HW_QUEUES", "default"), "lanes": lanes, "dense_stream": token, "lds_kb": lds, "runs_per_s": round(reps * R / dt), "ms_per_run": round(dt / reps * 1e3, 4),
                              "device_ms_last": round(st["ms_total"], 4), "plan": plan_of(st), "ok": ok}), flush=True)
            G.free(); ctx.close()

    if "tiny" in what:
        ctx = E.SpfContext(0)
        for side in (5, 8, 10, 12, 14, 16, 20):
            nn = side * side
            gg = synth._routers_only(nn, synth._grid4_links(side, side), synth.SEED, 1, 100, synth.MAX_PATH_METRIC_OSPF, "g", {})
            G = ctx.upload(gg.row_ptr, gg.col, gg.metric, gg.vflags, gg.max_path_metric)
            r1 = np.array([0], np.uint32)
            b = bufs(1, nn, 1)
            for _ in range(5):
                ctx.run_device(G, r1, 0, **kw(b, 1))
            wall, dev, st = sync_loop(ctx, G, r1, b, 1, 300)
            rf = go.run(gg.row_ptr, gg.col, gg.metric, gg.vflags, gg.max_path_metric, r1, 0, go.HEAP, mask_words_=1)
            print(json.dumps({"probe": "tiny", "n": nn, "gpu_wall_ms": round(wall, 4), "gpu_device_ms": round(dev, 4), "single_wg": st["single_wg"],
                              "recommend_cpu": int(ctx.lib.hspf_recommend_cpu(nn, int(gg.e), 1)), "ok": same(b, rf)}), flush=True)
            G.free()
        ctx.close()


if __name__ == "__main__":
    main()
