#!/usr/bin/env python3
"""bench.py — full-SPF runs/sec on the 100k-vertex synthetic LSDB (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: 64 concurrent SPF roots on the
isis-100k graph (BASELINE.json configs[2]: IS-IS L2, 100 000 routers, 1 000 000 directed
IS-reachability entries, metrics U[1,100]) through the C ABI (hspf_run_device): distances, hops
(first-discoverer rule) and ECMP first-hop masks for every (root, vertex), written row-major
into HBM.  The graph is uploaded (resident in HBM) before the timed region; results stay in HBM.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): roots are independent units over a replicated
read-only graph, so every rank runs its own 64 roots per step (weak scaling).  The sharding and the exchange are the C
ABI's (hspf_multi_run): whole 64-root batches per rank, ONE RCCL all-gather of the per-root distance tables per step
(librccl.so loaded by the library; the communicator id travels over torch.distributed), issued asynchronously on a
communication stream so that it overlaps the next step's kernels.  At N = 1 the same entry point runs
a one-rank job.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW"


def b_alg(n: int, e: int, m: int) -> int:
    """Algorithmic bytes of ONE SPF run: row_ptr + (col,metric) per entry read
    once + one 8+8m-byte result record per vertex written once."""
    return 4 * (n + 1) + 8 * e + (8 + 8 * m) * n


def roots_for_rank(n: int, rank: int, world: int, per_rank: int = 64) -> np.ndarray:
    """Weak scaling: 64 roots per GPU per step.  The job's root list is floor(i*N/(64*world)) and
    rank g takes its contiguous slice (holo_amd.shard: whole 64-root batches per rank)."""
    from holo_amd import shard
    total = per_rank * world
    all_roots = ((np.arange(total, dtype=np.int64) * n) // total).astype(np.uint32)
    return shard.shard_roots(all_roots, rank, world)


def cpu_baseline(g, roots, budget_s: float = 8.0) -> dict:
    """The oracle's heap variant (a reasonable CPU implementation with identical outputs) on a bounded sample of the
    same workload (the 64 roots, cycled): ~budget_s on ONE thread (`value`, `cores` = 1) and ~budget_s with whole
    roots dealt to every host core (`all_cores`).  Also, for context only, the reference-SHAPED
    variant (ordered map + linear candidate scan + per-link two-way rescan, holo-isis/src/spf.rs:552-706) on the
    10k-router graph: its candidate scan makes it quadratic, one run at 100k vertices takes minutes.  Checker code is
    only timed here."""
    from oracle import graph_oracle as go
    from holo_amd import synth
    go.build()
    cores = os.cpu_count() or 1
    per_call = min(1024, max(64, 2 * cores))
    rn = go.Runner(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, per_call, 1)   # result arrays allocated once
    big = np.resize(roots, per_call)
    rn.run(big, threads=cores)                    # touches every page of the result arrays
    t0 = time.perf_counter()
    rn.run(roots[:4])
    one = (time.perf_counter() - t0) / 4
    k = int(max(8, budget_s / max(one, 1e-6)))
    sample = np.resize(roots, k)
    t0 = time.perf_counter()
    for a in range(0, k, 64):
        rn.run(sample[a:a + 64])
    dt = time.perf_counter() - t0
    # all cores: whole roots dealt to threads, calls of 2 x cores roots until ~budget_s
    ka, ta = 0, 0.0
    t0 = time.perf_counter()
    while ta < budget_s:
        rn.run(big, threads=cores)
        ka += per_call
        ta = time.perf_counter() - t0
    g10 = synth.ospf_10k()
    t1 = time.perf_counter()
    go.run(g10.row_ptr, g10.col, g10.metric, g10.vflags, g10.max_path_metric, np.array([0], np.uint32), 1, go.REF, mask_words_=1)
    dref = time.perf_counter() - t1
    stored = None
    sp = os.path.join(ROOT, "profiles", "r03_cpu_ref_shaped_100k.json")
    if os.path.exists(sp):
        try:
            j = json.load(open(sp))
            stored = {"runs_per_s": j["runs_per_s"], "seconds_per_run": j["seconds"], "identical_to_heap_variant": j["identical_to_heap_variant"],
                      "note": "STORED figure (profiles/r03_cpu_ref_shaped_100k.json, measured once in the build container, 1 thread): the "
                              "reference-shaped variant on THIS workload's graph, one root; not re-timed here (11 minutes per run)"}
        except Exception:   # noqa: BLE001
            stored = None
    return {"value": round(k / dt, 3), "unit": "spf_runs/s", "cores": 1, "kind": "port",
            "sample": f"oracle heap-Dijkstra restatement (dist+hops+first-hop masks), {k} runs cycling the 64 roots of "
                      f"isis-100k, 1 thread, {dt:.2f} s; host has {cores} cores",
            "all_cores": {"value": round(ka / ta, 3), "unit": "spf_runs/s", "cores": cores,
                          "sample": f"same restatement, {ka} runs (the 64 roots cycled), whole roots dealt to {cores} threads, {ta:.2f} s"},
            "reference_shaped_isis_100k": stored,
            "reference_shaped_ospf_10k": {"runs_per_s": round(1.0 / dref, 3),
                                          "note": "ordered-map + linear-scan shape of the reference loop (holo-isis/src/spf.rs:552-706), 10k routers / 80k entries, "
                                                  "1 root, 1 thread; the candidate scan is quadratic: ~100x this time per run at 100k vertices"}}


def latency_1root(ctx, dev) -> dict:
    """One root per run — what run_area / compute_spt is called with (holo-ospf/src/spf.rs:540-542): wall time of
    hspf_run_device including its stream synchronisation, median of 15 after 3 warm-up runs, next to the oracle's heap
    variant on one host thread."""
    import torch
    from holo_amd import synth
    from oracle import graph_oracle as go
    out = {}
    for name, g in (("ospf-500", synth.ospf_500()), ("ospf-10k", synth.ospf_10k()), ("isis-100k", synth.isis_100k())):
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        roots = np.array([0], np.uint32)
        fl = 1 if name.startswith("ospf") else 0
        W = G.mask_words(roots)
        d = torch.empty((1, g.n), dtype=torch.int32, device=dev); h = torch.empty((1, g.n), dtype=torch.int16, device=dev)
        f = torch.empty((1, g.n), dtype=torch.int16, device=dev); m = torch.empty((1, g.n, W), dtype=torch.int64, device=dev)
        wall, devms, launches = [], [], 0
        for it in range(18):
            t0 = time.perf_counter()
            st = ctx.run_device(G, roots, fl, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                mask_ptr=m.data_ptr(), mask_words=W)
            wall.append((time.perf_counter() - t0) * 1e3); devms.append(st["ms_total"])
            launches = st["n_relax_launches"] + st["n_dag_launches"]
        tc = []
        for it in range(5):
            t0 = time.perf_counter()
            ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, fl, go.HEAP, mask_words_=W)
            tc.append((time.perf_counter() - t0) * 1e3)
        ok = bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref.hops)
                  and np.array_equal(m.cpu().numpy().view(np.uint64), ref.mask))
        out[name] = {"gpu_wall_ms": round(float(np.median(wall[3:])), 4), "gpu_device_ms": round(float(np.median(devms[3:])), 4),
                     "launches": launches,
                     "path": path_of(st), "cpu_heap_1thread_ms": round(float(min(tc)), 4), "identical_to_oracle": ok}
        G.free()
    return out


def two_instances(g, dev, steps_each: int = 150) -> dict:
    """TWO independent SPF instances on the one GPU (what a router with IS-IS level 1 and level 2, or two OSPF areas,
    is): each its own hspf_ctx, stream and host thread, each running 64-root batches of the headline workload back to
    back (its own root set).  The sparse head and tail of a run leave most of the chip idle (28 launches, 10 of them
    under 12 us); the other instance's dense sweeps fill it.  Reported NEXT TO the headline, never as it: `value` stays
    one instance, one batch in flight.  The last run of both instances is compared with the oracle, every (root, vertex)."""
    import threading
    import torch
    from holo_amd import engine as E
    from oracle import graph_oracle as go
    n = g.n
    inst = []
    for i in range(2):
        c = E.SpfContext(dev.index or 0)
        G = c.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        r = (((np.arange(64, dtype=np.int64) * n) // 64 + 7 * i) % n).astype(np.uint32)
        b = (torch.empty((64, n), dtype=torch.int32, device=dev), torch.empty((64, n), dtype=torch.int16, device=dev),
             torch.empty((64, n), dtype=torch.int16, device=dev), torch.empty((64, n, 1), dtype=torch.int64, device=dev))
        inst.append((c, G, r, b))

    def loop(i, k):
        c, G, r, (d, h, f, m) = inst[i]
        for _ in range(k):
            c.run_device(G, r, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=1)
    for i in range(2):
        loop(i, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(0, steps_each); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    ts = [threading.Thread(target=loop, args=(i, steps_each)) for i in range(2)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    ok = True
    for c, G, r, (d, h, f, m) in inst:
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, r, 0, go.HEAP, mask_words_=1, threads=min(64, os.cpu_count() or 1))
        ok = ok and bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref.hops)
                         and np.array_equal(m.cpu().numpy().view(np.uint64), ref.mask))
        G.free(); c.close()
    one, two = 64 * steps_each / t1, 64 * 2 * steps_each / t2
    ba = b_alg(n, g.e, 1)
    return {"instances": 2, "roots_per_run": 64, "runs_each": steps_each, "runs_per_s": round(two), "one_instance_same_loop_runs_per_s": round(one),
            "ms_per_64_root_run": round(t2 / (2 * steps_each) * 1e3, 4), "whole_run_frac": round(two * ba / HBM_PEAK, 5),
            "identical_to_oracle": ok}


def pipeline_1root(ctx, dev) -> dict:
    """What one LSP refresh with changed costs costs end to end for one root on the headline graph (SURVEY.md 8f-1, 8f-2,
    8f-4 chained): hspf_graph_patch of the router's row -> hspf_run_device -> hspf_routes_device over 120 000 prefixes
    -> hspf_routes_diff_device against the previous table -> hspf_routes_pack (the changed routes, one copy to the
    host).  Wall time of the five C calls, median of 20 after 3 warm-ups, alternating between two cost sets; the SPT,
    the route table and the record stream of the last iteration are checked against the oracle and a numpy fold."""
    import torch
    from holo_amd import synth
    from holo_amd import engine as E
    from oracle import graph_oracle as go
    g = synth.isis_100k()
    n = g.n
    rng = np.random.default_rng(12)
    n2 = 20000                                            # prefixes 0 .. n-1: one owner each; then n2 with two owners
    two = np.sort(rng.integers(0, n, (n2, 2)), axis=1)
    two[:, 1] = np.where(two[:, 1] == two[:, 0], (two[:, 0] + 1) % n, two[:, 1]); two.sort(axis=1)
    vtx = np.concatenate([np.arange(n), two.reshape(-1)]).astype(np.uint32)
    ptr = np.concatenate([np.arange(n), n + 2 * np.arange(n2 + 1)]).astype(np.uint32)
    met = np.concatenate([np.arange(n) % 7, rng.integers(0, 10, 2 * n2)]).astype(np.uint32)
    P = n + n2
    roots = np.array([0], np.uint32)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    W = G.mask_words(roots)
    d = torch.empty((1, n), dtype=torch.int32, device=dev); h = torch.empty((1, n), dtype=torch.int16, device=dev)
    f = torch.empty((1, n), dtype=torch.int16, device=dev); m = torch.empty((1, n, W), dtype=torch.int64, device=dev)
    sets = [(torch.empty((1, P), dtype=torch.int32, device=dev), torch.empty((1, P), dtype=torch.int32, device=dev),
             torch.empty((1, P, W), dtype=torch.int64, device=dev)) for _ in range(2)]
    act = torch.empty((1, P), dtype=torch.uint8, device=dev); chg = torch.empty((P,), dtype=torch.int32, device=dev)
    cptr = torch.empty((2,), dtype=torch.int32, device=dev)
    u = n // 3
    a, b = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
    costs = [g.metric[a:b] + 5, g.metric[a:b].copy()]
    col_u, vf_u = g.col[a:b].copy(), [g.vflags[u]]

    def spt_and_routes(k):
        ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
        ctx.routes_device(n, 1, W, d.data_ptr(), f.data_ptr(), m.data_ptr(), ptr, vtx, met, best_metric_ptr=sets[k][0].data_ptr(),
                          best_entry_ptr=sets[k][1].data_ptr(), nexthop_mask_ptr=sets[k][2].data_ptr())
    spt_and_routes(0)
    cur, stages, rec = 0, [], None
    for it in range(23):
        t = [time.perf_counter()]
        G.patch([u], [(col_u, costs[it & 1])], vf_u); t.append(time.perf_counter())
        ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
        t.append(time.perf_counter())
        ctx.routes_device(n, 1, W, d.data_ptr(), f.data_ptr(), m.data_ptr(), ptr, vtx, met, best_metric_ptr=sets[cur ^ 1][0].data_ptr(),
                          best_entry_ptr=sets[cur ^ 1][1].data_ptr(), nexthop_mask_ptr=sets[cur ^ 1][2].data_ptr(),
                          flags=E.PFX_RESIDENT)                  # the prefix table did not change: a metric refresh, not a new prefix
        t.append(time.perf_counter())
        ctx.routes_diff_device(1, P, W, tuple(x.data_ptr() for x in sets[cur]), tuple(x.data_ptr() for x in sets[cur ^ 1]),
                               action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
        t.append(time.perf_counter())
        rec = ctx.routes_pack(1, P, W, tuple(x.data_ptr() for x in sets[cur ^ 1]), action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(),
                              changed_ptr_ptr=cptr.data_ptr())
        t.append(time.perf_counter())
        stages.append(np.diff(t) * 1e3)
        cur ^= 1
    mode = int(G.export("build_mode")[0])
    # ---- check of the last iteration (it = 22: costs[0], before it costs[1])
    def fold(metric, row_ptr=None, col=None):
        ref = go.run(g.row_ptr if row_ptr is None else row_ptr, g.col if col is None else col, metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=W)
        dd = ref.dist[0].astype(np.uint64); reach = (ref.flags[0] & 1) != 0
        cand = np.where(reach[vtx], (dd[vtx] + met) & 0xFFFFFFFF, np.uint64(1) << 40)
        bm = np.minimum.reduceat(cand, ptr[:-1].astype(np.int64))
        tie = cand == np.repeat(bm, np.diff(ptr.astype(np.int64)))
        mk = np.where(tie[:, None] & reach[vtx][:, None], ref.mask[0][vtx], 0).astype(np.uint64)
        nh = np.bitwise_or.reduceat(mk, ptr[:-1].astype(np.int64), axis=0)
        return ref, bm, nh
    m_new = g.metric.copy(); m_new[a:b] = costs[0]
    (_, bm_old, nh_old), (ref, bm_new, nh_new) = fold(g.metric), fold(m_new)
    ok_spt = bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref.hops)
                  and np.array_equal(m.cpu().numpy().view(np.uint64), ref.mask))
    got_bm = sets[cur][0].cpu().numpy().view(np.uint32)[0]; got_nh = sets[cur][2].cpu().numpy().view(np.uint64)[0]
    has = bm_new < (1 << 40)
    ok_routes = bool(np.array_equal(got_bm[has], bm_new[has].astype(np.uint32)) and np.array_equal(got_nh[has], nh_new[has]))
    want = np.nonzero(((bm_old != bm_new) | (nh_old != nh_new).any(axis=1)) & has & nh_new.any(axis=1))[0]
    ok_rec = bool(np.array_equal(rec[:, 1], want.astype(np.uint32)) and np.array_equal(rec[:, 3], bm_new[want].astype(np.uint32))
                  and np.array_equal(rec[:, 6:].copy().view(np.uint64).reshape(len(rec), W), nh_new[want]))
    # ---- the same chain behind a STRUCTURAL change: the router's last link withdrawn / announced again, so
    # the row changes its length, every later row moves and the link's far end loses / regains its two-way partner
    rows2 = [(col_u[:-1].copy(), g.metric[a:b][:-1].copy()), (col_u, g.metric[a:b].copy())]
    G.patch([u], [rows2[1]], vf_u)
    spt_and_routes(cur)
    stages2, rec2, calls2 = [], None, []
    for it in range(13):
        t = [time.perf_counter()]
        G.patch([u], [rows2[it & 1]], vf_u); t.append(time.perf_counter()); calls2.append(G.last_patch_call_ms)
        ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(), mask_ptr=m.data_ptr(), mask_words=W)
        t.append(time.perf_counter())
        ctx.routes_device(n, 1, W, d.data_ptr(), f.data_ptr(), m.data_ptr(), ptr, vtx, met, best_metric_ptr=sets[cur ^ 1][0].data_ptr(),
                          best_entry_ptr=sets[cur ^ 1][1].data_ptr(), nexthop_mask_ptr=sets[cur ^ 1][2].data_ptr(), flags=E.PFX_RESIDENT)
        t.append(time.perf_counter())
        ctx.routes_diff_device(1, P, W, tuple(x.data_ptr() for x in sets[cur]), tuple(x.data_ptr() for x in sets[cur ^ 1]),
                               action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
        t.append(time.perf_counter())
        rec2 = ctx.routes_pack(1, P, W, tuple(x.data_ptr() for x in sets[cur ^ 1]), action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(),
                               changed_ptr_ptr=cptr.data_ptr())
        t.append(time.perf_counter())
        stages2.append(np.diff(t) * 1e3)
        cur ^= 1
    mode2 = int(G.export("build_mode")[0])
    # last iteration (it = 12): the link withdrawn, before it the original rows
    (_, bm_o, nh_o), (ref2, bm_w, nh_w) = fold(g.metric), fold(G.metric, G.row_ptr, G.col)
    ok_spt2 = bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref2.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref2.hops)
                   and np.array_equal(m.cpu().numpy().view(np.uint64), ref2.mask))
    got_bm = sets[cur][0].cpu().numpy().view(np.uint32)[0]; got_nh = sets[cur][2].cpu().numpy().view(np.uint64)[0]
    has2 = bm_w < (1 << 40)
    ok_routes2 = bool(np.array_equal(got_bm[has2], bm_w[has2].astype(np.uint32)) and np.array_equal(got_nh[has2], nh_w[has2]))
    want2 = np.nonzero(((bm_o != bm_w) | (nh_o != nh_w).any(axis=1)) & has2 & nh_w.any(axis=1))[0]
    ok_rec2 = bool(np.array_equal(rec2[:, 1], want2.astype(np.uint32)) and np.array_equal(rec2[:, 3], bm_w[want2].astype(np.uint32))
                   and np.array_equal(rec2[:, 6:].copy().view(np.uint64).reshape(len(rec2), W), nh_w[want2]))
    st2 = np.median(np.array(stages2[3:]), axis=0)
    structural = {"change": "the router's last link withdrawn / announced again (row length, every later row's position and the far end's two-way partner change)",
                  "wall_ms": round(float(np.median(np.array(stages2[3:]).sum(axis=1))), 4),
                  "stages_ms": {k: round(float(v), 4) for k, v in zip(("graph_patch", "run_device", "routes_device", "routes_diff_device", "routes_pack"), st2)},
                  "graph_patch_c_call_ms": round(float(np.median(calls2[3:])), 4),
                  "graph_patch_is": "stages_ms.graph_patch = the Python twin's G.patch (C call + splice of its numpy mirrors of 1 M links); graph_patch_c_call_ms = hspf_graph_patch alone",
                  "patch_mode": {0: "rebuild", 1: "rebuild (hub)", 2: "costs in place", 3: "affected rows re-derived, arrays shifted"}[mode2], "records_to_host": int(len(rec2)),
                  "spt_identical_to_oracle": ok_spt2, "routes_identical_to_fold": ok_routes2, "records_identical_to_fold": ok_rec2}
    G.free()
    st = np.median(np.array(stages[3:]), axis=0)
    return {"graph": "isis-100k", "roots": 1, "prefixes": int(P), "prefix_entries": int(len(vtx)), "changed_row": int(u), "structural": structural,
            "wall_ms": round(float(np.median(np.array(stages[3:]).sum(axis=1))), 4),
            "stages_ms": {k: round(float(v), 4) for k, v in zip(("graph_patch", "run_device", "routes_device", "routes_diff_device", "routes_pack"), st)},
            "patch_mode": {0: "rebuild", 1: "rebuild (hub)", 2: "costs in place", 3: "affected rows re-derived, arrays shifted"}[mode], "records_to_host": int(len(rec)),
            "record_bytes": int(rec.nbytes), "spt_identical_to_oracle": ok_spt, "routes_identical_to_fold": ok_routes,
            "records_identical_to_fold": ok_rec}


def _bufs(torch, dev, R, n, W):
    return dict(dist=torch.empty((R, n), dtype=torch.int32, device=dev), hops=torch.empty((R, n), dtype=torch.int16, device=dev),
                flags=torch.empty((R, n), dtype=torch.int16, device=dev), mask=torch.empty((R, n, W), dtype=torch.int64, device=dev))


def _kw(b, W):
    return dict(dist_ptr=b["dist"].data_ptr(), hops_ptr=b["hops"].data_ptr(), flags_ptr=b["flags"].data_ptr(),
                mask_ptr=b["mask"].data_ptr(), mask_words=W)


def _same(b, ref) -> bool:
    return bool(np.array_equal(b["dist"].cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(b["hops"].cpu().numpy().view(np.uint16), ref.hops)
                and np.array_equal(b["flags"].cpu().numpy().view(np.uint16) & 1, ref.flags) and np.array_equal(b["mask"].cpu().numpy().view(np.uint64), ref.mask))


def cold_block(g, dev, roots) -> dict:
    """What a run costs when nothing is warm: the FIRST run of a fresh context on a fresh graph
    handle, the first run with OTHER roots, the first run after a STRUCTURAL one-row patch (a link taken out on both
    sides), next to the steady state of the same loop — one run at a time (hspf_run_device), device ms by HIP events and
    wall ms.  The lean sweep's plan (head sweeps / dense stretch / tail) is sized from the previous run of the context and
    decided on the device, so there is no rehearsal run to lose; `ratio_device` = cold device ms / steady device ms.
    Every run is compared with the oracle."""
    import torch
    from holo_amd import engine as E
    from oracle import graph_oracle as go
    thr = min(64, os.cpu_count() or 1)
    n = g.n
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1, threads=thr)
    b = _bufs(torch, dev, len(roots), n, 1)
    ctx = E.SpfContext(dev.index or 0)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)

    def one(rts):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ctx.run_device(G, rts, 0, **_kw(b, 1))
        return {"wall_ms": round((time.perf_counter() - t0) * 1e3, 4), "device_ms": round(st["ms_total"], 4), "launches": st["n_relax_launches"],
                "dense_passes": st["dbg"][1] & 0xFF}
    first = one(roots); first["identical_to_oracle"] = _same(b, ref)
    second = one(roots)
    for _ in range(6):
        one(roots)
    steady = [one(roots) for _ in range(30)]
    sd, sw = float(np.median([x["device_ms"] for x in steady])), float(np.median([x["wall_ms"] for x in steady]))
    other = ((roots.astype(np.int64) + 777) % n).astype(np.uint32)
    oth = one(other)
    oth["identical_to_oracle"] = _same(b, go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, other, 0, go.HEAP, mask_words_=1, threads=thr))
    one(roots)
    u = n // 3
    a0, b0 = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
    v = int(g.col[a0])
    c0, d0 = int(g.row_ptr[v]), int(g.row_ptr[v + 1])
    keep_u = np.arange(a0, b0)[1:]
    keep_v = np.array([k for k in range(c0, d0) if int(g.col[k]) != u], dtype=np.int64)
    t0 = time.perf_counter()
    G.patch([u, v], [(g.col[keep_u], g.metric[keep_u]), (g.col[keep_v], g.metric[keep_v])], [g.vflags[u], g.vflags[v]])
    tp = (time.perf_counter() - t0) * 1e3
    pat = one(roots)
    pat["identical_to_oracle"] = _same(b, go.run(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1, threads=thr))
    pat["patch_wall_ms"] = round(tp, 4)                   # the first structural patch of the context (scratch allocated), Python twin included
    pat["patch_c_call_ms"] = round(G.last_patch_call_ms, 4)
    G.free(); ctx.close()
    for x in (first, second, oth, pat):
        x["ratio_device"] = round(x["device_ms"] / sd, 3)
    return {"steady_one_at_a_time": {"device_ms": round(sd, 4), "wall_ms": round(sw, 4), "runs_per_s": round(len(roots) / sw * 1e3)},
            "fresh_context_first_run": first, "fresh_context_second_run": second, "other_roots_first_run": oth,
            "after_structural_patch_first_run": pat,
            "note": "first runs include the first touch of the context's scratch (wall) — device ms is the like-for-like figure"}


def consumer_64root(dev) -> dict:
    """What a CONSUMER of a 64-root batch gets, two ways, on the headline graph:
    `pipeline_64root` — nothing but the changed routes leaves the device: cost patch of one router's row ->
    hspf_run_device (64 roots) -> hspf_routes_device (120 000 prefixes x 64 roots) -> hspf_routes_diff_device against the
    previous tables -> hspf_routes_pack; wall ms of the five calls (median of 12 after 3), records checked against a numpy
    fold over the oracle's SPTs for all 64 roots;
    `hspf_run_64root_host` — the SPT tables themselves in HOST memory ("GPU side times hspf_run wall-clock
    including D2H of results"): hspf_run into pinned buffers (102 MB per batch over PCIe), and the same with the copy of
    batch k overlapping the run of batch k + 1 (hspf_run_device_async into device tables + a copy stream)."""
    import ctypes
    import torch
    from holo_amd import synth
    from holo_amd import engine as E
    from holo_amd import _lib as L
    from oracle import graph_oracle as go
    thr = min(64, os.cpu_count() or 1)
    g = synth.isis_100k()
    n, R = g.n, 64
    roots = ((np.arange(R, dtype=np.int64) * n) // R).astype(np.uint32)
    ctx = E.SpfContext(dev.index or 0)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    W = G.mask_words(roots)
    out = {}
    # ---- (a) routes pipeline, 64 roots
    rng = np.random.default_rng(12)
    n2 = 20000
    two = np.sort(rng.integers(0, n, (n2, 2)), axis=1)
    two[:, 1] = np.where(two[:, 1] == two[:, 0], (two[:, 0] + 1) % n, two[:, 1]); two.sort(axis=1)
    vtx = np.concatenate([np.arange(n), two.reshape(-1)]).astype(np.uint32)
    ptr = np.concatenate([np.arange(n), n + 2 * np.arange(n2 + 1)]).astype(np.uint32)
    met = np.concatenate([np.arange(n) % 7, rng.integers(0, 10, 2 * n2)]).astype(np.uint32)
    P = n + n2
    b = _bufs(torch, dev, R, n, W)
    sets = [(torch.empty((R, P), dtype=torch.int32, device=dev), torch.empty((R, P), dtype=torch.int32, device=dev),
             torch.empty((R, P, W), dtype=torch.int64, device=dev)) for _ in range(2)]
    act = torch.empty((R, P), dtype=torch.uint8, device=dev); chg = torch.empty((R * P,), dtype=torch.int32, device=dev)
    cptr = torch.empty((R + 1,), dtype=torch.int32, device=dev)
    u = n // 3
    a0, b0 = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
    costs = [g.metric[a0:b0] + 5, g.metric[a0:b0].copy()]
    col_u, vf_u = g.col[a0:b0].copy(), [g.vflags[u]]

    def routes(k, fl=0):
        ctx.routes_device(n, R, W, b["dist"].data_ptr(), b["flags"].data_ptr(), b["mask"].data_ptr(), ptr, vtx, met,
                          best_metric_ptr=sets[k][0].data_ptr(), best_entry_ptr=sets[k][1].data_ptr(), nexthop_mask_ptr=sets[k][2].data_ptr(), flags=fl)
    ctx.run_device(G, roots, 0, **_kw(b, W)); routes(0)
    cur, stages, rec = 0, [], None
    for it in range(15):
        t = [time.perf_counter()]
        G.patch([u], [(col_u, costs[it & 1])], vf_u); t.append(time.perf_counter())
        ctx.run_device(G, roots, 0, **_kw(b, W)); t.append(time.perf_counter())
        routes(cur ^ 1, E.PFX_RESIDENT); t.append(time.perf_counter())
        ctx.routes_diff_device(R, P, W, tuple(x.data_ptr() for x in sets[cur]), tuple(x.data_ptr() for x in sets[cur ^ 1]),
                               action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
        t.append(time.perf_counter())
        rec = ctx.routes_pack(R, P, W, tuple(x.data_ptr() for x in sets[cur ^ 1]), action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(),
                              changed_ptr_ptr=cptr.data_ptr())
        t.append(time.perf_counter())
        stages.append(np.diff(t) * 1e3)
        cur ^= 1
    # check of the last iteration (it = 14: costs[0]; before it costs[1] = the original costs): every root
    def fold(metric):
        ref = go.run(g.row_ptr, g.col, metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=W, threads=thr)
        bms, nhs = [], []
        for r in range(R):
            dd = ref.dist[r].astype(np.uint64); reach = (ref.flags[r] & 1) != 0
            cand = np.where(reach[vtx], (dd[vtx] + met) & 0xFFFFFFFF, np.uint64(1) << 40)
            bm = np.minimum.reduceat(cand, ptr[:-1].astype(np.int64))
            tie = cand == np.repeat(bm, np.diff(ptr.astype(np.int64)))
            mk = np.where(tie[:, None] & reach[vtx][:, None], ref.mask[r][vtx], 0).astype(np.uint64)
            bms.append(bm); nhs.append(np.bitwise_or.reduceat(mk, ptr[:-1].astype(np.int64), axis=0))
        return ref, np.stack(bms), np.stack(nhs)
    m_new = g.metric.copy(); m_new[a0:b0] = costs[0]
    (_, bm_old, nh_old), (ref, bm_new, nh_new) = fold(g.metric), fold(m_new)
    ok_spt = _same(b, ref)
    has = bm_new < (1 << 40)
    want_r, want_p = np.nonzero(((bm_old != bm_new) | (nh_old != nh_new).any(axis=2)) & has & nh_new.any(axis=2))
    ok_rec = bool(len(rec) == len(want_r) and np.array_equal(rec[:, 0], want_r.astype(np.uint32)) and np.array_equal(rec[:, 1], want_p.astype(np.uint32))
                  and np.array_equal(rec[:, 3], bm_new[want_r, want_p].astype(np.uint32))
                  and np.array_equal(rec[:, 6:].copy().view(np.uint64).reshape(len(rec), W), nh_new[want_r, want_p]))
    st = np.median(np.array(stages[3:]), axis=0)
    wall = float(np.median(np.array(stages[3:]).sum(axis=1)))
    out["pipeline_64root"] = {"graph": "isis-100k", "roots": R, "prefixes": int(P), "changed_row": int(u), "wall_ms": round(wall, 4),
                              "runs_per_s": round(R / wall * 1e3),
                              "stages_ms": {k: round(float(v), 4) for k, v in zip(("graph_patch", "run_device", "routes_device", "routes_diff_device", "routes_pack"), st)},
                              "records_to_host": int(len(rec)), "record_bytes": int(rec.nbytes), "spt_identical_to_oracle": ok_spt,
                              "identical_to_oracle": bool(ok_spt and ok_rec)}
    del sets, act, chg
    # ---- (b) SPT tables to the host
    G.patch([u], [(col_u, costs[1])], vf_u)                           # the original costs again
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=W, threads=thr)
    hb = [dict(dist=torch.empty((R, n), dtype=torch.int32).pin_memory(), hops=torch.empty((R, n), dtype=torch.int16).pin_memory(),
               flags=torch.empty((R, n), dtype=torch.int16).pin_memory(), mask=torch.empty((R, n, W), dtype=torch.int64).pin_memory()) for _ in range(2)]
    res = L.HspfResult(hb[0]["dist"].data_ptr(), hb[0]["hops"].data_ptr(), hb[0]["flags"].data_ptr(), hb[0]["mask"].data_ptr(), W, None)
    rp = roots.ctypes.data_as(L.u32p)
    ts = []
    for it in range(9):
        t0 = time.perf_counter()
        rc = ctx.lib.hspf_run(ctx.handle, G.handle, rp, R, 0, ctypes.byref(res))
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, ctx.last_error()
    st1 = ctx.stats()
    ok_host = bool(np.array_equal(hb[0]["dist"].numpy().view(np.uint32), ref.dist) and np.array_equal(hb[0]["hops"].numpy().view(np.uint16), ref.hops)
                   and np.array_equal(hb[0]["mask"].numpy().view(np.uint64), ref.mask))
    sync_ms = float(np.median(ts[2:]))
    # overlapped: run k + 1 on the lanes while batch k crosses the bus on a copy stream
    db = [_bufs(torch, dev, R, n, W) for _ in range(3)]
    cs = torch.cuda.Stream(device=dev)
    K = 24

    def overlapped(K):
        tickets, copies = [], []
        t0 = time.perf_counter()
        for i in range(K):
            tickets.append((ctx.run_device_async(G, roots, 0, **_kw(db[i % 3], W)), i))
            if len(tickets) >= 2:
                tk, j = tickets.pop(0)
                ctx.wait(tk)
                if len(copies) >= 2:
                    copies.pop(0).synchronize()                   # the host buffer of batch j - 2 is free again
                with torch.cuda.stream(cs):
                    for key in ("dist", "hops", "flags", "mask"):
                        hb[j & 1][key].copy_(db[j % 3][key], non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(cs); copies.append(ev)
        while tickets:
            tk, j = tickets.pop(0)
            ctx.wait(tk)
            if len(copies) >= 2:
                copies.pop(0).synchronize()
            with torch.cuda.stream(cs):
                for key in ("dist", "hops", "flags", "mask"):
                    hb[j & 1][key].copy_(db[j % 3][key], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(cs); copies.append(ev)
        for ev in copies:
            ev.synchronize()
        return (time.perf_counter() - t0) / K * 1e3
    overlapped(6)
    ov_ms = overlapped(K)
    last = hb[(K - 1) & 1]
    ok_ov = bool(np.array_equal(last["dist"].numpy().view(np.uint32), ref.dist) and np.array_equal(last["hops"].numpy().view(np.uint16), ref.hops)
                 and np.array_equal(last["mask"].numpy().view(np.uint64), ref.mask))
    bytes_per_batch = R * n * (8 + 8 * W)
    # ---- (c) the PACKED hand-off (ABI 7): the engine's own word per (root, vertex) — 4 bytes here — instead of 16
    pb = [ctx.host_alloc(8 * R * n) for _ in range(4)]
    tp = []
    for it in range(9):
        t0 = time.perf_counter()
        pr = ctx.run_packed(G, roots, 0, buffer=pb[0])
        tp.append((time.perf_counter() - t0) * 1e3)
    st_p = pr.stats
    ok_packed = bool(np.array_equal(pr.dist, ref.dist) and np.array_equal(pr.hops, ref.hops) and np.array_equal(pr.first_hop_mask, ref.mask[..., :1])
                     and np.array_equal(pr.in_spt, (ref.flags & 1) != 0))
    packed_sync_ms = float(np.median(tp[2:]))
    word_bytes = pr.word_bytes

    def packed_in_flight(K, depth):
        hs = []
        t0 = time.perf_counter()
        last = None
        for i in range(K):
            if len(hs) >= depth:
                last = ctx.wait_packed(hs.pop(0))
            hs.append(ctx.run_packed_async(G, roots, 0, pb[i % len(pb)]))
        while hs:
            last = ctx.wait_packed(hs.pop(0))
        return (time.perf_counter() - t0) / K * 1e3, last
    packed_in_flight(8, 3)
    pk_ov3_ms, last_pr = packed_in_flight(48, 3)
    pk_ov4_ms, last_pr = packed_in_flight(48, 4)
    pk_ov_ms, pk_depth = min((pk_ov3_ms, 3), (pk_ov4_ms, 4))
    ok_packed_ov = bool(np.array_equal(last_pr.dist, ref.dist) and np.array_equal(last_pr.first_hop_mask, ref.mask[..., :1]))
    pageable = np.zeros(8 * R * n + 4096, np.uint8)[4096:]
    tq = []
    for it in range(5):
        t0 = time.perf_counter()
        prq = ctx.run_packed(G, roots, 0, buffer=pageable)
        tq.append((time.perf_counter() - t0) * 1e3)
    ok_pageable = bool(np.array_equal(prq.words, last_pr.words))
    t0 = time.perf_counter()
    _ = (last_pr.dist, last_pr.hops, last_pr.first_hop_mask)
    decode_all_ms = (time.perf_counter() - t0) * 1e3
    packed_bytes = R * n * word_bytes
    out["hspf_run_64root_host"] = {"graph": "isis-100k", "roots": R, "bytes_to_host_per_batch": int(bytes_per_batch), "pinned": True,
                                   "hspf_run_wall_ms": round(sync_ms, 4), "hspf_run_device_ms": round(st1["ms_total"], 4), "hspf_run_d2h_ms": round(st1["ms_d2h"], 4),
                                   "hspf_run_runs_per_s": round(R / sync_ms * 1e3),
                                   "overlapped_wall_ms_per_batch": round(ov_ms, 4), "overlapped_runs_per_s": round(R / ov_ms * 1e3),
                                   "overlapped_pcie_GBps": round(bytes_per_batch / ov_ms / 1e6, 1),
                                   "identical_to_oracle": bool(ok_host and ok_ov),
                                   "packed": {"entry": "hspf_run_packed / hspf_run_packed_async + hspf_wait_packed (ABI 7)", "word_bytes": int(word_bytes),
                                              "bytes_to_host_per_batch": int(packed_bytes),
                                              "sync_wall_ms": round(packed_sync_ms, 4), "sync_runs_per_s": round(R / packed_sync_ms * 1e3),
                                              "sync_device_ms": round(st_p["ms_total"], 4), "sync_d2h_ms": round(st_p["ms_d2h"], 4),
                                              "in_flight_wall_ms_per_batch": round(pk_ov_ms, 4), "in_flight_runs_per_s": round(R / pk_ov_ms * 1e3),
                                              "in_flight_pcie_GBps": round(packed_bytes / pk_ov_ms / 1e6, 1), "tickets_in_flight": pk_depth,
                                              "in_flight_wall_ms_by_depth": {"3": round(pk_ov3_ms, 4), "4": round(pk_ov4_ms, 4)},
                                              "pageable_sync_wall_ms": round(float(np.median(tq[1:])), 4), "pageable_runs_per_s": round(R / float(np.median(tq[1:])) * 1e3),
                                              "numpy_decode_all_tables_ms": round(decode_all_ms, 2),
                                              "identical_to_oracle_after_decode": bool(ok_packed and ok_packed_ov and ok_pageable)},
                                   "note": "PCIe-inclusive: 102 MB of tables per 64-root batch (25.6 MB packed); never the headline `value` (results resident in HBM)"}
    for b_ in pb:
        b_.free()
    G.free(); ctx.close()
    return out


def dropin_e2e() -> dict:
    """What the drop-in costs END TO END, through the compiled C++ host side (include/holo_spf_isis.hpp on HipEngine = the
    C ABI; tests/cpp/dropin_e2e.cpp, built by __graft_entry__.build()): a synthetic isis-100k LSDB at LSP level ->
    LSDB->CSR (first and incremental), run + hand-off (packed, and hspf_run's 16 bytes per vertex for comparison), the
    `Spt` rebuild (holo-isis/src/spf.rs:224-242), compute_routes (:840-949), the persistent device pipeline — next to the
    SAME host code on the CPU stand-in for the engine (oracle heap loop, the cpu_baseline leg) and the stored figure of the
    reference-shaped loop.  The RIB of a down-sized twin of this LSDB is compared with oracle/isis_ref.py in
    tests/test_dropin_e2e.py."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_e2e")
    if not os.path.exists(exe):
        return {"error": "tests/cpp/dropin_e2e is not built (__graft_entry__.build())"}

    def run(*args):
        try:
            p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=240)
            if p.returncode != 0:
                return {"error": f"rc {p.returncode}: {p.stderr.strip()[-300:]}"}
            return json.loads(p.stdout.strip().splitlines()[-1])
        except Exception as ex:  # noqa: BLE001
            return {"error": repr(ex)}
    out = {"host_side": "C++17 twin of the Rust glue (include/holo_spf_isis.hpp), one core", "lsdb": "isis-100k at LSP level: 100 000 LSPs, 1 000 000 TLV-22 adjacencies, 120 000 prefix entries",
           "hip": run("--engine", "hip", "--n", "100000", "--reps", "5", "--batch", "16"),
           "hip_full_tables": run("--engine", "hip", "--n", "100000", "--reps", "3", "--batch", "16", "--no-packed"),
           "cpu_engine_same_host_code": run("--engine", "oracle", "--n", "100000", "--reps", "2", "--batch", "4")}
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "r03_cpu_ref_shaped_100k.json")))
        out["reference_shaped_loop_s_per_run"] = ref.get("seconds")
    except Exception:  # noqa: BLE001
        pass
    return out


def exact_path(ctx, dev) -> dict:
    """Roots with a DYNAMIC pop order: ospf-10k and isis-100k with 0.1 % / 1 % of their link entries at
    metric 0 (holo-isis/src/spf.rs:629-704 treats 0 like any metric), 1 and 64 roots, results in HBM.  Until round 6 every such
    root was re-run by the sequential kernel (one GPU thread per root: seconds per batch at 100 k vertices); now the sweep
    kernels deliver the distances and k_repair (holo_amd/csrc/spf_repair.hip.h) recomputes hops / masks in the true order.
    Beside it: the same graph without zero-cost links, and the CPU heap restatement (1 thread) on the same roots; one root of
    every case is verified against the oracle's literal loop."""
    import torch
    from holo_amd import synth
    from oracle import graph_oracle as go
    out = {}
    for name, g0 in (("ospf-10k", synth.ospf_10k()), ("isis-100k", synth.isis_100k())):
        for share in (0.0, 0.001, 0.01):
            g = g0
            if share:
                rng = np.random.default_rng(11)
                m = g0.metric.copy()
                m[rng.random(len(m)) < share] = 0
                g = synth.CsrGraph(g0.row_ptr, g0.col, m, g0.vflags, g0.max_path_metric, g0.name)
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            for rn in (1, 64):
                roots = (np.arange(rn, dtype=np.uint64) * g.n // rn).astype(np.uint32)
                W = G.mask_words(roots)
                b = _bufs(torch, dev, rn, g.n, W)
                kw = _kw(b, W)
                for _ in range(3):
                    st = ctx.run_device(G, roots, 0, **kw)
                torch.cuda.synchronize()
                reps = 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    st = ctx.run_device(G, roots, 0, **kw)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / reps
                rec = {"ms_per_call": round(ms, 4), "runs_per_s": round(rn / ms * 1e3, 1), "n_exact_roots": st["n_exact_roots"],
                       "n_repaired_roots": st["n_repaired_roots"], "ms_repair": round(st["ms_repair"], 4), "repair_sweeps": st["repair_sweeps"],
                       "repair_evals": st["repair_evals"]}
                if share:
                    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots[:1], 0, go.MAP, mask_words_=W)
                    t0 = time.perf_counter()
                    go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots[:1], 0, go.HEAP, mask_words_=W)
                    rec["cpu_heap_ms_per_run"] = round((time.perf_counter() - t0) * 1e3, 2)
                    rec["first_root_identical_to_the_literal_loop"] = bool(
                        np.array_equal(b["dist"][0].cpu().numpy().view(np.uint32), ref.dist[0]) and np.array_equal(b["hops"][0].cpu().numpy().view(np.uint16), ref.hops[0])
                        and np.array_equal(b["mask"][0].cpu().numpy().view(np.uint64), ref.mask[0]))
                out[f"{name}, {share * 100:g} % zero-cost entries, {rn} root(s)"] = rec
            G.free()
    return out


def path_of(st) -> str:
    """Which engine path a run took, from its hspf_stats."""
    if st.get("single_wg") == 2:
        return "k_xcd (one XCD per root, state replicated in every CU's LDS, one launch)"
    if st.get("single_wg"):
        return "k_single (one workgroup per root)"
    if st.get("lane_vertex"):
        return "k_lv (lane = vertex)"
    dbg = st.get("dbg", [0, 0, 0, 0])
    packed = None
    if st["state_bytes"] == 4:
        packed = "k_fused_lean, 4-byte state" if dbg[0] else "k_fused, 4-byte state"
    if st["state_bytes"] == 8:
        packed = "k_fused, 8-byte state"
    if dbg[1] & 0x80000000:                  # a wide-mask class ran with the graph's leaves left to the emit (low bits: dense passes of the lean sweep)
        return "k_fw (wide masks, leaves derived in the emit)" + (" + " + packed + " for the roots that fit it" if packed else "")
    if packed:
        return packed
    return "k_relax + k_dag (two-phase)" if st["n_dag_launches"] else "k_fw (wide masks)"


def areas_on_two_contexts(dev) -> float:
    """configs[3] as the reference runs it — every area its own SPF instance —, two instances at a time: the ten areas
    dealt to two contexts / host threads (graphs resident, results in HBM), wall time for all 10 000 SPTs, best of 3.
    (One context after the other is `device_ms`; more than two contexts bring nothing on one GPU.)"""
    import threading
    import torch
    from holo_amd import synth
    from holo_amd import engine as E
    areas = synth.ospf_multi_area()
    ctxs = [E.SpfContext(dev.index or 0) for _ in range(2)]
    jobs = [[], []]
    for i, g in enumerate(areas):
        c = ctxs[i & 1]
        roots = np.asarray(g.meta["roots"], np.uint32)
        G = c.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        W, R, n = G.mask_words(roots), len(roots), g.n
        bufs = (torch.empty((R, n), dtype=torch.int32, device=dev), torch.empty((R, n), dtype=torch.int16, device=dev),
                torch.empty((R, n), dtype=torch.int16, device=dev), torch.empty((R, n, W), dtype=torch.int64, device=dev))
        jobs[i & 1].append((G, roots, W, bufs))

    def work(k):
        for G, roots, W, (d, h, f, m) in jobs[k]:
            ctxs[k].run_device(G, roots, E.RUN_NET_NEXTHOPS, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                               mask_ptr=m.data_ptr(), mask_words=W)
    best = 1e9
    for rep in range(4):
        ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        if rep:
            best = min(best, (time.perf_counter() - t0) * 1e3)
    for k in range(2):
        for G, *_ in jobs[k]:
            G.free()
        ctxs[k].close()
    return round(best, 3)


def areas_in_flight(dev) -> float:
    """configs[3] through the asynchronous entry point: the ten areas of one SPF event handed to ONE context
    (hspf_run_device_async, one ticket per area, each area its own graph) and collected in order — what the reference's
    per-area loop (holo-ospf/src/spf.rs:540-542) becomes; wall ms for all 10 000 SPTs, best of 3."""
    import torch
    from holo_amd import synth
    from holo_amd import engine as E
    ctx = E.SpfContext(dev.index or 0)
    jobs = []
    for g in synth.ospf_multi_area():
        roots = np.asarray(g.meta["roots"], np.uint32)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        W = G.mask_words(roots)
        jobs.append((G, roots, W, _bufs(torch, dev, len(roots), g.n, W)))
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tickets = [ctx.run_device_async(G, roots, E.RUN_NET_NEXTHOPS, **_kw(b, W)) for G, roots, W, b in jobs]
        for t in tickets:
            ctx.wait(t)
        if rep:
            best = min(best, (time.perf_counter() - t0) * 1e3)
    for G, *_ in jobs:
        G.free()
    ctx.close()
    return round(best, 3)


def other_configs(ctx, dev) -> dict:
    """The other single-GPU BASELINE configs through the same C ABI entry point (hspf_run_device), results in HBM:
    device time (median of 5 after 2 warm-up runs), runs/s, fraction of the HBM roofline by SURVEY.md 8(d)'s B_alg, the
    engine path, and EVERY (root, vertex) result compared bit for bit with the CPU oracle (outside the timed runs)."""
    import torch
    from holo_amd import synth
    from holo_amd import engine as E
    from oracle import graph_oracle as go
    thr = min(64, os.cpu_count() or 1)

    def one(g, roots, fl):
        roots = np.asarray(roots, np.uint32)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        W = G.mask_words(roots)
        R, n = len(roots), g.n
        d = torch.empty((R, n), dtype=torch.int32, device=dev); h = torch.empty((R, n), dtype=torch.int16, device=dev)
        f = torch.empty((R, n), dtype=torch.int16, device=dev); m = torch.empty((R, n, W), dtype=torch.int64, device=dev)
        ms = []
        # (up to eight roots the graph first tries both kernels — k_xcd twice, the other one three times — and then runs the
        # faster one: the median is taken over runs after that)
        for _ in range(14 if len(roots) <= 8 else 7):
            st = ctx.run_device(G, roots, fl, dist_ptr=d.data_ptr(), hops_ptr=h.data_ptr(), flags_ptr=f.data_ptr(),
                                mask_ptr=m.data_ptr(), mask_words=W)
            ms.append(st["ms_total"])
        ms = ms[-7:]
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, fl, go.HEAP, mask_words_=W, threads=thr)
        ok = bool(np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist) and np.array_equal(h.cpu().numpy().view(np.uint16), ref.hops)
                  and np.array_equal(f.cpu().numpy().view(np.uint16) & 1, ref.flags) and np.array_equal(m.cpu().numpy().view(np.uint64), ref.mask))
        G.free()
        del d, h, f, m, ref
        return float(np.median(ms[2:])), st, W, ok

    out = {}
    g10 = synth.ospf_10k()
    for R in (8, 64, 1024):
        roots = ((np.arange(R, dtype=np.int64) * g10.n) // R).astype(np.uint32)
        ms, st, W, ok = one(g10, roots, E.RUN_NET_NEXTHOPS)
        rps = R / (ms * 1e-3)
        out[f"ospf-10k, {R} roots (configs[1] graph)"] = {
            "device_ms": round(ms, 4), "runs_per_s": round(rps), "roofline_frac": round(rps * b_alg(g10.n, g10.e, W) / HBM_PEAK, 5),
            "alg_bytes_per_run": b_alg(g10.n, g10.e, W), "path": path_of(st), "launches": st["n_relax_launches"] + st["n_dag_launches"],
            "roots_verified": R, "identical_to_oracle": ok}
    # configs[3]: 10 areas x 1000 roots, each area its own graph, one after the other on this context
    tot, nroots, allok, ba, last = 0.0, 0, True, 0, None
    for g in synth.ospf_multi_area():
        ms, st, W, ok = one(g, g.meta["roots"], E.RUN_NET_NEXTHOPS)
        tot += ms; nroots += len(g.meta["roots"]); allok = allok and ok; ba = b_alg(g.n, g.e, W); last = st
    rps = nroots / (tot * 1e-3)
    out["ospf multi-area, 10 areas x 5000 routers x 1000 roots (configs[3], one GPU)"] = {
        "device_ms": round(tot, 3), "runs_per_s": round(rps), "roofline_frac": round(rps * ba / HBM_PEAK, 5), "alg_bytes_per_run": ba,
        "path": path_of(last), "roots_verified": nroots, "identical_to_oracle": allok,
        "wall_ms_areas_on_two_contexts": areas_on_two_contexts(dev), "wall_ms_areas_in_flight_one_context": areas_in_flight(dev)}
    gf = synth.isis_fattree(100)
    ms, st, W, ok = one(gf, gf.meta["roots"], 0)
    rps = len(gf.meta["roots"]) / (ms * 1e-3)
    out["isis fat-tree k=100, 262 500 vertices / 1.5 M entries, 101 roots (configs[4], one GPU)"] = {
        "device_ms": round(ms, 3), "runs_per_s": round(rps), "roofline_frac": round(rps * b_alg(gf.n, gf.e, W) / HBM_PEAK, 5),
        "alg_bytes_per_run": b_alg(gf.n, gf.e, W), "mask_words": W, "path": path_of(st),
        "launches": st["n_relax_launches"] + st["n_dag_launches"], "roots_verified": len(gf.meta["roots"]), "identical_to_oracle": ok}
    return out


def preflight(rank: int, world: int, local_rank: int, dist) -> None:
    """N > 1, before anything is timed: what each rank runs on and how the ranks see each other — device, the RCCL the
    library will load, the peer-access row of this rank — gathered and printed by rank 0 on stderr (the first multi-GPU
    run of this file should explain itself)."""
    import torch
    info = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(local_rank),
            "visible_devices": torch.cuda.device_count(),
            "peer_access": [bool(torch.cuda.can_device_access_peer(local_rank, j)) if j != local_rank else True
                            for j in range(torch.cuda.device_count())],
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "rccl": None}
    try:
        import ctypes.util
        info["rccl"] = ctypes.util.find_library("rccl") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    except Exception:  # noqa: BLE001
        pass
    allp = [None] * world
    dist.all_gather_object(allp, info)
    if rank == 0:
        sys.stderr.write("[bench preflight] " + json.dumps(allp) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gather", choices=["dist", "none"], default="dist",
                    help="N>1: all-gather the per-root distance tables each step (default) or not")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-timed-ms", type=float, default=200.0,
                    help="the timed region is repeated in whole multiples of --steps until it holds at least this much work")
    args = ap.parse_args()

    # the host driver only supports dmabuf IPC (RCCL / cross-process device memory); harmless when already set
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the lanes of the asynchronous runs are streams of their own: with the HIP default of 4 hardware queues two of them
    # share one and run one after the other (137 k vs 149 k runs/s); read when the HIP runtime starts, i.e. before torch.cuda
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand with --gpus N: become the launcher (one rank per GPU over RCCL, rendezvous on 127.0.0.1)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from holo_amd import synth
    # HSPF_BENCH_STUB (tests/test_bench_ranks_gloo.py ONLY): a module that stands in for holo_amd.engine — the oracle behind the
    # MultiEngine interface, tables in host memory — so that THIS file's rank code (launcher, id exchange, sharding offsets,
    # in-flight loop, gather, exchange block, JSON assembly) runs at world size 2 under gloo where there is no GPU.  Never
    # set by the driver; the line says "stub" in `data` when it is.
    stub = os.environ.get("HSPF_BENCH_STUB")
    if stub:
        import importlib
        E = importlib.import_module(stub)
        dev = torch.device("cpu")

        class cuda:                                                   # noqa: N801
            synchronize = staticmethod(lambda *a, **k: None)
        backend = "gloo"
    else:
        from holo_amd import engine as E
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); "
                             "the SPF engine has no CPU path")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        cuda = torch.cuda
        backend = "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        if not stub:
            preflight(rank, world, local_rank, dist)

    g = E.bench_graph() if stub else synth.isis_100k()
    n, e = g.n, g.e
    # The sharded run goes through the C ABI's multi-GPU entry points (hspf_multi_*): one rank = one engine context on
    # this process's GPU; at N > 1 the communicator id of the library's own RCCL all-gather is created on rank 0 and
    # carried to the other processes by torch.distributed (plumbing: holo would use its ibus).
    gather_via = "none"
    m = None
    if world > 1 and args.gather == "dist":
        try:
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.tensor(list(E.multi_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            m = E.MultiEngine([local_rank], world=world, first_rank=rank, unique_id=bytes(idt.cpu().numpy().tobytes()))
            gather_via = "hspf_multi_run: RCCL all-gather inside the C ABI (librccl.so), asynchronous"
        except Exception as ex:  # noqa: BLE001   (plumbing failure: keep the job alive, say so in the JSON line)
            sys.stderr.write(f"[rank {rank}] hspf_multi RCCL communicator failed ({ex}); gathering through torch.distributed\n")
            m = None
        ok = torch.tensor([1 if m is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and m is not None:
            m.close(); m = None
        if m is None:
            gather_via = "torch.distributed all_gather_into_tensor (RCCL) after hspf_multi_run without exchange"
    sharded_in_lib = m is not None
    if m is None:
        m = E.MultiEngine([local_rank])                           # a one-rank job: no exchange inside the library
    mg = m.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    R = 64                                                         # roots per GPU per step (weak scaling)
    all_roots = ((np.arange(R * world, dtype=np.int64) * n) // (R * world)).astype(np.uint32)
    lo, hi = E.shard_bounds(len(all_roots), world, rank)
    roots = all_roots[lo:hi]
    assert hi - lo == R
    W = m.mask_words(mg, all_roots)
    run_roots = all_roots if sharded_in_lib else roots             # the library shards; or this rank's slice, one-rank job
    RA = len(all_roots) if (sharded_in_lib or (world > 1 and args.gather == "dist")) else R
    row0 = lo if RA == len(all_roots) else 0                       # where this rank's rows sit in its tables

    # The timed loop keeps `depth` steps IN FLIGHT (hspf_multi_run_async / hspf_multi_run_wait, ABI 6): a step alone leaves
    # most of the chip idle during the chains of small launches at its head and tail; the lanes of one context interleave
    # those of the steps in flight.  Every step is still ONE 64-root batch of configs[2], complete (results in HBM, exchange
    # issued) when its wait returns; `value` = steps retired per second.  depth = 1 (HSPF_BENCH_DEPTH=1) is one step at a
    # time, reported anyway as `pipeline.one_at_a_time`.
    depth = max(1, int(os.environ.get("HSPF_BENCH_DEPTH", "4")))
    nb = depth + 1
    # device results (row-major [root][vertex]), one set per step in flight + 1; the distance table holds ALL roots when gathered
    bufs = []
    for _ in range(nb + 1):                    # (+ 1: the untimed one-at-a-time pass and the row-count run keep out of the timed steps' tables)
        bufs.append(dict(
            dist=torch.empty((RA, n), dtype=torch.int32, device=dev),
            hops=torch.empty((RA if sharded_in_lib else R, n), dtype=torch.int16, device=dev),
            flags=torch.empty((RA if sharded_in_lib else R, n), dtype=torch.int16, device=dev),
            mask=torch.empty((RA if sharded_in_lib else R, n, W), dtype=torch.int64, device=dev)))

    pending = [None]
    phase = {"relax_ms": 0.0, "dag_ms": 0.0, "finish_ms": 0.0, "total_ms": 0.0, "n_relax": 0, "n_dag": 0,
             "n_exact": 0, "state_bytes": 0, "narrow_overflow": 0, "steps": 0}

    def ptrs(b, own_rows_only: bool):
        off = row0 * n if own_rows_only else 0
        return dict(dist=b["dist"].data_ptr() + off * 4, hops=b["hops"].data_ptr(), flags=b["flags"].data_ptr(),
                    mask=b["mask"].data_ptr(), mask_words=W)

    def record_stats():
        st = m.stats(0)
        phase["relax_ms"] += st["ms_relax"]; phase["dag_ms"] += st["ms_dag"]
        phase["finish_ms"] += st["ms_finish"]; phase["total_ms"] += st["ms_total"]
        phase["n_relax"] += st["n_relax_launches"]; phase["n_dag"] += st["n_dag_launches"]
        phase["n_exact"] += st["n_exact_roots"]
        phase["state_bytes"] = st["state_bytes"]; phase["narrow_overflow"] += st["narrow_overflow"]
        phase["lean"] = int(st.get("dbg", [0])[0]); phase["dense_passes"] = int(st.get("dbg", [0, 0])[1]) & 0xFF
        phase["steps"] += 1

    inflight = []

    def retire(record: bool):
        t, b = inflight.pop(0)
        if sharded_in_lib:
            m.run_wait(t, [ptrs(b, False)], E.GATHER_DIST | E.GATHER_ASYNC)
        else:
            m.run_wait(t, [ptrs(b, True)], 0)
            if world > 1 and args.gather == "dist":
                if pending[0] is not None:
                    pending[0].wait()
                pending[0] = dist.all_gather_into_tensor(b["dist"], b["dist"][row0:row0 + R], async_op=True)
        if record and depth == 1:
            record_stats()

    def step(i: int, record: bool):
        b = bufs[i % nb]
        inflight.append((m.run_async(mg, run_roots, 0, [ptrs(b, not sharded_in_lib)]), b))
        if len(inflight) >= depth:
            retire(record)

    def drain(record: bool = False):
        while inflight:
            retire(record)

    def fence():
        drain()
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None
        m.wait()
        cuda.synchronize()
        if world > 1:
            dist.barrier()
        cuda.synchronize()

    tw = time.perf_counter()
    for i in range(args.warmup):
        step(i, False)
    fence()
    tw = time.perf_counter() - tw
    # The timed region: --steps steps, repeated in whole multiples until it holds >= --min-timed-ms of work (one clock
    # hiccup must not move the headline; 20 steps are 9 ms).  The multiple is fixed BEFORE timing, from the warm-up
    # rate (max over ranks), so every rank times the same number of steps.
    est = tw / max(args.warmup, 1) if args.warmup else 1e-3
    # The warm-up holds the slow runs of a fresh instance (first touch of every buffer): its rate under-counts how many
    # steps 200 ms take.  A few more untimed steps give the rate the timed region will run at (warm-up too: nothing of
    # them is reported).
    tc = time.perf_counter()
    for i in range(12):
        step(i, False)
    fence()
    est = min(est, (time.perf_counter() - tc) / 12)
    if world > 1:
        t = torch.tensor([est], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        est = float(t.item())
    reps = max(1, int(np.ceil(1.3 * args.min_timed_ms * 1e-3 / max(args.steps * est, 1e-9))))   # warm-up steps run long: margin
    timed_steps = args.steps * reps
    t0 = time.perf_counter()
    for i in range(timed_steps):
        step(i, True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # One step at a time on the same context, UNTIMED for the headline: the per-kernel figures of the roofline block come
    # from here (a lane's HIP events also span the launches of the other steps in flight), and so does the rate a caller
    # gets that has only one batch to run.
    one_steps = 60
    for i in range(4):
        m.run(mg, run_roots, 0, [ptrs(bufs[nb], not sharded_in_lib)], 0)
    cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(one_steps):
        m.run(mg, run_roots, 0, [ptrs(bufs[nb], not sharded_in_lib)], 0)
        if depth != 1:
            record_stats()
    cuda.synchronize()
    one_dt = (time.perf_counter() - t1) / one_steps

    last = bufs[(timed_steps - 1) % nb]
    own = slice(row0, row0 + R)
    hrow = own if sharded_in_lib else slice(0, R)
    assert phase["n_exact"] == 0, "headline workload must stay on the wavefront-parallel path"

    # what was timed is what is checked: the last timed step's results of this rank's 64 roots, every (root, vertex),
    # bit for bit against the CPU oracle (outside the timed region; the oracle deals the roots to the host's cores); at
    # N > 1 rank 0 also checks the GATHERED distance table of all N x 64 roots
    verified = verified_gathered = 0
    if rank == 0:
        from oracle import graph_oracle as go
        chk = all_roots if RA == len(all_roots) else roots
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, chk, 0, go.HEAP, mask_words_=W,
                     threads=min(64, os.cpu_count() or 1))
        assert np.array_equal(last["dist"].cpu().numpy().view(np.uint32), ref.dist), "bench: distances differ from the oracle"
        o = slice(row0, row0 + R) if RA == len(all_roots) else slice(0, R)
        assert np.array_equal(last["hops"][hrow].cpu().numpy().view(np.uint16), ref.hops[o]), "bench: hops differ from the oracle"
        assert np.array_equal(last["flags"][hrow].cpu().numpy().view(np.uint16) & 1, ref.flags[o]), "bench: in-SPT flags differ from the oracle"
        assert np.array_equal(last["mask"][hrow].cpu().numpy().view(np.uint64), ref.mask[o]), "bench: first-hop masks differ from the oracle"
        verified = R
        verified_gathered = len(chk) if world > 1 else 0
        del ref
    # one extra, untimed run with the row counter on (the counting kernel instantiation is slower)
    m.run(mg, run_roots, E.RUN_COUNT_ROWS, [ptrs(bufs[nb], not sharded_in_lib)], 0)
    rows_recomputed = int(m.stats(0)["rows_recomputed"])

    # N > 1: the exchange on its own, outside the timed region (inside it the gather is asynchronous and overlaps the next
    # step): one SYNCHRONOUS all-gather of the distance table per rank, so that the first multi-GPU run reports what the
    # exchange costs next to what it moves ((R / G) x N x 4 bytes per rank)
    exchange = None
    if world > 1 and sharded_in_lib:
        ts = []
        for _ in range(3):
            fence()
            t0 = time.perf_counter()
            m.allgather_rows([last["dist"].data_ptr()], n * 4, RA)
            cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        mine = {"rank": rank, "bytes_sent": int(R * n * 4), "bytes_received": int((RA - R) * n * 4), "sync_allgather_ms": round(min(ts), 4)}
        allx = [None] * world
        dist.all_gather_object(allx, mine)
        exchange = {"table": "dist (u32)", "rows_per_rank": R, "row_bytes": n * 4, "per_rank": allx,
                    "note": "synchronous hspf_multi_allgather_rows after the timed region; in the timed region the same exchange is asynchronous"}

    if rank == 0:
        runs = timed_steps * R * world
        value = runs / dt
        ba = b_alg(n, e, W)
        # dominant kernel = the phase with the larger device time; its average launch duration is HIP-event time of the
        # phase / launches (events recorded on the engine's own stream), from the one-step-at-a-time pass.
        K = max(phase["steps"], 1)
        relax_avg = phase["relax_ms"] / max(phase["n_relax"], 1) * 1e-3
        dag_avg = phase["dag_ms"] / max(phase["n_dag"], 1) * 1e-3
        if phase["n_dag"] == 0:
            dominant = "k_fused_lean" if phase.get("lean") else "k_fused"      # fused fast path: one kernel does distances + hops + masks
        else:
            dominant = "k_dag" if phase["dag_ms"] >= phase["relax_ms"] else "k_relax"
        avg = dag_avg if dominant == "k_dag" else relax_avg
        launches_per_step = (phase["n_relax"] + phase["n_dag"]) / K
        bytes_per_launch = R * ba / launches_per_step          # §8d figure x units per launch
        kernel_achieved = bytes_per_launch / avg
        # whole-run figure = what `value` is: algorithmic bytes of the runs retired per second (emit, init and the idle
        # gaps between launches included)
        achieved = value / world * ba
        # HBM-side traffic of ONE step, all kernels (rocprofv3 PMC, separate passes: profiles/traffic.json, made from the
        # binary whose git revision it records): sum over kernels of launches x (2 x FETCH_SIZE + WRITE_SIZE) — the x2 is
        # the guide's gfx950 correction for FETCH_SIZE, calibrated on the emit kernel
        traffic = traffic_step = traffic_rev = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dominant, {}).get("hbm_bytes_per_launch")
                traffic_step = tj.get("per_step", {}).get("hbm_bytes")
                traffic_rev = tj.get("git_rev")
            except Exception:   # noqa: BLE001
                traffic = None
        dev_ms = phase["total_ms"] / K
        out = {
            "metric": "full-SPF runs/sec on 100k-vertex synthetic LSDB",
            "value": round(value, 2), "unit": "spf_runs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / timed_steps * 1e3, 4),
            "timed_steps": timed_steps, "timed_ms": round(dt * 1e3, 2), "verified_roots": verified,
            "verified_gathered_dist_roots": verified_gathered,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic" if not stub else "synthetic (STUB engine: rank-code test, not a measurement)",
            "config": {"workload": "isis-100k: IS-IS L2 100000 routers / 1000000 directed entries, metrics U[1,100], "
                                   "64 concurrent SPF roots per GPU per step (BASELINE.json configs[2])",
                       "n_vertices": n, "n_entries": e, "roots_per_step_per_gpu": R, "mask_words": W,
                       "outputs": "dist u32 + hops u16 + flags u16 + first-hop mask u64 per (root,vertex), in HBM",
                       "parallelism": f"roots sharded over {world} GPU(s) by hspf_multi_run_async / _wait (C ABI), graph replicated",
                       "steps_in_flight": depth,
                       "gather": gather_via},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved / 1e9, 3),
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 5),
                         "frac_is": "whole run: value x B_alg / peak (emit, init and launch gaps included)",
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per LAUNCH of the dominant kernel (PMC)",
                         "alg_bytes_per_run": ba, "alg_bytes_per_step": R * ba,
                         "traffic_per_step": traffic_step, "traffic_ratio": (round(traffic_step / (R * ba), 3) if traffic_step else None),
                         "hbm_gbps_counter": (round(traffic_step / (dev_ms * 1e-3) / 1e9, 1) if traffic_step and dev_ms else None),
                         "hbm_frac_counter": (round(traffic_step / (dev_ms * 1e-3) / HBM_PEAK, 4) if traffic_step and dev_ms else None),
                         "traffic_git_rev": traffic_rev,
                         "kernel_achieved": round(kernel_achieved / 1e9, 3), "kernel_frac": round(kernel_achieved / HBM_PEAK, 5),
                         "kernel_frac_is": "sweep launches only, one step at a time: 64 B_alg / launches per step / average launch duration",
                         "launches_per_step": round(launches_per_step, 2), "avg_launch_us": round(avg * 1e6, 2),
                         "whole_run_frac": round(value / world * ba / HBM_PEAK, 5)},
            "pipeline": {"steps_in_flight": depth, "lanes": E.SpfContext.async_lanes_of(m.ctx_handle(0)),
                         "one_at_a_time": {"runs_per_s": round(R / one_dt), "ms_per_step": round(one_dt * 1e3, 4),
                                           "whole_run_frac": round(R / one_dt * ba / HBM_PEAK, 5), "steps": one_steps}},
            "phases_ms_per_step": {"measured": "one step at a time (the events of a step in flight also span the other steps' launches)",
                                   "relax": round(phase["relax_ms"] / K, 4), "dag": round(phase["dag_ms"] / K, 4),
                                   "finish": round(phase["finish_ms"] / K, 4), "device_total": round(phase["total_ms"] / K, 4),
                                   "relax_launches": phase["n_relax"] / K, "dag_launches": phase["n_dag"] / K,
                                   "dense_passes": phase.get("dense_passes"),
                                   "fused_state_bytes": phase["state_bytes"], "narrow_overflows": phase["narrow_overflow"],
                                   "rows_recomputed_per_step": rows_recomputed, "rows_x_N": round(rows_recomputed / n, 2)},
        }
        if exchange is not None:
            out["exchange"] = exchange
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g, roots)
            ctx1 = E.SpfContext(local_rank)
            # the blocks beside the headline: a failure in one of them is recorded in its place, the line is printed all the same
            def block(key, fn, *a):
                try:
                    r = fn(*a)
                except Exception as ex:  # noqa: BLE001
                    r = {"error": repr(ex)}
                if key is None:
                    out.update(r if "error" not in r else {"consumer_64root": r})
                else:
                    out[key] = r
            block("latency_1root", latency_1root, ctx1, dev)
            block("pipeline_1root", pipeline_1root, ctx1, dev)
            block("cold", cold_block, g, dev, roots)
            block(None, consumer_64root, dev)
            block("two_instances", two_instances, g, dev)
            block("configs", other_configs, ctx1, dev)
            block("dropin_e2e", dropin_e2e)
            block("exact_path", exact_path, ctx1, dev)
        print(json.dumps(out), flush=True)

    m.free_graph(mg)
    m.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
