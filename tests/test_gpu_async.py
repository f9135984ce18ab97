"""Asynchronous runs (hspf_run_device_async / hspf_wait, ABI 6): several runs in flight on the lanes of ONE context give
the results of the synchronous call, bit for bit, and both equal the CPU oracle.  The reference's shape for this is one
SPF per area / level / neighbour of one event (holo-ospf/src/spf.rs:540-542, holo-isis/src/spf.rs:746-761,
holo-isis/src/flooding/manet.rs:59-69)."""
import ctypes
import os

import numpy as np
import pytest

from holo_amd import synth
from holo_amd import engine as E
from oracle import graph_oracle as go

pytestmark = pytest.mark.gpu
ORACLE_THREADS = min(64, os.cpu_count() or 1)


def _tables(torch, dev, R, n, W):
    return dict(dist=torch.zeros((R, n), dtype=torch.int32, device=dev), hops=torch.zeros((R, n), dtype=torch.int16, device=dev),
                flags=torch.zeros((R, n), dtype=torch.int16, device=dev), mask=torch.zeros((R, n, W), dtype=torch.int64, device=dev))


def _kw(t, W):
    return dict(dist_ptr=t["dist"].data_ptr(), hops_ptr=t["hops"].data_ptr(), flags_ptr=t["flags"].data_ptr(),
                mask_ptr=t["mask"].data_ptr(), mask_words=W)


def _check(t, ref):
    assert np.array_equal(t["dist"].cpu().numpy().view(np.uint32), ref.dist)
    assert np.array_equal(t["hops"].cpu().numpy().view(np.uint16), ref.hops)
    assert np.array_equal(t["flags"].cpu().numpy().view(np.uint16) & 1, ref.flags)
    assert np.array_equal(t["mask"].cpu().numpy().view(np.uint64), ref.mask)


def _ctx(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return E.SpfContext(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_runs_in_flight_equal_the_oracle_isis_100k(lanes):
    """BASELINE configs[2] shape: 64-root batches of isis-100k, `lanes` of them in flight, different roots per ticket."""
    import torch
    dev = torch.device("cuda:0")
    g = synth.isis_100k()
    n = g.n
    ctx = _ctx(HSPF_ASYNC_LANES=lanes)
    try:
        assert ctx.async_lanes() == lanes
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        sets = [((np.arange(64, dtype=np.int64) * n // 64 + 131 * k) % n).astype(np.uint32) for k in range(5)]
        refs = [go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, r, 0, go.HEAP, mask_words_=1, threads=ORACLE_THREADS) for r in sets]
        tabs = [_tables(torch, dev, 64, n, 1) for _ in sets]
        for rounds in range(2):                                   # the second round runs on plans sized by the first
            tickets = [ctx.run_device_async(G, r, 0, **_kw(t, 1)) for r, t in zip(sets, tabs)]
            assert tickets == sorted(tickets) and len(set(tickets)) == len(tickets)
            stats = [ctx.wait(t) for t in tickets]
            for st, t, ref in zip(stats, tabs, refs):
                assert st["n_roots"] == 64
                _check(t, ref)
                t["dist"].zero_(); t["mask"].zero_()
        # the synchronous call on the same context, between asynchronous ones
        t1 = ctx.run_device_async(G, sets[0], 0, **_kw(tabs[0], 1))
        ctx.run_device(G, sets[1], 0, **_kw(tabs[1], 1))
        ctx.wait(t1)
        _check(tabs[0], refs[0]); _check(tabs[1], refs[1])
        G.free()
    finally:
        ctx.close()


def test_patch_waits_for_runs_in_flight_and_errors_are_codes():
    """hspf_graph_patch on a context with runs in flight waits for them (they read the arrays it rewrites); a bad ticket
    is HSPF_E_INVAL, a bad root comes back through hspf_wait as the run's own code."""
    import torch
    dev = torch.device("cuda:0")
    g = synth.ospf_10k()
    n = g.n
    ctx = _ctx(HSPF_ASYNC_LANES=3)
    try:
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        roots = (np.arange(128, dtype=np.int64) * n // 128).astype(np.uint32)
        W = G.mask_words(roots)
        tabs = [_tables(torch, dev, 128, n, W) for _ in range(3)]
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=W, threads=ORACLE_THREADS)
        tickets = [ctx.run_device_async(G, roots, 0, **_kw(t, W)) for t in tabs]
        u = n // 2
        a0, b0 = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
        G.patch([u], [(g.col[a0:b0], g.metric[a0:b0] + 5)], [g.vflags[u]])        # returns only after the three runs
        for t in tabs:
            _check(t, ref)
        for tk in tickets:
            ctx.wait(tk)
        g2 = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
        ref2 = go.run(g2.row_ptr, g2.col, g2.metric, g2.vflags, g2.max_path_metric, roots, 0, go.HEAP, mask_words_=W, threads=ORACLE_THREADS)
        tk = ctx.run_device_async(G, roots, 0, **_kw(tabs[0], W))
        ctx.wait(tk)
        _check(tabs[0], ref2)
        with pytest.raises(E.HspfError):
            ctx.wait(10 ** 9)                                                      # never handed out
        bad = roots.copy(); bad[3] = n + 7
        tk = ctx.run_device_async(G, bad, 0, **_kw(tabs[1], W))
        with pytest.raises(E.HspfError) as ei:
            ctx.wait(tk)
        assert "root out of range" in str(ei.value)
        G.free()
    finally:
        ctx.close()


def test_random_lsdbs_async_equal_sync():
    """Adversarial LSDBs (LANs, one-way and zero-cost links, overload bits): every engine path behind the asynchronous
    entry point, three runs in flight, against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    ctx = _ctx(HSPF_ASYNC_LANES=3)
    try:
        for seed in range(12):
            g = synth.random_lsdb(300 + 40 * seed, 10 + seed, 3.0, 100 + seed, metric_hi=7, zero_cost_router_links=(seed % 4 == 3))
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            rsets = [np.arange(k, k + 70, dtype=np.uint32) % g.n for k in (0, 50, 111)]
            W = max(G.mask_words(r) for r in rsets)
            tabs = [_tables(torch, dev, 70, g.n, W) for _ in rsets]
            flags = E.RUN_NET_NEXTHOPS if seed % 2 else 0
            tickets = [ctx.run_device_async(G, r, flags, **_kw(t, W)) for r, t in zip(rsets, tabs)]
            for tk, r, t in zip(tickets, rsets, tabs):
                ctx.wait(tk)
                ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, r, flags, go.MAP, mask_words_=W)
                _check(t, ref)
            G.free()
    finally:
        ctx.close()


def test_classes_of_one_run_side_by_side():
    """A synchronous run whose roots fall into two state classes (fat-tree k=64: the edge switch and its 32 aggregation
    switches have 64 first-hop slots -> k_fw, its 32 hosts have one -> the packed path) sends one class to a lane and runs
    the other itself (run_classes): same tables as one class after the other (HSPF_VARIANT bit 20) and as the oracle —
    also with an asynchronous ticket of another run queued on the lanes in front of it, and through the host-output call."""
    import torch
    dev = torch.device("cuda:0")
    g = synth.isis_fattree(64)
    roots = np.asarray(g.meta["roots"], np.uint32)
    assert len(roots) == 65 and g.n * len(roots) >= 1 << 22
    g2 = synth.ospf_10k()
    roots2 = np.arange(64, dtype=np.uint32) * 150
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1, threads=ORACLE_THREADS)
    ref2 = go.run(g2.row_ptr, g2.col, g2.metric, g2.vflags, g2.max_path_metric, roots2, 0, go.HEAP, mask_words_=1, threads=ORACLE_THREADS)
    launches = {}
    # (HSPF_VARIANT bit 23: the host roots run as a class of their own here, not derived from their switch's rows — that path
    # has its own test, tests/test_gpu_parity.py::test_leaf_roots_derived_from_their_neighbour)
    for name, env in (("side by side", {"HSPF_VARIANT": 8388608}), ("one after the other", {"HSPF_VARIANT": 8388608 | 1048576})):
        ctx = _ctx(**env)
        try:
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            G2 = ctx.upload(g2.row_ptr, g2.col, g2.metric, g2.vflags, g2.max_path_metric)
            W = G.mask_words(roots)
            assert W == 1
            t, t2 = _tables(torch, dev, len(roots), g.n, W), _tables(torch, dev, 64, g2.n, 1)
            for rep in range(3):
                for x in t.values():
                    x.zero_()
                tk = ctx.run_device_async(G2, roots2, 0, **_kw(t2, 1)) if rep == 1 else None
                st = ctx.run_device(G, roots, 0, **_kw(t, W))
                if tk is not None:
                    ctx.wait(tk)
                    _check(t2, ref2)
                _check(t, ref)
                assert st["n_roots"] == len(roots)
            launches[name] = st["n_relax_launches"]
            res = ctx.run(G, roots, 0)                                  # host outputs: the classes write into the staging, one copy out
            assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops)
            assert np.array_equal(res.flags & 1, ref.flags) and np.array_equal(res.first_hop_mask, ref.mask)
            G.free(); G2.free()
        finally:
            ctx.close()
    assert launches["side by side"] == launches["one after the other"]


def test_thread_model_shutdown_with_tickets_in_flight_and_expired_tickets():
    """The lanes are host threads owned by the context (INTEGRATION.md 5f "thread model"): hspf_shutdown with runs still in flight
    drains them and joins the threads (no crash, no hang, the tables are written); a ticket whose result has been pushed out
    of its lane's ring of eight is an error CODE (HSPF_E_INVAL with a text), never stale data."""
    import torch
    dev = torch.device("cuda:0")
    g = synth.ospf_10k()
    n = g.n
    roots = (np.arange(64, dtype=np.int64) * n // 64).astype(np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1, threads=ORACLE_THREADS)
    ctx = _ctx(HSPF_ASYNC_LANES=2)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    tabs = [_tables(torch, dev, 64, n, 1) for _ in range(4)]
    for t in tabs:
        ctx.run_device_async(G, roots, 0, **_kw(t, 1))
    ctx.close()                                       # four runs in flight: shutdown waits for them, then frees everything
    torch.cuda.synchronize()
    for t in tabs:
        _check(t, ref)
    ctx = _ctx(HSPF_ASYNC_LANES=1)
    try:
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        tickets = [ctx.run_device_async(G, roots, 0, **_kw(tabs[i % 4], 1)) for i in range(11)]
        ctx.wait(tickets[-1])
        with pytest.raises(E.HspfError) as ei:
            ctx.wait(tickets[0])                      # ten later runs on its lane: the result slot has been reused
        assert ei.value.code == -1 and "result is gone" in str(ei.value)
        ctx.wait(tickets[-2])                         # still in the ring
        G.free()
    finally:
        ctx.close()
