"""k_xcd — one to eight roots on a mid-size graph, one XCD per root, the state replicated in every CU's LDS, ONE launch with a
barrier inside the XCD per sweep (holo_amd/csrc/spf_kernels.hip.h) — against the CPU oracle, bit for bit; the choice the
product context makes between it and the launch-per-sweep engine; and the safety net: a run whose barrier gives up is redone
on the launch-per-sweep path and the context stops trying.  (The "xcd" configuration of tests/_engines.py sends every
small adversarial graph of the suite through the kernel as well.)"""
import os

import numpy as np
import pytest

from holo_amd import synth
from holo_amd import engine as E
from oracle import graph_oracle as go

pytestmark = pytest.mark.gpu


def _ctx(**env):
    env = {k: str(v) for k, v in env.items()}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return E.SpfContext(0)                     # the switches are read once, at hspf_init
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _same(res, ref):
    return (np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops) and np.array_equal(res.flags & 1, ref.flags)
            and np.array_equal(res.first_hop_mask, ref.mask))


def _graphs():
    yield synth.ospf_10k(), 1
    yield synth.random_lsdb(5000, 300, 3.0, 77, metric_hi=60, lan_size=6), 1
    yield synth.random_lsdb(18000, 800, 3.2, 78, metric_hi=60, lan_size=8, p_overload=0.02, p_oneway=0.03), 0
    yield synth.random_lsdb(6000, 200, 3.0, 79, hopcount=True, lan_size=5), 2
    yield synth.random_lsdb(3000, 40, 2.5, 80, metric_hi=3, lan_size=30, zero_cost_router_links=True), 3      # roots for the sequential kernel


@pytest.fixture(scope="module")
def xcd_ctx():
    ctx = _ctx(HSPF_XCD_ALWAYS=1)
    yield ctx
    ctx.close()


def test_one_to_eight_roots_on_mid_size_graphs(xcd_ctx):
    rng = np.random.default_rng(5)
    for g, fl in _graphs():
        G = xcd_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for k in (1, 2, 3, 5, 8):
            roots = rng.choice(g.n, size=k, replace=False).astype(np.uint32)
            if k == 5:
                roots[2] = E.NO_ROOT                                   # a padding entry: empty SPT
            if k == 3:
                roots[0] = int(np.flatnonzero(g.vflags & 1)[0]) if (g.vflags & 1).any() else roots[0]   # a network vertex as root
            res = xcd_ctx.run(G, roots, fl)
            ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, fl & 3, go.MAP, mask_words_=res.first_hop_mask.shape[2])
            assert _same(res, ref), (g.name, k, res.stats)
            if res.first_hop_mask.shape[2] == 1 and res.stats["state_bytes"]:
                assert res.stats["single_wg"] == 2 and (res.stats["dbg"][1] & 0xFFFF) > 0, res.stats      # k_xcd, and its sweeps
                assert not res.stats["dbg"][1] >> 31, "the workgroups of a root did not share an XCD"
                pr = xcd_ctx.run_packed(G, roots, fl)                  # the same run through the packed hand-off
                assert np.array_equal(pr.dist, ref.dist) and np.array_equal(pr.hops, ref.hops) and np.array_equal(pr.in_spt, ref.flags.astype(bool))
                assert np.array_equal(pr.first_hop_mask[..., 0], ref.mask[..., 0])
        G.free()


def test_patches_between_runs(xcd_ctx):
    g = synth.random_lsdb(4000, 100, 3.0, 91, metric_hi=20, lan_size=6)
    G = xcd_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    rng = np.random.default_rng(6)
    roots = np.asarray([150, 2000], np.uint32)
    for step in range(6):
        vs = np.sort(rng.choice(np.arange(100, g.n), size=3, replace=False))
        rows, fl = [], []
        for v in vs.tolist():
            c = G.col[G.row_ptr[v]:G.row_ptr[v + 1]]; m = G.metric[G.row_ptr[v]:G.row_ptr[v + 1]].copy()
            if step % 2 and len(c) > 1:
                c, m = c[1:], m[1:]                                    # a link goes away: structural
            elif len(m):
                m[:] = rng.integers(1, 30, size=len(m))                # costs only: in place
            rows.append((c, m)); fl.append(int(G.vflags[v]))
        G.patch(vs, rows, fl)
        res = xcd_ctx.run(G, roots, 1)
        ref = go.run(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, roots, 1, go.MAP, mask_words_=res.first_hop_mask.shape[2])
        assert _same(res, ref), (step, res.stats)
    G.free()


def test_the_product_context_chooses_by_the_shape_of_the_run_alone():
    """Round 6: no per-graph history — one to eight roots on a graph of at most 20 000 vertices take
    k_xcd, run after run; the reference's own case (a lean 500-router area, one root) stays on the one-workgroup kernel; more
    than eight roots take the batched sweeps."""
    ctx = E.SpfContext(0)
    for g, roots, want in ((synth.ospf_10k(), [0], 2), (synth.ospf_10k(), [0, 5, 9000], 2), (synth.ospf_500(), [0], 1), (synth.ospf_500(), [0, 7], 2),
                           (synth.ospf_10k(), list(range(9)), 0)):
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        roots = np.asarray(roots, np.uint32)
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 1, go.MAP, mask_words_=1)
        for it in range(4):
            res = ctx.run(G, roots, 1)
            assert _same(res, ref), (it, res.stats)
            assert res.stats["single_wg"] == want, (g.name, len(roots), it, res.stats)
        G.free()
    ctx.close()


def test_a_barrier_that_gives_up_sends_the_run_to_the_sweep_engine():
    ctx = _ctx(HSPF_XCD_ALWAYS=1, HSPF_XCD_TIMEOUT_MS=0)                  # every wait of the kernel gives up at once
    g = synth.random_lsdb(5000, 300, 3.0, 77, metric_hi=60, lan_size=6)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.asarray([400, 900, 4000], np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 1, go.MAP, mask_words_=1)
    for it in range(3):
        res = ctx.run(G, roots, 1)
        assert _same(res, ref), (it, res.stats)
        assert res.stats["single_wg"] != 2, res.stats                     # the launch-per-sweep path delivered (and from run 2 on k_xcd is not tried)
    pr = ctx.run_packed(G, roots, 1)
    assert np.array_equal(pr.dist, ref.dist) and np.array_equal(pr.first_hop_mask[..., 0], ref.mask[..., 0])
    G.free()
    ctx.close()


def test_workgroups_on_different_xcds_send_the_run_to_the_sweep_engine():
    """the barrier's plain stores and sc1 loads are coherent inside ONE XCD's L2 only.  HSPF_XCD_SKEW (tests)
    makes a root's workgroups consecutive blocks, i.e. spreads them over the eight XCDs: whether that launch times out on a
    barrier or runs through, its workgroups report different XCC ids and the host must NOT accept what it wrote — the run is
    redone by the launch-per-sweep engine, results equal the oracle's, and the context stops trying the kernel."""
    ctx = _ctx(HSPF_XCD_ALWAYS=1, HSPF_XCD_SKEW=1)
    g = synth.random_lsdb(5000, 300, 3.0, 77, metric_hi=60, lan_size=6)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.asarray([400, 900, 4000], np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 1, go.MAP, mask_words_=1)
    for it in range(3):
        res = ctx.run(G, roots, 1)
        assert _same(res, ref), (it, res.stats)
        assert res.stats["single_wg"] != 2, res.stats                     # never the skewed kernel's own output
    G.free()
    ctx.close()


def test_two_instances_at_once_stay_correct():
    """Two contexts on two host threads (two protocol instances), each sending one-root runs through k_xcd back to back: the
    kernels of the two may land on the same XCD (a workgroup per CU each: they queue behind each other), every result is
    right, and a run that gave up on a barrier — none is expected — would have been redone by the sweep engine."""
    import threading
    g = synth.ospf_10k()
    ref = {r: go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, np.asarray([r], np.uint32), 1, go.MAP, mask_words_=1) for r in (0, 5000)}
    out = {}

    def instance(i, root):
        ctx = _ctx(HSPF_XCD_ALWAYS=1)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        good = taken = 0
        for it in range(150):
            res = ctx.run(G, np.asarray([root], np.uint32), 1)
            good += _same(res, ref[root])
            taken += res.stats["single_wg"] == 2
        G.free(); ctx.close()
        out[i] = (good, taken)

    ts = [threading.Thread(target=instance, args=(i, r)) for i, r in enumerate((0, 5000))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out[0][0] == 150 and out[1][0] == 150, out
    print("k_xcd runs of the two instances:", out)
    assert out[0][1] >= 140 and out[1][1] >= 140, out                 # (a context that gave up once stops using the kernel: far fewer)


def test_tickets_in_flight_through_k_xcd():
    """Several one- to eight-root runs of one instance in flight (hspf_run_device_async: the lanes are private contexts on
    their own streams and host threads): every lane's k_xcd launch takes its own XCDs' worth of workgroups next to the
    others', every table is right; then the same tickets again (the graph has tried both kernels by then and chosen)."""
    import torch
    dev = torch.device("cuda:0")
    g = synth.ospf_10k()
    ctx = _ctx(HSPF_ASYNC_LANES=3)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    sets = [np.asarray([17 * k + 3], np.uint32) if k % 2 else (np.arange(1 + k, dtype=np.uint32) * 997 + k) for k in range(6)]
    refs = [go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, r, 1, go.MAP, mask_words_=1) for r in sets]
    tabs = [dict(dist=torch.zeros((len(r), g.n), dtype=torch.int32, device=dev), hops=torch.zeros((len(r), g.n), dtype=torch.int16, device=dev),
                 flags=torch.zeros((len(r), g.n), dtype=torch.int16, device=dev), mask=torch.zeros((len(r), g.n, 1), dtype=torch.int64, device=dev)) for r in sets]
    for rnd in range(4):
        for t in tabs:
            for x in t.values():
                x.zero_()
        tickets = [ctx.run_device_async(G, r, 1, dist_ptr=t["dist"].data_ptr(), hops_ptr=t["hops"].data_ptr(), flags_ptr=t["flags"].data_ptr(),
                                        mask_ptr=t["mask"].data_ptr(), mask_words=1) for r, t in zip(sets, tabs)]
        for tk in tickets:
            ctx.wait(tk)
        for t, ref in zip(tabs, refs):
            assert np.array_equal(t["dist"].cpu().numpy().view(np.uint32), ref.dist), rnd
            assert np.array_equal(t["hops"].cpu().numpy().view(np.uint16), ref.hops), rnd
            assert np.array_equal(t["flags"].cpu().numpy().view(np.uint16) & 1, ref.flags), rnd
            assert np.array_equal(t["mask"].cpu().numpy().view(np.uint64), ref.mask), rnd
    G.free()
    ctx.close()


@pytest.mark.parametrize("n_routers,expect_xcd", [(19999 - 300, True), (20000 - 300, True), (20001 - 300, False)])
def test_the_largest_graph_the_kernel_takes(xcd_ctx, n_routers, expect_xcd):
    """20 000 vertices x 8 bytes = the 160 000 bytes of LDS a workgroup's replica may take: the graph at the limit, one below
    and one above (which the launch-per-sweep engine runs), one and eight roots, against the oracle."""
    g = synth.random_lsdb(n_routers, 300, 2.6, 1234 + n_routers, metric_hi=30, lan_size=5)
    assert g.n == n_routers + 300
    G = xcd_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    for roots in (np.asarray([g.n - 1], np.uint32), np.asarray([300, 5000, 9999, 12345, 15000, 17000, 19000, g.n - 2], np.uint32)):
        res = xcd_ctx.run(G, roots, 1)
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 1, go.MAP, mask_words_=res.first_hop_mask.shape[2])
        assert _same(res, ref), (g.n, len(roots), res.stats)
        assert (res.stats["single_wg"] == 2) == expect_xcd, res.stats
    G.free()
