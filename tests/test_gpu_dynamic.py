"""Roots whose pop order is dynamic (zero-cost router links) on the GPU: distances from the sweep kernels, hops and first-hop
masks recomputed in the true pop order by k_repair (holo_amd/csrc/spf_repair.hip.h) — bit for bit against the oracle's literal
loop (holo-isis/src/spf.rs:629-704, holo-ospf/src/spf.rs:666-719: metric 0 is a metric like any other), in every engine
configuration, WITHOUT the sequential kernel (hspf_stats.n_exact_roots == 0)."""
import os

import numpy as np
import pytest

from holo_amd import synth
from holo_amd import engine as E
from oracle import graph_oracle as go

from _engines import all_engines, both_engines

pytestmark = pytest.mark.gpu
THREADS = min(64, os.cpu_count() or 1)


def _with_zero_links(g, share, seed):
    """`share` of the links (both directions of a link independently) set to cost 0."""
    rng = np.random.default_rng(seed)
    m = g.metric.copy()
    m[rng.random(len(m)) < share] = 0
    return synth.CsrGraph(g.row_ptr, g.col, m, g.vflags, g.max_path_metric, g.name + f"-zero{share}")


def check_dynamic(ctx, g, roots, run_flags=0, threads=1, want_repaired=True):
    roots = np.asarray(roots, np.uint32)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        res = ctx.run(G, roots, run_flags)
    finally:
        G.free()
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, run_flags & 3, go.MAP,
                 mask_words_=res.first_hop_mask.shape[2], threads=threads)
    assert np.array_equal(res.dist, ref.dist), "dist"
    assert np.array_equal(res.flags & 1, ref.flags), "in-SPT flag"
    bad = [(int(r), int(np.nonzero(res.hops[i] != ref.hops[i])[0][0])) for i, r in enumerate(roots) if not np.array_equal(res.hops[i], ref.hops[i])]
    assert not bad, ("hops", bad[:4], res.stats)
    bad = [(int(r), int(np.nonzero((res.first_hop_mask[i] != ref.mask[i]).any(axis=1))[0][0])) for i, r in enumerate(roots)
           if not np.array_equal(res.first_hop_mask[i], ref.mask[i])]
    assert not bad, ("first-hop mask", bad[:4], res.stats)
    if res.pop_rank is not None:
        bad = [int(r) for i, r in enumerate(roots) if not np.array_equal(res.pop_rank[i], ref.pop_rank[i])]
        assert not bad, ("pop rank", bad[:4], res.stats)
    st = res.stats
    assert st["n_exact_roots"] == 0, st
    if want_repaired:
        assert st["n_repaired_roots"] > 0, st
    # RF_EXACT = "this root's pop order is not the static one": the rows k_repair went over (and the rows of leaf roots that
    # were derived from such a neighbour's rows, k_leaf_root_rows)
    marked = ((res.flags & E.RF_EXACT) != 0).any(axis=1)
    assert int(marked.sum()) >= st["n_repaired_roots"] and (st["n_repaired_roots"] > 0) == bool(marked.any()), st
    for i in np.nonzero(marked)[0]:
        assert not (((res.flags[i] & E.RF_EXACT) != 0) & ((res.flags[i] & 1) == 0)).any()       # only vertices of the SPT carry it
    return res, ref


@all_engines
@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("run_flags", [0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD])
def test_zero_cost_router_links_without_the_sequential_kernel(spf_ctx, seed, run_flags):
    hi = 1 + seed % 4                     # half ... a fifth of the router links cost 0: nested groups, zero-cost cycles
    g = synth.random_lsdb(70 + 5 * (seed % 3), 6 if seed % 2 else 0, 2.6 + 0.2 * (seed % 3), 7000 + seed, metric_hi=hi, zero_cost_router_links=True)
    roots = np.arange(g.n, dtype=np.uint32)[: 40 + 9 * (seed % 4)]
    if spf_ctx.mode == "xcd":
        roots = roots[:8]
    check_dynamic(spf_ctx, g, roots, run_flags)


@all_engines
@pytest.mark.parametrize("seed", range(6))
def test_pop_rank_of_a_dynamic_order_without_the_sequential_kernel(spf_ctx, seed):
    """HSPF_RUN_POP_RANK: the position of every vertex in the reference's pop order — the order of the keys (dist, R, pos) —
    equals the oracle's literal loop, nested groups and zero-cost cycles included."""
    g = synth.random_lsdb(80, 6 if seed % 2 else 0, 2.8, 7400 + seed, metric_hi=1 + seed % 3, zero_cost_router_links=True)
    roots = np.arange(g.n, dtype=np.uint32)[:50]
    if spf_ctx.mode == "xcd":
        roots = roots[:8]
    check_dynamic(spf_ctx, g, roots, E.RUN_POP_RANK | (E.RUN_NET_NEXTHOPS if seed % 2 else 0))


def test_pop_rank_isis_100k_with_zero_cost_links(spf_ctx):
    g = _with_zero_links(synth.isis_100k(), 0.01, 13)
    roots = (np.arange(8, dtype=np.uint64) * g.n // 8).astype(np.uint32)
    check_dynamic(spf_ctx, g, roots, E.RUN_POP_RANK, threads=8)


@both_engines
def test_more_roots_than_a_batch_and_padding(spf_ctx):
    g = synth.random_lsdb(150, 10, 3.0, 7100, metric_hi=2, zero_cost_router_links=True)
    roots = np.arange(g.n, dtype=np.uint32)
    roots[7] = E.NO_ROOT
    roots[100] = E.NO_ROOT
    check_dynamic(spf_ctx, g, roots)


@both_engines
def test_the_sequential_kernel_gives_the_same_rows(spf_ctx, monkeypatch):
    """HSPF_VARIANT bit 27 (the path before round 6: every flagged root through k_exact) on a context of its own: identical
    tables, flags included."""
    g = synth.random_lsdb(90, 6, 3.0, 7200, metric_hi=2, zero_cost_router_links=True)
    roots = np.arange(10, 60, dtype=np.uint32)
    if spf_ctx.mode == "xcd":
        roots = roots[:8]
    res, _ = check_dynamic(spf_ctx, g, roots)
    monkeypatch.setenv("HSPF_VARIANT", str(1 << 27))
    old = E.SpfContext(0)
    try:
        G = old.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        seq = old.run(G, roots, 0)
        G.free()
    finally:
        old.close()
    assert seq.stats["n_exact_roots"] == res.stats["n_repaired_roots"] and seq.stats["n_repaired_roots"] == 0
    for f in ("dist", "hops", "flags", "first_hop_mask"):
        assert np.array_equal(getattr(seq, f), getattr(res, f)), f


@both_engines
def test_distances_only_need_no_repair(spf_ctx):
    """A caller that asks for distances alone (LFA-style consumers): they are final as the sweep kernels leave them."""
    import torch
    g = synth.random_lsdb(120, 0, 3.0, 7300, metric_hi=2, zero_cost_router_links=True)
    roots = np.arange(0, 64, dtype=np.uint32)
    if spf_ctx.mode == "xcd":
        roots = roots[:8]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        d = torch.empty((len(roots), g.n), dtype=torch.int32, device="cuda:0")
        st = spf_ctx.run_device(G, roots, 0, dist_ptr=d.data_ptr())
    finally:
        G.free()
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.MAP)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), ref.dist)
    assert st["n_exact_roots"] == 0 and st["n_repaired_roots"] == 0


@pytest.mark.parametrize("share", [0.001, 0.01, 0.05])
def test_isis_100k_with_zero_cost_links_full_size(spf_ctx, share):
    """BASELINE configs[2] with 0.1 % / 1 % / 5 % of its 1 M link entries at metric 0, all 64 roots verified."""
    g = _with_zero_links(synth.isis_100k(), share, 11)
    roots = (np.arange(64, dtype=np.uint64) * g.n // 64).astype(np.uint32)
    res, _ = check_dynamic(spf_ctx, g, roots, threads=THREADS)
    assert res.stats["n_repaired_roots"] == 64


def test_ospf_10k_with_zero_cost_links_one_and_eight_roots(spf_ctx):
    g = _with_zero_links(synth.ospf_10k(), 0.01, 12)
    for roots in ([0], [0, 1234, 5000, 9999, 17, 4242, 7777, 2]):
        check_dynamic(spf_ctx, g, roots, E.RUN_NET_NEXTHOPS, threads=8, want_repaired=False)
