"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle, bit for bit.

Integer path: every comparison is exact (dist u32, hops u16, in-SPT flag, first-hop mask).
"""
import numpy as np
import pytest

from holo_amd import synth
from holo_amd import engine as E
from oracle import graph_oracle as go

from _engines import all_engines, both_engines, sweeps_engine  # noqa: E402

pytestmark = pytest.mark.gpu

import os
ORACLE_THREADS = min(64, os.cpu_count() or 1)      # the oracle deals whole roots to host threads (exhaustive full-size checks)


def check(ctx, g, roots, run_flags=0, oracle_variant=go.MAP, expect_exact=None):
    roots = np.asarray(roots, np.uint32)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        res = ctx.run(G, roots, run_flags)
    finally:
        G.free()
    oflags = run_flags & (E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, oflags, oracle_variant,
                 mask_words_=res.first_hop_mask.shape[2])
    assert np.array_equal(res.dist, ref.dist), "dist"
    assert np.array_equal(res.flags & 1, ref.flags), "in-SPT flag"
    assert np.array_equal(res.hops, ref.hops), "hops"
    assert np.array_equal(res.first_hop_mask, ref.mask), "first-hop mask"
    if res.pop_rank is not None:
        assert np.array_equal(res.pop_rank, ref.pop_rank), "pop rank"
    if expect_exact is not None:
        assert (res.stats["n_exact_roots"] > 0) == expect_exact
    return res, ref


@both_engines
@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("run_flags", [0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD])
def test_random_lsdb_normal_metrics(spf_ctx, seed, run_flags):
    """Routers + LANs, parallel / one-way links, overload and no-expand vertices, tie-heavy
    metrics >= 1: the static-order fast path must cover these (no exact-kernel roots)."""
    g = synth.random_lsdb(60, 8, 3.0, seed, metric_hi=6)
    roots = np.arange(8, 8 + 40, dtype=np.uint32)
    check(spf_ctx, g, roots, run_flags, expect_exact=False)


@both_engines
@pytest.mark.parametrize("seed", range(6))
def test_random_lsdb_all_roots_ragged(spf_ctx, seed):
    """Ragged root counts (not multiples of 64), incl. network vertices as roots and padding."""
    g = synth.random_lsdb(90, 10, 2.5, 100 + seed, metric_hi=4)
    roots = np.arange(g.n, dtype=np.uint32)             # 100 roots -> 2 batches, second ragged
    roots[5] = E.NO_ROOT
    check(spf_ctx, g, roots, 0)


@both_engines
@pytest.mark.parametrize("seed", range(6))
def test_zero_cost_router_links_exact_path(spf_ctx, seed):
    """Zero-cost router links make the reference's pop order dynamic: flagged roots go through
    the sequential exact kernel and must still match bit for bit."""
    g = synth.random_lsdb(50, 6, 3.0, 200 + seed, metric_hi=3, zero_cost_router_links=True)
    roots = np.arange(6, 6 + 30, dtype=np.uint32)
    check(spf_ctx, g, roots, 0)


@both_engines
@pytest.mark.parametrize("seed", range(4))
def test_hopcount_mode(spf_ctx, seed):
    """MetricMode::HopCount (holo-isis/src/flooding/manet.rs:59-69): cost 0 to pseudonodes, 1 to
    routers, overload ignored, local = false."""
    g = synth.random_lsdb(50, 8, 2.5, 300 + seed, hopcount=True)
    roots = np.arange(8, 8 + 20, dtype=np.uint32)
    check(spf_ctx, g, roots, E.RUN_IGNORE_OVERLOAD)


@both_engines
@pytest.mark.parametrize("seed", range(4))
def test_forced_exact_and_pop_rank(spf_ctx, seed):
    g = synth.random_lsdb(40, 5, 3.0, 400 + seed, metric_hi=5)
    roots = np.arange(5, 5 + 10, dtype=np.uint32)
    res, ref = check(spf_ctx, g, roots, E.RUN_POP_RANK)
    assert res.stats["n_exact_roots"] == 0 and res.stats["n_repaired_roots"] == 10      # round 6: the ranks come out of the parallel path
    res, ref = check(spf_ctx, g, roots, E.RUN_FORCE_EXACT | E.RUN_POP_RANK)
    assert res.stats["n_exact_roots"] == 10


@both_engines
def test_standard_metric_max_path_prune(spf_ctx):
    """MAX_PATH_METRIC_STANDARD = 1023 (holo-isis/src/spf.rs:45, 637-647): a long chain is cut."""
    n = 40
    src = np.arange(n - 1); dst = src + 1
    s = np.concatenate([src, dst]); d = np.concatenate([dst, src])
    m = np.full(2 * (n - 1), 63)
    row_ptr, col, metric = synth._csr_from_links(n, s, d, m)
    g = synth.CsrGraph(row_ptr, col, metric, np.zeros(n, np.uint8), synth.MAX_PATH_METRIC_STANDARD)
    res, ref = check(spf_ctx, g, [0, n - 1, n // 2])
    assert (res.dist[0] != E.DIST_INF).sum() == 1023 // 63 + 1


@both_engines
def test_ospf_saturation_goes_exact(spf_ctx):
    """u32 saturating add (holo-ospf/src/spf.rs:672) is only representable on the exact path."""
    n = 4
    s = np.array([0, 1, 1, 2, 2, 3]); d = np.array([1, 0, 2, 1, 3, 2])
    m = np.array([0xFFFFFFF0, 1, 0x20, 1, 5, 1])
    row_ptr, col, metric = synth._csr_from_links(n, s, d, m)
    g = synth.CsrGraph(row_ptr, col, metric, np.zeros(n, np.uint8), synth.MAX_PATH_METRIC_OSPF)
    res, ref = check(spf_ctx, g, [0, 3], E.RUN_NET_NEXTHOPS)
    assert res.dist[0, 2] == 0xFFFFFFFF and (res.flags[0, 2] & 1)


def test_config_ospf_500_and_10k(spf_ctx):
    for g in (synth.ospf_500(), synth.ospf_10k()):
        check(spf_ctx, g, g.meta["roots"], E.RUN_NET_NEXTHOPS, expect_exact=False)


def test_config_isis_100k_all_roots(spf_ctx):
    """Headline graph, 64 roots on the GPU; ALL 64 bit for bit against the oracle (heap variant, roots dealt to the
    host's cores)."""
    g = synth.isis_100k()
    roots = np.asarray(g.meta["roots"], np.uint32)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    res = spf_ctx.run(G, roots, 0)
    G.free()
    assert res.stats["n_exact_roots"] == 0
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP,
                 mask_words_=res.first_hop_mask.shape[2], threads=ORACLE_THREADS)
    assert np.array_equal(res.dist, ref.dist)
    assert np.array_equal(res.hops, ref.hops)
    assert np.array_equal(res.flags & 1, ref.flags)
    assert np.array_equal(res.first_hop_mask, ref.mask)
    # size-independent properties over all 64 roots
    N = g.n
    assert (res.dist[np.arange(64), roots] == 0).all()
    assert ((res.flags & 1) == 1).all()                       # connected graph: everything reached
    # triangle inequality on every kept link: dist[t] <= dist[u] + w   (fixed point of relaxation)
    u = np.repeat(np.arange(N), np.diff(g.row_ptr).astype(np.int64))
    for r in range(0, 64, 9):
        assert (res.dist[r][g.col].astype(np.int64) <= res.dist[r][u].astype(np.int64) + g.metric).all()
    # symmetry-free check: first-hop mask non-empty exactly for non-root vertices
    nz = res.first_hop_mask.any(axis=2)
    assert (nz.sum(axis=1) == N - 1).all()


@both_engines
def test_determinism(spf_ctx):
    """Replay (holo-tools/holo-replay) needs bit-identical reruns."""
    g = synth.ospf_10k()
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.arange(0, 10000, 157, dtype=np.uint32)
    a = spf_ctx.run(G, roots, E.RUN_NET_NEXTHOPS)
    b = spf_ctx.run(G, roots, E.RUN_NET_NEXTHOPS)
    G.free()
    for f in ("dist", "hops", "flags", "first_hop_mask"):
        assert np.array_equal(getattr(a, f), getattr(b, f))


def test_errors_are_codes_not_crashes(spf_ctx):
    g = synth.ospf_500()
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    with pytest.raises(E.HspfError) as ei:
        spf_ctx.run(G, [g.n + 5])
    assert ei.value.code == -1
    with pytest.raises(E.HspfError):
        spf_ctx.upload(g.row_ptr, g.col + 100000, g.metric, g.vflags, g.max_path_metric)
    G.free()


def _chain(n, metric, max_path=synth.MAX_PATH_METRIC_WIDE):
    src = np.arange(n - 1); dst = src + 1
    s = np.concatenate([src, dst]); d = np.concatenate([dst, src])
    row_ptr, col, met = synth._csr_from_links(n, s, d, np.full(2 * (n - 1), metric))
    return synth.CsrGraph(row_ptr, col, met, np.zeros(n, np.uint8), max_path)


@sweeps_engine
def test_narrow_state_hops_overflow_falls_back_to_wide(spf_ctx):
    """The 4-byte fused state has 7 hop bits: a 300-router chain leaves the field, the run must be
    redone with the 8-byte state and still be exact."""
    g = _chain(300, 1)
    res, ref = check(spf_ctx, g, [0, 150, 299], expect_exact=False)
    assert res.hops.max() == 299


@sweeps_engine
def test_narrow_state_distance_overflow_falls_back_to_wide(spf_ctx):
    g = _chain(120, 1 << 20)            # 2 first-hop slots -> 23 distance bits; 119 * 2^20 does not fit
    res, ref = check(spf_ctx, g, [0, 60], expect_exact=False)
    assert int(res.dist[0].max()) == 119 << 20


@sweeps_engine
def test_costs_too_large_for_narrow_state_use_wide_directly(spf_ctx):
    g = _chain(50, 0x00FFFFFE)          # MAX_LINK_METRIC_WIDE - 1 (holo-isis/src/spf.rs:49)
    check(spf_ctx, g, [0, 49], expect_exact=False)


def test_two_contexts_on_two_threads(spf_ctx):
    """holo runs one OS thread per protocol instance (holo-protocol/src/lib.rs:427-430): distinct
    contexts must work concurrently (own stream, own scratch) and give the same answers."""
    import threading
    g = synth.ospf_10k()
    roots = np.arange(0, 10000, 79, dtype=np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, go.RUN_NET_NEXTHOPS, go.HEAP)
    out, errs = {}, []

    def work(tag):
        try:
            ctx = E.SpfContext(0)
            G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            for _ in range(3):
                out[tag] = ctx.run(G, roots, E.RUN_NET_NEXTHOPS)
            G.free(); ctx.close()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errs, errs
    for tag in (0, 1):
        assert np.array_equal(out[tag].dist, ref.dist) and np.array_equal(out[tag].hops, ref.hops)
        assert np.array_equal(out[tag].first_hop_mask, ref.mask[:, :, :out[tag].first_hop_mask.shape[2]])


@all_engines
def test_fattree_two_mask_words_two_phase_path(spf_ctx):
    """configs[4] shape at reduced k: an edge switch of a k=16 fat-tree has 16 first-hop slots with
    its 8 hosts... scaled: k=40 -> 40 slots per edge switch, core switches 40: > 16 slots selects the
    two-phase path (k_relax + k_dag<W>), unit metrics = maximal ECMP."""
    g = synth.isis_fattree(k=40)
    roots = np.asarray(g.meta["roots"][:24], np.uint32)
    res, ref = check(spf_ctx, g, roots, 0, oracle_variant=go.HEAP, expect_exact=False)
    assert res.stats["state_bytes"] == 0                      # not the packed-state path: k_fw, or k_relax + k_dag


@both_engines
def test_single_vertex_and_isolated_root(spf_ctx):
    g = synth.CsrGraph(np.zeros(2, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(1, np.uint8),
                       synth.MAX_PATH_METRIC_WIDE)
    res, ref = check(spf_ctx, g, [0])
    assert res.dist[0, 0] == 0 and res.hops[0, 0] == 0
    # three vertices, one of them with no links at all
    row_ptr, col, met = synth._csr_from_links(3, np.array([0, 1]), np.array([1, 0]), np.array([5, 7]))
    g = synth.CsrGraph(row_ptr, col, met, np.zeros(3, np.uint8), synth.MAX_PATH_METRIC_WIDE)
    res, ref = check(spf_ctx, g, [0, 1, 2])
    assert res.dist[2, 0] == E.DIST_INF and (res.flags[2] & 1).sum() == 1


@all_engines
@pytest.mark.parametrize("fanout", [70, 150, 200, 330])       # 2, 3, 4 and 6 mask words (3 and 6: not a power of two)
def test_star_rows_with_more_than_64_links(spf_ctx, fanout):
    """The hub row has `fanout` in- and out-links: multi-chunk general row routine and the
    more-than-64 wake-up path of the fused kernel; leaf roots keep the run on the fused path (1 slot),
    the hub as a root needs ceil(fanout/64) mask words (two-phase path)."""
    hub = 0
    leaves = np.arange(1, fanout + 1)
    chain = np.arange(fanout + 1, fanout + 30)                 # a tail behind leaf 1 so that hops grow
    s = np.concatenate([np.full(fanout, hub), leaves, [1], [chain[0]], chain[:-1], chain[1:]])
    d = np.concatenate([leaves, np.full(fanout, hub), [chain[0]], [1], chain[1:], chain[:-1]])
    rng = np.random.default_rng(fanout)
    m = rng.integers(1, 4, len(s))
    n = fanout + 30
    row_ptr, col, met = synth._csr_from_links(n, s, d, m)
    g = synth.CsrGraph(row_ptr, col, met, np.zeros(n, np.uint8), synth.MAX_PATH_METRIC_WIDE)
    res, ref = check(spf_ctx, g, [1, 2, int(chain[-1]), 5], expect_exact=False)
    assert res.stats["state_bytes"] in (4, 8) or spf_ctx.mode == "widemask"
    res, ref = check(spf_ctx, g, [hub, 3], expect_exact=False)
    assert res.first_hop_mask.shape[2] == (fanout + 63) // 64


@all_engines
@pytest.mark.parametrize("n,hubs", [(530, (0, 17, 300, 529)), (96, (95,)), (1000, (15, 16, 31, 32, 999)), (40, (3,))])
def test_heavy_chunks_become_one_row_per_wave_work_units(spf_ctx, n, hubs):
    """Hubs of 40-130 links at the first / last / chunk-boundary vertices of a sparse random graph (n not a multiple of
    16): chunks with a row of more than 32 in-links are cut into one-row-per-wave work units and every XCD takes an even
    share of both unit classes (GraphDev::unit_first); results of every kernel that walks units — fused, wide-mask,
    two-phase, lane = vertex — against the oracle, roots on and off the hubs."""
    rng = np.random.default_rng(n)
    s = list(rng.integers(0, n, 2 * n)); d = list(rng.integers(0, n, 2 * n))
    for h in hubs:
        fan = int(rng.integers(40, 131))
        nb = rng.choice(np.setdiff1d(np.arange(n), [h]), size=min(fan, n - 1), replace=False)
        s += [h] * len(nb); d += list(nb)
    s, d = np.asarray(s), np.asarray(d)
    keep = s != d
    s, d = s[keep], d[keep]
    s, d = np.concatenate([s, d]), np.concatenate([d, s])              # both directions: every link is two-way
    m = rng.integers(1, 6, len(s))
    row_ptr, col, met = synth._csr_from_links(n, s, d, m)
    g = synth.CsrGraph(row_ptr, col, met, np.zeros(n, np.uint8), synth.MAX_PATH_METRIC_WIDE)
    leaf_roots = [int(v) for v in rng.choice(np.setdiff1d(np.arange(n), hubs), size=min(70, n - len(hubs)), replace=False)]
    check(spf_ctx, g, leaf_roots, expect_exact=False)                   # <= 24 slots mostly: fused path
    check(spf_ctx, g, list(hubs) + leaf_roots[:5], expect_exact=False)  # hubs as roots: wide masks (k_fw / two-phase)
    check(spf_ctx, g, [leaf_roots[0]], expect_exact=False)


@pytest.mark.parametrize("rows,cols,chords,maxpath", [(20, 25, 0, synth.MAX_PATH_METRIC_OSPF), (8, 9, 20, synth.MAX_PATH_METRIC_WIDE),
                                                      (30, 33, 60, synth.MAX_PATH_METRIC_OSPF), (1, 2, 0, synth.MAX_PATH_METRIC_WIDE),
                                                      (5, 5, 0, 40)])
def test_lean_graphs_take_the_register_resident_kernel(spf_ctx, rows, cols, chords, maxpath):
    """Router-only graphs without row flags and with in-degrees <= 8 (hspf_graph::lean: the reference's 500-router
    benchmark shape) run k_single_lean: one vertex per thread, links in registers, free-running sweeps.  Corner, centre
    and every-vertex root sets, tie-heavy costs, a tight max path metric; against the oracle."""
    n = rows * cols
    links = synth._grid4_links(rows, cols)
    if chords:
        links = synth._add_chords(n, links, len(links) + chords, 17 + n)
    g = synth._routers_only(n, links, 99 + n, 1, 4, maxpath, f"lean-{n}", {})
    deg = np.diff(g.row_ptr.astype(np.int64))
    if deg.max() > 8:
        pytest.skip("a chord made a vertex too wide for the lean kernel")
    for roots in ([0], [n - 1, n // 2], list(range(n))[:96]):
        res, ref = check(spf_ctx, g, roots, expect_exact=False)
        # (up to eight roots the graph tries k_xcd, one XCD per root, next to the one-workgroup kernel and keeps the faster)
        assert res.stats["single_wg"] == 1 or (len(roots) <= 8 and res.stats["single_wg"] == 2)


@sweeps_engine
def test_scratch_prefilled_for_the_next_run_is_only_taken_when_it_fits(spf_ctx):
    """A fused run leaves the NEXT run's scratch filled behind its results (state, stamps, per-batch row flags of ITS
    graph).  Alternating two graphs of the same size whose row flags differ (overloaded sources / none), a patch in
    between, another root count and the other state width: every run against the oracle."""
    ga = synth.random_lsdb(300, 12, 3.0, 4242, metric_hi=5, p_overload=0.25)
    gb = synth.random_lsdb(300, 12, 3.0, 4243, metric_hi=5, p_overload=0.0)
    assert ga.n == gb.n
    roots = list(range(12, 12 + 64))
    Ga = spf_ctx.upload(ga.row_ptr, ga.col, ga.metric, ga.vflags, ga.max_path_metric)
    Gb = spf_ctx.upload(gb.row_ptr, gb.col, gb.metric, gb.vflags, gb.max_path_metric)
    try:
        for it in range(3):
            for G, g in ((Ga, ga), (Gb, gb), (Gb, gb), (Ga, ga)):
                for rs in (roots, roots[:40], roots + roots[:30]):
                    res = spf_ctx.run(G, rs, 0)
                    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, np.asarray(rs, np.uint32), 0, go.HEAP,
                                 mask_words_=res.first_hop_mask.shape[2])
                    assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops)
                    assert np.array_equal(res.first_hop_mask, ref.mask)
            # replace one row of ga: same vertex count, other row flags
            v = 20 + it
            cols = ga.col[ga.row_ptr[v]:ga.row_ptr[v + 1]].copy(); mets = ga.metric[ga.row_ptr[v]:ga.row_ptr[v + 1]].copy() + 1
            nf = np.uint8(int(ga.vflags[v]) ^ synth.VF_NO_TRANSIT)
            Ga.patch([v], [(cols, mets)], [nf])
            ga = synth.CsrGraph(Ga.row_ptr.copy(), Ga.col.copy(), Ga.metric.copy(), Ga.vflags.copy(), ga.max_path_metric)
    finally:
        Ga.free(); Gb.free()


def _properties(g, roots, res, sample, variant=go.HEAP, run_flags=0):
    """Full-size check: the roots in `sample` (None: ALL of them) bit for bit against the oracle, all roots through
    size-independent properties (root at distance 0, fixed point of relaxation on every kept link,
    hops 0 only at the root, first-hop mask non-empty exactly for reached non-root vertices)."""
    roots = np.asarray(roots, np.uint32)
    if sample is None:
        sample = np.arange(len(roots))
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots[sample], run_flags, variant,
                 mask_words_=res.first_hop_mask.shape[2], threads=ORACLE_THREADS)
    assert np.array_equal(res.dist[sample], ref.dist)
    assert np.array_equal(res.hops[sample], ref.hops)
    assert np.array_equal(res.flags[sample] & 1, ref.flags)
    assert np.array_equal(res.first_hop_mask[sample], ref.mask)
    R = len(roots)
    assert (res.dist[np.arange(R), roots] == 0).all()
    u = np.repeat(np.arange(g.n), np.diff(g.row_ptr).astype(np.int64))
    for r in range(0, R, max(1, R // 8)):
        d = res.dist[r].astype(np.int64)
        assert (d[g.col] <= d[u] + g.metric).all()
        reached = (res.flags[r] & 1) == 1
        assert (res.first_hop_mask[r].any(axis=1) == (reached & (np.arange(g.n) != roots[r]))).all()
        assert ((res.hops[r] == 0) & reached).sum() == 1


def test_config_fattree_262k_two_mask_words(spf_ctx):
    """BASELINE configs[4] at full size: k=100 fat-tree, 262 500 vertices / 1 500 000 entries, unit
    metrics (maximal ECMP), self = an edge switch with 100 first-hop slots (2 mask words) + its 100
    neighbours as roots."""
    g = synth.isis_fattree(100)
    roots = np.asarray(g.meta["roots"], np.uint32)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    res = spf_ctx.run(G, roots, 0)
    G.free()
    assert res.first_hop_mask.shape[2] == 2 and res.stats["n_exact_roots"] == 0
    _properties(g, roots, res, None)                           # all 101 roots against the oracle


def test_config_multi_area_10k_roots(spf_ctx):
    """BASELINE configs[3] shape: 10 areas x 5 000 routers, 1 000 roots per area (16 wavefront
    batches per run, fused path), OSPF semantics."""
    total = 0
    for g in synth.ospf_multi_area():                          # all 10 areas, every root against the oracle
        roots = np.asarray(g.meta["roots"], np.uint32)
        G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        res = spf_ctx.run(G, roots, E.RUN_NET_NEXTHOPS)
        G.free()
        assert res.stats["n_exact_roots"] == 0 and res.stats["state_bytes"] in (4, 8)
        _properties(g, roots, res, None, run_flags=go.RUN_NET_NEXTHOPS)
        total += len(roots)
    assert total == 10000


def test_root_groups_bound_the_scratch(spf_ctx):
    """More (vertex, root) pairs than one pass may hold (2^26): the batch axis is split into groups;
    the caller sees one result.  80 000 x 1 100 roots = 88 M pairs -> 2 groups."""
    n = 80000
    links = synth._grid8_links(200, 400)
    g = synth._routers_only(n, links, 12345, 1, 9, synth.MAX_PATH_METRIC_WIDE, "grid-80k", {})
    roots = (np.arange(1100, dtype=np.uint32) * 71) % n
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    res = spf_ctx.run(G, roots, 0, want_mask=True)
    G.free()
    assert res.stats["n_roots"] == 1100 and res.stats["n_batches"] == 18
    _properties(g, roots, res, [0, 831, 832, 1099])


@both_engines
@pytest.mark.parametrize("seed", range(4))
def test_hopcount_lan_graphs_stay_on_the_parallel_path(spf_ctx, seed):
    """MetricMode::HopCount on an LSDB with LAN pseudonodes (flooding::manet::init_cache,
    holo-isis/src/flooding/manet.rs:47-69): links into pseudonodes cost 0 and come from
    higher-numbered routers, i.e. the reference's pop order is NOT the static one — the fused kernel
    resolves that plateau shape itself (one parent: the lowest-numbered router of the same distance),
    no root may fall back to the sequential kernel."""
    g = synth.random_lsdb(300, 40, 2.5, 700 + seed, hopcount=True, p_overload=0.0, p_noexpand=0.02)
    roots = np.arange(40, 40 + 70, dtype=np.uint32)
    check(spf_ctx, g, roots, E.RUN_IGNORE_OVERLOAD, expect_exact=False)            # > 16 slots: two-phase path
    few = [int(r) for r in roots if G_slots(g, int(r)) <= 16][:40]
    if few:
        res, _ = check(spf_ctx, g, few, E.RUN_IGNORE_OVERLOAD, expect_exact=False)  # fused path
        assert res.stats["state_bytes"] in (4, 8)


def G_slots(g, root):
    """first-hop slots of a root (include/holo_spf_hip.h): its links + the links of the network vertices
    reachable from it through networks only."""
    tot, seen, q = int(g.row_ptr[root + 1] - g.row_ptr[root]), {root}, [root]
    while q:
        p = q.pop()
        for k in range(int(g.row_ptr[p]), int(g.row_ptr[p + 1])):
            t = int(g.col[k])
            back = (g.col[g.row_ptr[t]:g.row_ptr[t + 1]] == p).any()
            if t not in seen and (g.vflags[t] & 1) and back:
                seen.add(t); q.append(t); tot += int(g.row_ptr[t + 1] - g.row_ptr[t])
    return tot


@sweeps_engine
@pytest.mark.parametrize("run_flags", [0, E.RUN_NET_NEXTHOPS])
def test_many_roots_regrouped_by_state_class(spf_ctx, run_flags):
    """Every router as a root on a graph where a few roots need wide masks (members of a 40-router LAN: 40+ first-hop
    slots -> two-phase path) and most need few: the call regroups the roots by the state they need, runs each class on
    its own and puts the rows back in the caller's order — results identical to the oracle's, row by row."""
    g = synth.random_lsdb(420, 2, 3.0, 4242, metric_hi=5, lan_size=40)
    roots = np.arange(2, g.n, dtype=np.uint32)
    rng = np.random.default_rng(1)
    rng.shuffle(roots)                                   # classes interleaved in the caller's order
    roots[17] = E.NO_ROOT
    res, ref = check(spf_ctx, g, roots, run_flags)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    tot = np.array([G.slot_table(int(r))[2] if r != E.NO_ROOT else 0 for r in roots])
    G.free()
    assert (tot > 16).sum() >= 20 and (tot <= 16).sum() >= 256        # the mixture this test is about
    assert res.stats["state_bytes"] in (4, 8)                # the packed-state class ran (the wide-mask class: k_fw or two-phase)
    assert res.stats["n_roots"] == len(roots)


@sweeps_engine
def test_many_roots_regrouped_device_outputs(spf_ctx):
    """Same regrouping through hspf_run_device (rows permuted straight into the caller's device buffers)."""
    import torch
    g = synth.random_lsdb(420, 2, 3.0, 777, metric_hi=5, lan_size=40)
    roots = np.arange(2, g.n, dtype=np.uint32)[::-1].copy()
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    W = G.mask_words(roots)
    R, n = len(roots), g.n
    dev = torch.device("cuda:0")
    dist = torch.empty((R, n), dtype=torch.int32, device=dev); hops = torch.empty((R, n), dtype=torch.int16, device=dev)
    flags = torch.empty((R, n), dtype=torch.int16, device=dev); mask = torch.empty((R, n, W), dtype=torch.int64, device=dev)
    spf_ctx.run_device(G, roots, 0, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                       mask_ptr=mask.data_ptr(), mask_words=W)
    torch.cuda.synchronize()
    G.free()
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.MAP, mask_words_=W)
    assert np.array_equal(dist.cpu().numpy().view(np.uint32), ref.dist)
    assert np.array_equal(hops.cpu().numpy().view(np.uint16), ref.hops)
    assert np.array_equal(flags.cpu().numpy().view(np.uint16) & 1, ref.flags)
    assert np.array_equal(mask.cpu().numpy().view(np.uint64), ref.mask)


@both_engines
@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("run_flags", [0, E.RUN_NET_NEXTHOPS])
def test_roots_with_17_to_24_slots_stay_on_the_fused_path(spf_ctx, seed, run_flags):
    """Routers on a 14-member LAN: their own links + the LAN's members give 17-24 first-hop slots — the 8-byte packed
    state with a 17..24-bit mask field (fewer hop bits) instead of the two-phase path."""
    g = synth.random_lsdb(120, 3, 2.0, 1700 + seed, metric_hi=5, lan_size=14, p_parallel=0.0)
    roots = [r for r in range(3, g.n) if 16 < G_slots(g, r) <= 24][:40]
    assert len(roots) >= 8
    res, _ = check(spf_ctx, g, roots, run_flags, expect_exact=False)
    assert res.stats["state_bytes"] == 8 and res.stats["n_dag_launches"] == 0


@all_engines
def test_wide_mask_hop_field_overflow_falls_back_to_two_phase(spf_ctx):
    """A hub with 24 neighbours (24 slots -> 8 hop bits) and a 300-router tail: hop counts beyond 255 raise the
    overflow flag, the run is redone on the two-phase path (u16 hops) and the graph remembers."""
    fan, tail = 24, 300
    hub = 0
    leaves = np.arange(1, fan + 1)
    chain = np.arange(fan + 1, fan + 1 + tail)
    s = np.concatenate([np.full(fan, hub), leaves, [1], [chain[0]], chain[:-1], chain[1:]])
    d = np.concatenate([leaves, np.full(fan, hub), [chain[0]], [1], chain[1:], chain[:-1]])
    m = np.ones(len(s), np.int64)
    n = fan + 1 + tail
    row_ptr, col, met = synth._csr_from_links(n, s, d, m)
    g = synth.CsrGraph(row_ptr, col, met, np.zeros(n, np.uint8), synth.MAX_PATH_METRIC_WIDE)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        for attempt in range(2):
            res = spf_ctx.run(G, np.array([hub, 5], np.uint32), 0)
            ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, np.array([hub, 5], np.uint32), 0, go.MAP,
                         mask_words_=res.first_hop_mask.shape[2])
            assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops)
            assert np.array_equal(res.first_hop_mask, ref.mask) and np.array_equal(res.flags & 1, ref.flags)
            assert int(res.hops.max()) > 255
            assert res.stats["state_bytes"] == 0                 # ended on the u16-hops path (k_fw, or k_relax + k_dag)
    finally:
        G.free()


@both_engines
@pytest.mark.parametrize("hopcount", [False, True])
@pytest.mark.parametrize("run_flags", [0, E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD])
def test_giant_rows_are_evaluated_in_slices(spf_ctx, hopcount, run_flags):
    """Three LANs of 700 routers among 3 000 (pseudonode rows of ~700 in-links: three slices each, k_giant_part), overloaded
    members, roots off the LANs (one mask word: the packed fused path), one batch and three ragged ones; then a row patch
    that shrinks one LAN below the giant threshold."""
    g = synth.random_lsdb(3000, 3, 2.5, 4242, metric_hi=9, lan_size=700, p_overload=0.05, hopcount=hopcount)
    if hopcount:
        run_flags |= E.RUN_IGNORE_OVERLOAD
    on_lan = np.zeros(g.n, bool)
    for net in range(3):
        on_lan[g.col[g.row_ptr[net]:g.row_ptr[net + 1]]] = True
    off = np.nonzero(~on_lan)[0]
    off = off[off >= 3].astype(np.uint32)
    assert len(off) > 300
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert int(np.diff(G.export("in_ptr")).max()) > 256 and int(((G.export("rowflags") & 16) != 0).sum()) == 3
        for roots in (off[:64], off[40:40 + 170]):
            res = spf_ctx.run(G, roots, run_flags)
            ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, run_flags & 3, go.HEAP,
                         mask_words_=res.first_hop_mask.shape[2])
            assert res.first_hop_mask.shape[2] == 1
            for f in ("dist", "hops", "first_hop_mask"):
                assert np.array_equal(getattr(res, f), getattr(ref, "mask" if f == "first_hop_mask" else f)), f
        a, b = int(g.row_ptr[1]), int(g.row_ptr[2])
        G.patch([1], [(g.col[a:a + 100].copy(), g.metric[a:a + 100].copy())], [g.vflags[1]])
        assert int(((G.export("rowflags") & 16) != 0).sum()) == 2
        g2 = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
        res = spf_ctx.run(G, off[:64], run_flags)
        ref = go.run(g2.row_ptr, g2.col, g2.metric, g2.vflags, g2.max_path_metric, off[:64], run_flags & 3, go.HEAP, mask_words_=1)
        assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops) and np.array_equal(res.first_hop_mask, ref.mask)
    finally:
        G.free()


# ---- the lean sweep (k_fused_lean): taken where its 4-byte state fits, left cleanly where it does not ------------------

@sweeps_engine
def test_lean_sweep_is_taken_and_bit_identical(spf_ctx):
    """The sweep engine on adversarial LSDBs: wherever the run's 4-byte state fits it must be k_fused_lean's
    (hspf_stats.dbg[0]) and agree with the oracle — rows with odd and even link counts (pad links), more than 16 links,
    none at all, network vertices, overloaded sources and zero-cost links (the general routine next to the fast one)."""
    lean = 0
    for seed in range(10):
        g = synth.random_lsdb(300, 20, 3.0 + (seed % 3), 4000 + seed, metric_hi=9, zero_cost_router_links=(seed % 4 == 3))
        roots = np.arange(20, 20 + 64 + seed, dtype=np.uint32)
        res, _ = check(spf_ctx, g, roots, E.RUN_NET_NEXTHOPS if seed & 1 else 0)
        if res.stats["state_bytes"] == 4:
            assert (res.stats["dbg"][0] & 1) == 1, seed
            lean += 1
    assert lean >= 2


@sweeps_engine
def test_lean_sweep_on_grids_with_every_degree(spf_ctx):
    """8-neighbour grid + chords: in-degrees 3 .. 12+, ties everywhere (costs 1 .. 3): the first-discoverer rule through
    the tag field, both halves of rows with more than 8 links."""
    n = 40 * 50
    links = synth._add_chords(n, synth._grid8_links(40, 50), 9000, 77)
    g = synth._routers_only(n, links, 78, 1, 3, synth.MAX_PATH_METRIC_WIDE, "grid-ties", {})
    roots = ((np.arange(100, dtype=np.uint64) * n) // 100).astype(np.uint32)
    res, _ = check(spf_ctx, g, roots)
    assert (res.stats["dbg"][0] & 1) == 1


@sweeps_engine
def test_lean_sweep_overflow_falls_back_to_k_fused_and_the_graph_remembers(spf_ctx):
    """Distances beyond the lean state's distance field (three bits go to the tag): k_emit_fused raises LF_OVERFLOW on
    the final words, the run is redone by k_fused with identical results, and the next run on the graph goes there
    directly."""
    n = 120
    a = np.arange(n - 1, dtype=np.int64)
    links = np.stack([a, a + 1], axis=1)
    g = synth._routers_only(n, links, 5, 20000, 20000, synth.MAX_PATH_METRIC_WIDE, "long-chain", {})
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        roots = np.array([0, 1, 60, 119], np.uint32)
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.MAP)
        for it in range(2):
            res = spf_ctx.run(G, roots, 0)
            assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops)
            assert np.array_equal(res.first_hop_mask, ref.mask[:, :, :res.first_hop_mask.shape[2]])
            assert res.stats["dbg"][0] == 0                      # the results come from k_fused
        assert int(ref.dist.max()) > (1 << 20)
    finally:
        G.free()


@sweeps_engine
def test_lean_sweep_hop_field_saturation_is_an_overflow(spf_ctx):
    """A chain longer than the hop field: the lean sweep's hop count saturates (no per-row test any more), the final-word
    check catches it, and the wider states give the oracle's hops."""
    n = 300
    a = np.arange(n - 1, dtype=np.int64)
    g = synth._routers_only(n, np.stack([a, a + 1], axis=1), 6, 1, 2, synth.MAX_PATH_METRIC_WIDE, "deep-chain", {})
    res, ref = check(spf_ctx, g, np.array([0, 150, 299], np.uint32))
    assert int(ref.hops.max()) == 299 and res.stats["dbg"][0] == 0


# ---- leaves stay out of the wide-mask fixed point (k_fw<.., LEAF>, k_emit<W, true>) ------------------------------------

def _links_graph(n, pairs, costs, vflags=None):
    s = np.array([p[0] for p in pairs] + [p[1] for p in pairs]); d = np.array([p[1] for p in pairs] + [p[0] for p in pairs])
    m = np.array(list(costs) + list(costs), np.int64)
    row_ptr, col, met = synth._csr_from_links(n, s, d, m)
    return synth.CsrGraph(row_ptr, col, met, np.zeros(n, np.uint8) if vflags is None else np.asarray(vflags, np.uint8),
                          synth.MAX_PATH_METRIC_WIDE)


@pytest.mark.parametrize("spf_ctx", ["widemask", "default"], indirect=True)
def test_leaves_are_derived_in_the_emit(spf_ctx):
    """A core of 6 routers with hosts hanging off it (single-homed: leaves), a two-vertex component (both ends leaves),
    an overloaded router with its own host, a host behind a zero-cost link from a higher-numbered router; roots = hosts,
    core routers, both ends of the pair.  On the wide-mask path the leaves take no part in the sweeps
    (hspf_stats::dbg[1]) and every result equals the oracle's."""
    core = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0), (0, 3), (1, 4)]
    hosts = [(0, 6), (0, 7), (1, 8), (2, 9), (3, 10), (3, 11), (4, 12), (5, 13), (5, 14)]
    pair = [(15, 16)]
    pairs = core + hosts + pair
    rng = np.random.default_rng(5)
    costs = list(rng.integers(1, 4, len(core))) + list(rng.integers(1, 4, len(hosts))) + [2]
    vf = np.zeros(17, np.uint8); vf[2] = synth.VF_NO_TRANSIT
    g = _links_graph(17, pairs, costs, vf)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert G.export("leaf").tolist() == [0] * 6 + [1] * 11
    finally:
        G.free()
    for roots in ([6, 0, 9, 15, 16, 3, 14], list(range(17)), [9], [2]):
        for fl in (0, E.RUN_IGNORE_OVERLOAD):
            res, _ = check(spf_ctx, g, roots, fl, expect_exact=False)
            assert (res.stats["dbg"][1] >> 31) == 1 or spf_ctx.mode != "widemask"
    # zero-cost links: host 6 behind a zero-cost link (host < router and host > router both exist: 6 > 0, and 16 > 15)
    m0 = g.metric.copy()
    rp = g.row_ptr.astype(np.int64)
    src = np.repeat(np.arange(g.n), np.diff(rp))
    m0[((src == 0) & (g.col == 6)) | ((src == 6) & (g.col == 0)) | ((src == 16) & (g.col == 15))] = 0
    g0 = synth.CsrGraph(g.row_ptr, g.col, m0, g.vflags, g.max_path_metric)
    check(spf_ctx, g0, [6, 0, 9, 15, 16, 3, 14])
    check(spf_ctx, g0, list(range(17)))


@pytest.mark.parametrize("spf_ctx", ["widemask", "default"], indirect=True)
@pytest.mark.parametrize("seed", range(6))
def test_stub_lans_and_hosts_random(spf_ctx, seed):
    """random_lsdb plus 40 single-homed hosts and 10 stub LANs (a pseudonode with one attached router: cost c in,
    0 out); roots = some hosts, some routers, a stub LAN's router — also with next hops reported for networks."""
    g = synth.random_lsdb(60, 8, 3.0, 7700 + seed, metric_hi=5)
    rng = np.random.default_rng(seed)
    n0 = g.n
    rp = g.row_ptr.astype(np.int64)
    s0 = np.repeat(np.arange(n0), np.diff(rp)); d0 = g.col.astype(np.int64); m0 = g.metric.astype(np.int64)
    # vertices are renumbered: stub LANs must sort with the networks (first), hosts are routers (last)
    n_lan, n_host = 10, 40
    nn = g.meta["n_networks"]
    shift = lambda v: np.where(v < nn, v, v + n_lan)
    s0, d0 = shift(s0), shift(d0)
    lan_ids = nn + np.arange(n_lan); host_ids = n0 + n_lan + np.arange(n_host)
    routers = np.arange(nn + n_lan, n0 + n_lan)
    lr = rng.choice(routers, n_lan); hr = rng.choice(routers, n_host)
    lc = rng.integers(1, 5, n_lan); hc1 = rng.integers(1, 5, n_host); hc2 = rng.integers(1, 5, n_host)
    s = np.concatenate([s0, lr, lan_ids, hr, host_ids]); d = np.concatenate([d0, lan_ids, lr, host_ids, hr])
    m = np.concatenate([m0, lc, np.zeros(n_lan, np.int64), hc1, hc2])
    n = n0 + n_lan + n_host
    row_ptr, col, met = synth._csr_from_links(n, s, d, m)
    vf = np.zeros(n, np.uint8)
    vf[shift(np.arange(n0))] = g.vflags
    vf[lan_ids] = synth.VF_NETWORK
    g2 = synth.CsrGraph(row_ptr, col, met, vf, g.max_path_metric)
    roots = np.concatenate([host_ids[:12], routers[:20], lr[:3], host_ids[-2:]]).astype(np.uint32)
    for fl in (0, E.RUN_NET_NEXTHOPS):
        res, _ = check(spf_ctx, g2, roots, fl)
        assert (res.stats["dbg"][1] >> 31) == 1 or spf_ctx.mode != "widemask"


def test_fattree_full_size_leaves_deferred(spf_ctx):
    """configs[4] at full size (262 500 vertices, 250 000 of them single-homed hosts, 101 roots of which 50 are hosts):
    the product path is k_fw with the hosts derived in the emit; every (root, vertex) equals the oracle's."""
    g = synth.isis_fattree(100)
    roots = np.asarray(g.meta["roots"], np.uint32)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert int(G.export("leaf").sum()) == 250000
        res = spf_ctx.run(G, roots, 0)
    finally:
        G.free()
    assert res.stats["n_exact_roots"] == 0                    # (the host roots run as their own class on the packed path)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=res.first_hop_mask.shape[2],
                 threads=ORACLE_THREADS)
    assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops)
    assert np.array_equal(res.flags & 1, ref.flags) and np.array_equal(res.first_hop_mask, ref.mask)


# ---- the lean sweep's plan: head sweeps, dense stretch, all-due sweep, tail — decided on the device -------------------

@sweeps_engine
@pytest.mark.parametrize("shape", ["grid", "isis-100k", "ospf-10k x 1024 roots"])
def test_lean_plan_first_run_and_repeated_runs(spf_ctx, shape):
    """k_fused_lean's launches decide on the device whether they still have a job (head sweep skipped once the frontier
    covers the graph, dense pass skipped once the corrections thin out); the host only sizes the plan from the
    previous run.  Every run of the sequence — the first of a fresh graph handle, repeats, after a cost patch, after a
    structural patch, with other roots — equals the oracle bit for bit, and the dense stretch is there from the FIRST
    run (hspf_stats::dbg[1]: passes that did work | head sweeps that ran << 8 | planned << 16 / << 24)."""
    if shape == "grid":
        side = 30                                       # (hop counts stay inside the 4-byte state's 7 hop bits)
        idx = np.arange(side * side).reshape(side, side)
        a = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel(), idx[:-1, :-1].ravel()])
        b = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel(), idx[1:, 1:].ravel()])
        rng = np.random.default_rng(8)
        m = rng.integers(1, 30, len(a))
        row_ptr, col, met = synth._csr_from_links(side * side, np.concatenate([a, b]), np.concatenate([b, a]), np.concatenate([m, m]))
        g = synth.CsrGraph(row_ptr, col, met, np.zeros(side * side, np.uint8), synth.MAX_PATH_METRIC_WIDE)
        roots = (np.arange(64, dtype=np.int64) * g.n // 64).astype(np.uint32)
    elif shape == "isis-100k":                          # one batch, a pass of 6 250 workgroups: dense stretches as multi-pass launches
        g = synth.isis_100k()
        roots = (np.arange(64, dtype=np.int64) * g.n // 64).astype(np.uint32)
    else:                                               # 16 batches x 625 workgroups: a pass spans all batches
        g = synth.ospf_10k()
        roots = (np.arange(1024, dtype=np.int64) * g.n // 1024).astype(np.uint32)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)

    def one(rts, graph):
        res = spf_ctx.run(G, rts, 0)
        ref = go.run(graph.row_ptr, graph.col, graph.metric, graph.vflags, graph.max_path_metric, rts, 0, go.HEAP,
                     mask_words_=res.first_hop_mask.shape[2], threads=ORACLE_THREADS)
        assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops)
        assert np.array_equal(res.flags & 1, ref.flags) and np.array_equal(res.first_hop_mask, ref.mask)
        d = res.stats["dbg"][1]
        return res.stats, {"used": d & 0xFF, "head": (d >> 8) & 0xFF, "planned": (d >> 16) & 0xFF, "head_planned": (d >> 24) & 0x7F}
    try:
        plans = []
        for _ in range(5):
            st, pl = one(roots, g)
            assert st["state_bytes"] == 4 and (st["dbg"][0] & 1) == 1
            plans.append(pl)
        stamped_only = bool(int(os.environ.get("HSPF_VARIANT", "0"), 0) & 524288)
        if not stamped_only:
            assert all(p["used"] >= 1 and p["used"] <= p["planned"] for p in plans), plans     # dense passes from the first run on
            assert all(p["head"] <= p["head_planned"] for p in plans), plans
            # the plan settles where the corrections thin out within the longest stretch a plan may hold: isis-100k (random
            # chords: ~28 sweeps) on the same head sweeps and stretch from the third run on (+- a pass: sampled counters);
            # a chordless grid is corrected for as many sweeps as it is wide, its stretch simply grows to the cap
            if shape == "isis-100k":
                assert abs(plans[-1]["used"] - plans[-2]["used"]) <= 1 and plans[-1]["head"] == plans[-2]["head"], plans
            assert all(p["planned"] <= 62 for p in plans), plans
        u = g.n // 3
        a0, b0 = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
        G.patch([u], [(g.col[a0:b0], g.metric[a0:b0] + 3)], [g.vflags[u]])                     # costs only
        g2 = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
        _, pl = one(roots, g2)
        assert stamped_only or pl["used"] >= 1
        v = int(g.col[a0])                                                                      # structural: the link u - v goes away
        c0, d0 = int(G.row_ptr[v]), int(G.row_ptr[v + 1])
        keep_v = np.array([k for k in range(c0, d0) if int(G.col[k]) != u], dtype=np.int64)
        a1, b1 = int(G.row_ptr[u]), int(G.row_ptr[u + 1])
        keep_u = np.array([k for k in range(a1, b1) if int(G.col[k]) != v], dtype=np.int64)
        G.patch([u, v], [(G.col[keep_u], G.metric[keep_u]), (G.col[keep_v], G.metric[keep_v])], [g.vflags[u], g.vflags[v]])
        g3 = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
        _, pl = one(roots, g3)
        assert stamped_only or pl["used"] >= 1                                                  # the first run after the change: dense stretch there
        other = ((roots.astype(np.int64) + 11) % g.n).astype(np.uint32)
        _, pl = one(other, g3)
        assert stamped_only or pl["used"] >= 1                                                  # other roots: the same plan, decided on the device again
    finally:
        G.free()


# ---- leaf roots: rows derived from the neighbour's (run_classes / k_leaf_root_rows) ------------------------------------

def _caterpillar(seed):
    """A ring of routers with chords, each with stub routers hanging off it: single-homed stubs (leaves), a stub of a stub,
    a stub on two parallel links, a stub behind an overloaded router, a stub whose neighbour does not list it back, a stub
    on a tight max-path metric — every vertex is a root."""
    rng = np.random.default_rng(seed)
    ring = 30
    links = []                                                   # (u, v, cost u->v, cost v->u); None = one-way
    for i in range(ring):
        links.append((i, (i + 1) % ring, int(rng.integers(1, 9)), int(rng.integers(1, 9))))
    for _ in range(12):
        a, b = rng.choice(ring, 2, replace=False)
        links.append((int(a), int(b), int(rng.integers(1, 9)), int(rng.integers(1, 9))))
    n = ring
    stubs = []
    for i in range(ring):
        for _ in range(int(rng.integers(1, 4))):
            links.append((i, n, int(rng.integers(1, 9)), int(rng.integers(1, 9)))); stubs.append(n); n += 1
    links.append((stubs[0], n, 3, 2)); n += 1                    # a stub of a stub (its neighbour is no leaf: two kept links)
    links.append((1, n, 2, 2)); links.append((1, n, 5, 5)); n += 1   # two parallel links: not a leaf
    links.append((2, n, 4, None)); n += 1                        # the ring router does not list it back: one-way, tree = itself
    rows = [[] for _ in range(n)]
    for u, v, cuv, cvu in links:
        if cuv is not None: rows[u].append((v, cuv))
        if cvu is not None: rows[v].append((u, cvu))
    for r in rows:
        rng.shuffle(r)
    row_ptr = np.zeros(n + 1, np.uint32); row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.array([t for r in rows for t, _ in r], np.uint32); met = np.array([c for r in rows for _, c in r], np.uint32)
    vflags = np.zeros(n, np.uint8)
    vflags[3] |= synth.VF_NO_TRANSIT                            # stubs behind an overloaded router are not derived
    vflags[stubs[5]] |= synth.VF_NO_TRANSIT                      # an overloaded stub is still the root of its own tree
    return synth.CsrGraph(row_ptr, col, met, vflags, synth.MAX_PATH_METRIC_WIDE, "caterpillar", {})


@both_engines
@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("run_flags", [0, E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD])
def test_leaf_roots_derived_from_their_neighbour(spf_ctx, seed, run_flags):
    """Every vertex as a root (> 64 roots: the call goes through run_classes): the rows of single-homed stubs come from their
    neighbour's rows, everything else runs — all equal to the oracle's, also with a max-path metric that cuts the stubs'
    trees short of their neighbours'."""
    g = _caterpillar(seed)
    roots = np.arange(g.n, dtype=np.uint32)
    np.random.default_rng(seed).shuffle(roots)
    res, _ = check(spf_ctx, g, roots, run_flags)
    assert res.stats["n_roots"] == g.n
    assert res.stats["n_batches"] <= 1 + (g.n - 40) // 64        # the stubs (more than half of the vertices) took no batch
    g2 = synth.CsrGraph(g.row_ptr, g.col, g.metric, g.vflags, 14, g.name, g.meta)    # trees cut at distance 14
    check(spf_ctx, g2, roots, run_flags)
