"""Graph construction on the device (hspf_graph_upload) and incremental row replacement
(hspf_graph_patch): the exported device layout against the CPU restatement in
tests/_layout_ref.py, a patched graph against a fresh upload of the patched CSR array by array, and SPF on
the patched graph against the oracle."""
import numpy as np
import pytest

from holo_amd import synth
from holo_amd import engine as E
from oracle import graph_oracle as go
from _layout_ref import layout
from _engines import hub_engines, hubsort_engine

pytestmark = pytest.mark.gpu

BUILT = ("twoway", "in_ptr", "in_src", "in_cost", "in_pos", "out_ptr", "out_dst", "out_cost", "out_pos", "rowflags", "units", "leaf")
RAW = ("row_ptr", "col", "metric", "vflags")
DERIVED = ("ell_src", "ell_cost", "ell_out", "summary")      # no restatement: compared patched against fresh


def assert_layout(G, g):
    want = layout(g.row_ptr, g.col, g.metric, g.vflags)
    for name in BUILT:
        got = G.export(name)
        assert np.array_equal(got, want[name]), name
    for name in RAW:
        assert np.array_equal(G.export(name), getattr(g, name)), name
    assert np.array_equal(G.export("host_row_ptr"), g.row_ptr) and np.array_equal(G.export("host_col"), g.col)   # the host mirrors
    assert G.n_edges_kept == len(want["in_src"])


def graphs():
    yield synth.random_lsdb(60, 8, 3.0, 1, metric_hi=6)
    yield synth.random_lsdb(90, 10, 2.5, 2, metric_hi=3, p_oneway=0.3, p_parallel=0.4, p_noexpand=0.2, p_overload=0.3)
    yield synth.random_lsdb(50, 8, 2.5, 3, hopcount=True)
    yield synth.random_lsdb(40, 1, 3.0, 4, lan_size=40)                 # one LAN with 40 members: rows > 16 links
    yield synth.random_lsdb(3000, 100, 4.0, 5, metric_hi=2, zero_cost_router_links=True)   # several scan tiles
    yield synth.ospf_500()
    yield synth.random_lsdb(700, 9, 3.0, 6, lan_size=90)                # heavy chunks (work units) in the middle of the range


@pytest.mark.parametrize("i", range(7))
def test_device_layout_matches_restatement(spf_ctx, i):
    g = list(graphs())[i]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert_layout(G, g)
    finally:
        G.free()


@hubsort_engine
@pytest.mark.parametrize("i", range(7))
def test_hub_mode_layout_matches_restatement(spf_ctx, i):
    """Every graph through the sorted-key build (HSPF_HUB_DEG=0): the layout is the one the row scans give."""
    g = list(graphs())[i]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert int(G.export("build_mode")[0]) == 1
        assert_layout(G, g)
    finally:
        G.free()


def hub_lsdb(seed, n_leaves=3000, hub_links=2500, parallel=0):
    """Routers only: one router with `hub_links` point-to-point neighbours (a row far beyond HUB_DEG = 512 links), a sparse
    random mesh among the others, a few one-way links and, with `parallel`, that many routers each listing `parallel`
    links to one target that lists each of them once (in-degree beyond every out-degree)."""
    rng = np.random.default_rng(seed)
    n = n_leaves + 1
    hub = n // 3
    src, dst, met = [], [], []
    others = np.array([v for v in range(n) if v != hub])
    for v in rng.choice(others, size=hub_links, replace=False).tolist():
        src += [hub, v]; dst += [v, hub]; met += [int(rng.integers(1, 9)), int(rng.integers(1, 9))]
        if rng.random() < 0.05:                                   # a second, parallel link one way
            src.append(v); dst.append(hub); met.append(int(rng.integers(1, 9)))
    for _ in range(2 * n_leaves):
        u, v = rng.choice(others, size=2, replace=False).tolist()
        src.append(u); dst.append(v); met.append(int(rng.integers(1, 9)))
        if rng.random() > 0.1:
            src.append(v); dst.append(u); met.append(int(rng.integers(1, 9)))
    if parallel:
        t = int(others[7])
        for u in others[100:100 + parallel].tolist():
            src.append(t); dst.append(u); met.append(3)
            for _ in range(parallel):
                src.append(u); dst.append(t); met.append(int(rng.integers(1, 4)))
    src = np.array(src, np.int64); dst = np.array(dst, np.int64); met = np.array(met, np.int64)
    perm = rng.permutation(len(src))
    row_ptr, col, metric = synth._csr_from_links(n, src[perm], dst[perm], met[perm])
    vflags = np.zeros(n, np.uint8)
    vflags[rng.random(n) < 0.02] |= synth.VF_NO_TRANSIT
    vflags[hub] = 0
    return synth.CsrGraph(row_ptr, col, metric, vflags, synth.MAX_PATH_METRIC_WIDE, f"hub-{seed}", {"hub": hub})


def test_hub_row_takes_the_sorted_build(spf_ctx):
    """A 2 500-link row: the default context builds from sorted keys (mode 1), the layout is the restated one, SPF from
    leaves and from the hub's neighbours matches the oracle; a patch that cuts the hub down to 40 links goes back to the
    row scans (mode 0)."""
    g = hub_lsdb(1)
    hub = g.meta["hub"]
    assert int(np.diff(g.row_ptr).max()) >= 2500
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert int(G.export("build_mode")[0]) == 1
        assert_layout(G, g)
        roots = np.array([0, 1, 2, hub + 1, g.n - 1] + list(range(50, 109)), np.uint32)
        roots = roots[roots != hub]
        check_spf(spf_ctx, G, g, roots)
        a = int(g.row_ptr[hub])
        G.patch([hub], [(g.col[a:a + 40].copy(), g.metric[a:a + 40].copy())], [0])
        g2 = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
        assert int(G.export("build_mode")[0]) == 0
        assert_layout(G, g2)
        check_spf(spf_ctx, G, g2, roots)
    finally:
        G.free()


def test_parallel_links_piled_onto_one_row_rebuild_in_hub_mode(spf_ctx):
    """No row lists more than 512 links, but 30 routers x 30 parallel links land on one vertex (900 in-links): the plain
    pass reports the in-degree, the build runs again from sorted keys, and the layout is the restated one."""
    g = hub_lsdb(2, n_leaves=600, hub_links=300, parallel=30)
    assert int(np.diff(g.row_ptr).max()) <= 512
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert int(np.diff(G.export("in_ptr")).max()) > 512
        assert int(G.export("build_mode")[0]) == 1
        assert_layout(G, g)
        check_spf(spf_ctx, G, g, np.arange(0, 64, dtype=np.uint32))
    finally:
        G.free()


def test_star_of_100k_links(spf_ctx):
    """One router with 100 000 neighbours: upload, two-way flags and in-row order against numpy, SPF from 16 leaves
    against the heap oracle."""
    g = hub_lsdb(3, n_leaves=100_000, hub_links=100_000)
    hub = g.meta["hub"]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert int(G.export("build_mode")[0]) == 1
        n = g.n
        src = np.repeat(np.arange(n, dtype=np.int64), np.diff(g.row_ptr.astype(np.int64)))
        fwd = src * n + g.col.astype(np.int64)
        two = np.isin(g.col.astype(np.int64) * n + src, fwd)
        assert np.array_equal(G.export("twoway"), two.astype(np.uint8))
        keep = two                                                # no HSPF_VF_NO_EXPAND in this graph
        ks, kt, kw = src[keep], g.col[keep].astype(np.int64), g.metric[keep].astype(np.int64)
        kp = (np.arange(len(src), dtype=np.int64) - g.row_ptr.astype(np.int64)[src])[keep]
        order = np.lexsort((kp, ks, -kw, kt))
        assert np.array_equal(G.export("in_src") & 0x3FFFFFFF, ks[order].astype(np.uint32))
        assert np.array_equal(G.export("in_cost"), kw[order].astype(np.uint32))
        assert np.array_equal(G.export("in_pos"), kp[order].astype(np.uint32))
        assert np.array_equal(G.export("out_dst"), kt.astype(np.uint32))
        roots = (np.arange(16, dtype=np.uint64) * (n - 1) // 16).astype(np.uint32)
        roots = roots[roots != hub]
        res = spf_ctx.run(G, roots)
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP,
                     mask_words_=res.first_hop_mask.shape[2])
        assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops)
        assert np.array_equal(res.first_hop_mask, ref.mask)
    finally:
        G.free()


def test_graph_without_links_and_single_vertex(spf_ctx):
    for n in (1, 5):
        g = synth.CsrGraph(np.zeros(n + 1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(n, np.uint8))
        G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        assert_layout(G, g)
        res = spf_ctx.run(G, [0])
        assert res.dist[0, 0] == 0 and (res.dist[0, 1:] == E.DIST_INF).all()
        G.free()


def test_upload_rejects_bad_links_with_a_code(spf_ctx):
    g = synth.ospf_500()
    col = g.col.copy(); col[123] = g.n
    with pytest.raises(E.HspfError) as ei:
        spf_ctx.upload(g.row_ptr, col, g.metric, g.vflags, g.max_path_metric)
    assert ei.value.code == -1 and "col out of range" in str(ei.value)
    met = g.metric.copy(); met[7] = 0xFFFFFFFF
    with pytest.raises(E.HspfError) as ei:
        spf_ctx.upload(g.row_ptr, g.col, met, g.vflags, g.max_path_metric)
    assert ei.value.code == -1 and "reserved" in str(ei.value)


def random_rows(g, rng, k, grow=0):
    """k replacement rows: links dropped, re-costed, re-ordered, added (also one-way ones), flags flipped."""
    n = g.n
    vs = np.sort(rng.choice(n, size=min(k, n), replace=False))
    rows, flags = [], []
    nn = g.meta.get("n_networks", 0)
    for v in vs.tolist():
        c = g.col[g.row_ptr[v]:g.row_ptr[v + 1]].copy()
        m = g.metric[g.row_ptr[v]:g.row_ptr[v + 1]].copy()
        keep = rng.random(len(c)) > 0.3
        c, m = c[keep], m[keep]
        if len(m) and v >= nn:
            m = np.where(rng.random(len(m)) < 0.5, rng.integers(1, 7, len(m)), m).astype(np.uint32)
        extra = rng.integers(0, 4) + grow
        if v >= nn and extra:
            ec = rng.integers(nn, n, extra).astype(np.uint32)          # router -> router, mostly one-way
            c = np.concatenate([c, ec]); m = np.concatenate([m, rng.integers(1, 7, extra).astype(np.uint32)])
        p = rng.permutation(len(c))
        rows.append((c[p], m[p]))
        f = int(g.vflags[v])
        if v >= nn and rng.random() < 0.3:
            f ^= synth.VF_NO_TRANSIT
        if rng.random() < 0.2:
            f ^= synth.VF_NO_EXPAND
        flags.append(f)
    return vs, rows, np.array(flags, np.uint8)


def check_spf(ctx, G, g, roots, run_flags=0):
    res = ctx.run(G, roots, run_flags)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, np.asarray(roots, np.uint32), run_flags & 3,
                 go.MAP, mask_words_=res.first_hop_mask.shape[2])
    assert np.array_equal(res.dist, ref.dist)
    assert np.array_equal(res.hops, ref.hops)
    assert np.array_equal(res.flags & 1, ref.flags)
    assert np.array_equal(res.first_hop_mask, ref.mask)


@hub_engines
@pytest.mark.parametrize("seed", range(6))
def test_patch_equals_fresh_upload(spf_ctx, seed):
    rng = np.random.default_rng(seed)
    g = synth.random_lsdb(70, 9, 3.0, 700 + seed, metric_hi=6)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.arange(9, 9 + 40, dtype=np.uint32)
    try:
        for rnd in range(4):
            vs, rows, flags = random_rows(g, rng, [1, 3, 10, 79][rnd])
            G.patch(vs, rows, flags)
            g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
            F = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            try:
                for name in BUILT + RAW + DERIVED:
                    assert np.array_equal(G.export(name), F.export(name)), (rnd, name)
                assert G.n_edges_kept == F.n_edges_kept
            finally:
                F.free()
            assert_layout(G, g)
            check_spf(spf_ctx, G, g, roots, E.RUN_NET_NEXTHOPS)
    finally:
        G.free()


def patch_graphs():
    yield synth.random_lsdb(60, 8, 3.0, 21, metric_hi=6)
    yield synth.random_lsdb(90, 10, 2.5, 22, metric_hi=3, p_oneway=0.3, p_parallel=0.4, p_noexpand=0.2, p_overload=0.3)
    yield synth.random_lsdb(50, 8, 2.5, 23, hopcount=True)
    yield synth.random_lsdb(120, 6, 3.0, 24, metric_hi=2, zero_cost_router_links=True)
    yield synth.random_lsdb(700, 9, 3.0, 25, lan_size=90)                # heavy chunks (work units), rows of 90 links
    yield synth.random_lsdb(300, 40, 1.2, 26, lan_size=2, metric_hi=4)   # sparse: leaves, stub LANs, isolated vertices
    yield synth.random_lsdb(3000, 100, 4.0, 27, metric_hi=2, zero_cost_router_links=True)


@pytest.mark.parametrize("i", range(7))
def test_incremental_structural_patches_equal_fresh_uploads(spf_ctx, i):
    """Round 6 ("structural patch in O(changed rows)"): a patch that changes a row's targets re-derives only
    the AFFECTED rows (the replaced ones, their old and new targets) and shifts the compact arrays behind them
    (holo_amd/csrc/graph_patch.hip.h) instead of rebuilding the layout — a chain of such patches, one to three rows each
    (links withdrawn, announced, re-costed, re-ordered, one-way links, flags flipped, rows emptied and refilled), every
    exported array and the summary equal to a fresh upload's after every step, and the SPTs of the patched graph equal
    to the oracle's."""
    g = list(patch_graphs())[i]
    rng = np.random.default_rng(900 + i)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    nn = g.meta.get("n_networks", 0)
    roots = np.arange(nn, min(nn + 24, g.n), dtype=np.uint32)
    modes = []
    try:
        for rnd in range(14):
            if rnd == 5:                                               # a row emptied ...
                v = int(rng.integers(nn, g.n))
                saved = (v, g.col[g.row_ptr[v]:g.row_ptr[v + 1]].copy(), g.metric[g.row_ptr[v]:g.row_ptr[v + 1]].copy(), int(g.vflags[v]))
                vs, rows, flags = np.array([v]), [(np.zeros(0, np.uint32), np.zeros(0, np.uint32))], np.array([g.vflags[v]], np.uint8)
            elif rnd == 7:                                             # ... and back two patches later
                vs, rows, flags = np.array([saved[0]]), [(saved[1], saved[2])], np.array([saved[3]], np.uint8)
            else:
                vs, rows, flags = random_rows(g, rng, int(rng.integers(1, 4)))
            if rnd % 3 == 1:                                           # new costs on a few rows first, in place (the per-row facts the next summary reads must follow)
                vs2 = np.sort(rng.choice(g.n, size=min(g.n, 6), replace=False))
                rows2 = [(g.col[g.row_ptr[v]:g.row_ptr[v + 1]].copy(), rng.integers(0, 4, int(g.row_ptr[v + 1] - g.row_ptr[v])).astype(np.uint32)) for v in vs2.tolist()]
                G.patch(vs2, rows2, g.vflags[vs2].copy())
                g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
            G.patch(vs, rows, flags)
            modes.append(int(G.export("build_mode")[0]))
            g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
            F = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            try:
                for name in BUILT + RAW + DERIVED + ("zcyc",):
                    assert np.array_equal(G.export(name), F.export(name)), (rnd, name, modes)
                assert G.n_edges_kept == F.n_edges_kept
            finally:
                F.free()
            if rnd % 4 == 3:
                assert_layout(G, g)
                check_spf(spf_ctx, G, g, roots, E.RUN_NET_NEXTHOPS)
        assert modes.count(3) >= 8, modes                             # (2: a draw that changed costs only)
    finally:
        G.free()


@hub_engines
def test_structural_patch_keeps_the_host_side_current(spf_ctx):
    """A structural patch enqueues the device build and brings the host side up to date behind it: the mirror of the rows,
    their two-way flags (from the replaced rows and their old and new targets alone) and the summary of the caller's rows
    (longest row, network vertices, links in long rows) — all as a fresh upload has them, for the changes that exercise each
    rule: the longest row shrinks, rows that list each other are replaced together, a self-link, an emptied row, a vertex
    changing its kind, everything returning."""
    g0 = synth.random_lsdb(300, 6, 3.0, 4242, lan_size=60, metric_hi=5)
    g = g0
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    nn = 6
    roots = np.arange(nn, nn + 40, dtype=np.uint32)

    def row(v):
        return g.col[g.row_ptr[v]:g.row_ptr[v + 1]].copy(), g.metric[g.row_ptr[v]:g.row_ptr[v + 1]].copy()

    def step(tag, vs, rows, flags, spf=True):
        nonlocal g
        G.patch(vs, rows, flags)
        g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
        F = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        try:
            for name in BUILT + RAW + DERIVED:
                assert np.array_equal(G.export(name), F.export(name)), (tag, name)
        finally:
            F.free()
        assert_layout(G, g)
        if spf:
            check_spf(spf_ctx, G, g, roots, E.RUN_NET_NEXTHOPS)

    try:
        lens = np.diff(g.row_ptr.astype(np.int64))
        big = int(np.argmax(lens))
        assert lens[big] > 32 and int(G.export("summary")[8]) == lens[big]
        c, m = row(big)
        step("longest row shrinks", [big], [(c[:5], m[:5])], [g.vflags[big]])
        assert int(G.export("summary")[8]) == int(np.diff(g.row_ptr.astype(np.int64)).max())
        # two routers that list each other: one drops the other and lists itself instead, the other reverses its row
        u = next(v for v in range(nn, g.n) if any(t >= nn and v in row(int(t))[0] for t in row(v)[0]))
        t = int(next(t for t in row(u)[0] if t >= nn and u in row(int(t))[0]))
        cu, mu = row(u); ct, mt = row(t)
        cu2 = np.where(cu == t, u, cu).astype(np.uint32)
        vs = sorted([u, t])
        rows = {u: (cu2, mu), t: (ct[::-1].copy(), mt[::-1].copy())}
        step("partners replaced together, self-link", vs, [rows[v] for v in vs], [g.vflags[v] for v in vs])
        step("row emptied", [u], [(np.zeros(0, np.uint32), np.zeros(0, np.uint32))], [g.vflags[u]])
        # a network vertex becomes a router vertex and a router a network vertex (odd LSDBs are still graphs): layouts only
        step("kinds change", [0, u], [row(0), (cu, mu)], [g.vflags[0] ^ synth.VF_NETWORK, g.vflags[u] ^ synth.VF_NETWORK], spf=False)
        assert int(G.export("summary")[9]) == int((g.vflags & synth.VF_NETWORK != 0).sum())
        # everything returns
        vs = sorted({0, u, t, big})
        step("all back", vs, [(g0.col[g0.row_ptr[v]:g0.row_ptr[v + 1]], g0.metric[g0.row_ptr[v]:g0.row_ptr[v + 1]]) for v in vs], [g0.vflags[v] for v in vs])
        assert np.array_equal(G.col, g0.col) and np.array_equal(G.row_ptr, g0.row_ptr)
    finally:
        G.free()


def test_structural_patch_of_a_hub_row_fetches_the_two_way_flags(spf_ctx):
    """Replacing a row with thousands of links (a LAN's pseudonode losing half of its members, then getting them back) is
    beyond what the patch scans on the host to keep its two-way mirror (HSPF_TW_HOST_MAX: max(4096, links / 32) row entries):
    the flags come back from the device, as after an upload — same arrays as a fresh upload either way."""
    g = synth.random_lsdb(3000, 1, 2.0, 515, lan_size=2500, metric_hi=4)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        lan = 0
        a, b = int(g.row_ptr[lan]), int(g.row_ptr[lan + 1])
        assert b - a >= 2000
        full = (g.col[a:b].copy(), g.metric[a:b].copy())
        for tag, row in (("half", (full[0][::2].copy(), full[1][::2].copy())), ("back", full)):
            G.patch([lan], [row], [g.vflags[lan]])
            F = spf_ctx.upload(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
            try:
                for name in BUILT + RAW + DERIVED + ("host_row_ptr", "host_col"):
                    if name == "units":
                        continue
                    assert np.array_equal(G.export(name), F.export(name)), (tag, name)
            finally:
                F.free()
        g2 = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
        members = set(full[0].tolist())
        roots = np.array([v for v in range(1, g.n) if v not in members][:40], np.uint32)     # (a member has 2 500 first-hop slots: over the limit)
        assert len(roots) == 40
        check_spf(spf_ctx, G, g2, roots, E.RUN_NET_NEXTHOPS)
    finally:
        G.free()


def cost_rows(g, rng, k, lo, hi):
    """k rows with the same targets, order and flags and new costs in [lo, hi] (links into a network keep theirs
    with probability 1/2, so ties and zero costs stay around)."""
    vs = np.sort(rng.choice(g.n, size=min(k, g.n), replace=False))
    rows = []
    for v in vs.tolist():
        c = g.col[g.row_ptr[v]:g.row_ptr[v + 1]].copy()
        m = g.metric[g.row_ptr[v]:g.row_ptr[v + 1]].copy()
        m = np.where(rng.random(len(m)) < 0.6, rng.integers(lo, hi + 1, len(m)), m).astype(np.uint32)
        rows.append((c, m))
    return vs, rows, g.vflags[vs].copy()


def cost_graphs():
    yield synth.random_lsdb(70, 9, 3.0, 900, metric_hi=6), 0, 6
    yield synth.random_lsdb(90, 10, 2.5, 901, metric_hi=3, p_oneway=0.3, p_parallel=0.4, p_noexpand=0.2, p_overload=0.3), 0, 3
    yield synth.random_lsdb(300, 20, 4.0, 902, metric_hi=2, zero_cost_router_links=True), 0, 2
    yield synth.random_lsdb(50, 8, 2.5, 903, hopcount=True), 0, 1
    yield synth.random_lsdb(40, 1, 3.0, 904, lan_size=40), 1, 9          # rows of more than 16 links
    yield synth.random_lsdb(200, 0, 4.0, 905, metric_hi=60), 1, 1000      # routers only: the largest cost moves up and down


@pytest.mark.parametrize("i", range(6))
def test_cost_only_patch_equals_fresh_upload(spf_ctx, i):
    """Rows replaced with the same targets and new costs take the in-place path (build mode 2: nothing rebuilt, no
    per-link array uploaded) and leave every device array and every derived value as a fresh upload of the patched
    CSR has them — also when the change moves the hop-count shape, the RF_ZERO rows or the largest cost."""
    g, lo, hi = list(cost_graphs())[i]
    rng = np.random.default_rng(40 + i)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    nn = g.meta.get("n_networks", 0)
    roots = np.arange(nn, nn + min(40, g.n - nn), dtype=np.uint32)
    try:
        for rnd in range(6):
            if rnd == 4:                                      # 0 into a network, 1 into a router: the largest cost shrinks, hop-count shape returns
                vs = np.arange(g.n)
                rows = [(g.col[g.row_ptr[v]:g.row_ptr[v + 1]],
                         (g.col[g.row_ptr[v]:g.row_ptr[v + 1]] >= nn).astype(np.uint32)) for v in vs]
                flags = g.vflags.copy()
            else:
                vs, rows, flags = cost_rows(g, rng, [1, 3, 10, g.n, 1, 2][rnd], lo, hi)
            G.patch(vs, rows, flags)
            assert int(G.export("build_mode")[0]) == 2, rnd
            g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
            F = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            try:
                for name in BUILT + RAW + DERIVED:
                    assert np.array_equal(G.export(name), F.export(name)), (rnd, name)
            finally:
                F.free()
            assert_layout(G, g)
            check_spf(spf_ctx, G, g, roots, E.RUN_NET_NEXTHOPS)
    finally:
        G.free()


def test_cost_only_patch_full_size(spf_ctx):
    """isis-100k: one router's costs change (the headline's incremental update); arrays as a fresh upload has them,
    results equal, and the lean sweep still takes the run."""
    g = synth.isis_100k()
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = (np.arange(64, dtype=np.uint64) * g.n // 64).astype(np.uint32)
    rng = np.random.default_rng(3)
    try:
        for u in (5000, 0, g.n - 1):
            a, b = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
            G.patch([u], [(g.col[a:b], rng.integers(1, 101, b - a).astype(np.uint32))], [g.vflags[u]])
            assert int(G.export("build_mode")[0]) == 2
        F = spf_ctx.upload(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
        try:
            for name in BUILT + RAW + DERIVED:
                assert np.array_equal(G.export(name), F.export(name)), name
            r1, r2 = spf_ctx.run(G, roots), spf_ctx.run(F, roots)
            for f in ("dist", "hops", "flags", "first_hop_mask"):
                assert np.array_equal(getattr(r1, f), getattr(r2, f)), f
        finally:
            F.free()
    finally:
        G.free()


def test_patch_grows_past_the_spare_capacity(spf_ctx):
    rng = np.random.default_rng(11)
    g = synth.random_lsdb(200, 10, 3.0, 811, metric_hi=5)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        for _ in range(3):                                   # +~4000 links per round, spare capacity is 1024
            vs, rows, flags = random_rows(g, rng, 40, grow=100)
            G.patch(vs, rows, flags)
            g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
            assert_layout(G, g)
        check_spf(spf_ctx, G, g, np.arange(10, 74, dtype=np.uint32))
    finally:
        G.free()


def test_many_growing_patches_keep_the_host_row_pool_right(spf_ctx):
    """The host mirror of the caller's rows is a pool since round 6: a replaced row that grows is appended, its old place is
    dead, and the pool is packed again when half of it is (hspf_graph::mirror_compact).  Sixteen rounds of a hundred rows
    replaced by longer and shorter ones push it through that several times; the mirrors (host_row_ptr / host_col / twoway)
    and the layout stay those of a fresh upload, and the slot tables made from the mirror still give the oracle's masks."""
    rng = np.random.default_rng(12)
    g = synth.random_lsdb(400, 10, 3.0, 812, metric_hi=5)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        for rnd in range(16):
            vs, rows, flags = random_rows(g, rng, 100, grow=[180, 0, 60, 0][rnd % 4])
            G.patch(vs, rows, flags)
            g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
            if rnd % 3 == 2 or rnd == 15:
                assert_layout(G, g)
        check_spf(spf_ctx, G, g, np.arange(10, 74, dtype=np.uint32), E.RUN_NET_NEXTHOPS)
    finally:
        G.free()


def test_patch_hopcount_shape_follows_the_links(spf_ctx):
    """A hop-count graph (every root on the fused path) stops being one when a row gets real metrics, and is one
    again when the row is restored; results match the oracle in all three states."""
    g0 = synth.random_lsdb(50, 8, 2.5, 301, hopcount=True)
    G = spf_ctx.upload(g0.row_ptr, g0.col, g0.metric, g0.vflags, g0.max_path_metric)
    roots = np.arange(8, 8 + 20, dtype=np.uint32)
    try:
        check_spf(spf_ctx, G, g0, roots, E.RUN_IGNORE_OVERLOAD)
        exact0 = spf_ctx.stats()["n_exact_roots"]
        v = 20
        c = g0.col[g0.row_ptr[v]:g0.row_ptr[v + 1]]; m = g0.metric[g0.row_ptr[v]:g0.row_ptr[v + 1]]
        G.patch([v], [(c, m + 3)], [g0.vflags[v]])
        g1 = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g0.max_path_metric)
        check_spf(spf_ctx, G, g1, roots, E.RUN_IGNORE_OVERLOAD)
        G.patch([v], [(c, m)], [g0.vflags[v]])
        assert np.array_equal(G.col, g0.col) and np.array_equal(G.metric, g0.metric)
        check_spf(spf_ctx, G, g0, roots, E.RUN_IGNORE_OVERLOAD)
        assert spf_ctx.stats()["n_exact_roots"] == exact0
    finally:
        G.free()


def test_invalid_patch_changes_nothing(spf_ctx):
    g = synth.random_lsdb(60, 8, 3.0, 5, metric_hi=6)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        before = {name: G.export(name) for name in BUILT + RAW}
        bad = [
            ([10, 10], [([9], [1]), ([9], [1])], [0, 0]),            # not strictly ascending
            ([g.n], [([9], [1])], [0]),                              # vertex out of range
            ([10], [([g.n], [1])], [0]),                             # target out of range
            ([10], [([9], [0xFFFFFFFF])], [0]),                      # reserved cost
        ]
        for vs, rows, fl in bad:
            with pytest.raises(E.HspfError) as ei:
                G.patch(vs, rows, fl)
            assert ei.value.code == -1
        for name in BUILT + RAW:
            assert np.array_equal(G.export(name), before[name]), name
        assert np.array_equal(G.col, g.col)
    finally:
        G.free()


def test_patch_full_size_router_purge_and_return(spf_ctx):
    """isis-100k: one router's LSP is purged (its row becomes empty: every link to it now fails the two-way check on
    the unchanged rows of its neighbours) and re-originated; the patched graph gives the results of a fresh upload,
    and the return restores the original results bit for bit."""
    g = synth.isis_100k()
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = (np.arange(64, dtype=np.uint64) * g.n // 64).astype(np.uint32)
    try:
        base = spf_ctx.run(G, roots)
        Gexp0 = {name: G.export(name) for name in BUILT + RAW + DERIVED}
        u = 5000
        a, b = int(g.row_ptr[u]), int(g.row_ptr[u + 1])
        row_u = (g.col[a:b].copy(), g.metric[a:b].copy())
        kept0 = G.n_edges_kept
        G.patch([u], [(np.zeros(0, np.uint32), np.zeros(0, np.uint32))], [g.vflags[u]])
        assert G.n_edges_kept == kept0 - 2 * (b - a)
        assert int(G.export("build_mode")[0]) == (3 if getattr(spf_ctx, "mode", "default") == "default" else 0)
        F = spf_ctx.upload(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
        for name in BUILT + RAW + DERIVED:
            assert np.array_equal(G.export(name), F.export(name)), name
        r1, r2 = spf_ctx.run(G, roots), spf_ctx.run(F, roots)
        F.free()
        for f in ("dist", "hops", "flags", "first_hop_mask"):
            assert np.array_equal(getattr(r1, f), getattr(r2, f)), f
        assert (r1.dist[:, u] == E.DIST_INF).all() and (base.dist[:, u] != E.DIST_INF).all()
        G.patch([u], [row_u], [g.vflags[u]])
        assert G.n_edges_kept == kept0
        for name in BUILT + RAW + DERIVED:
            assert np.array_equal(G.export(name), Gexp0[name]), name
        r3 = spf_ctx.run(G, roots)
        for f in ("dist", "hops", "flags", "first_hop_mask"):
            assert np.array_equal(getattr(r3, f), getattr(base, f)), f
    finally:
        G.free()
