"""hspf_graph_upload_keyed (LSDB -> CSR on the device): vertices by key in ANY order, links as (target key,
cost), targets unresolved, some of them absent — the graph the device builds equals, array by array, the one hspf_graph_upload
builds from the CSR the host twin derives (vertex index = rank in VertexId order, links to absent vertices dropped), and runs
on it give the oracle's answers."""
import glob
import json
import os

import numpy as np
import pytest

from holo_amd import engine as E
from holo_amd import isis as H
from holo_amd import synth
from oracle import graph_oracle as go

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ARRAYS = ("row_ptr", "col", "metric", "vflags", "in_ptr", "in_src", "in_cost", "in_pos", "out_ptr", "out_dst", "out_cost", "out_pos", "rowflags", "leaf", "twoway", "summary")


def _keyed(g, seed, n_absent=0):
    """The CSR `g` as keyed records: random distinct keys in index order, vertices shuffled, plus links to `n_absent` keys no
    vertex has (spread over random rows, at random positions)."""
    rng = np.random.default_rng(seed)
    keys = np.unique(rng.integers(1, 1 << 40, size=2 * (g.n + n_absent) + 64, dtype=np.uint64))       # (never an arange of the key SPACE)
    keys = np.sort(rng.choice(keys, size=g.n + n_absent, replace=False))
    absent = rng.choice(len(keys), size=n_absent, replace=False)
    present = np.setdiff1d(np.arange(len(keys)), absent)
    vkey = keys[present]                                            # vkey[i] = key of vertex i (ascending: index = rank)
    order = rng.permutation(g.n)
    rows, mets = [], []
    extra = {int(v): [] for v in rng.integers(0, g.n, size=n_absent)} if n_absent else {}
    for a, v in zip(absent.tolist(), list(extra) * (n_absent // max(len(extra), 1) + 1)):
        extra[v].append(int(keys[a]))
    for v in order.tolist():
        tk = vkey[g.col[g.row_ptr[v]:g.row_ptr[v + 1]]].tolist()
        mt = g.metric[g.row_ptr[v]:g.row_ptr[v + 1]].tolist()
        for k in extra.get(v, []):
            p = int(rng.integers(0, len(tk) + 1))
            tk.insert(p, k); mt.insert(p, 7)
        rows.append(tk); mets.append(mt)
    rp = np.zeros(g.n + 1, np.uint32)
    rp[1:] = np.cumsum([len(r) for r in rows])
    return (vkey[order], rp, np.asarray([k for r in rows for k in r], np.uint64), np.asarray([m for r in mets for m in r], np.uint32), g.vflags[order], order)


def _check(ctx, g, seed, n_absent, roots):
    vk, rp, tk, mt, vf, order = _keyed(g, seed, n_absent)
    K, rank = E.SpfGraph.from_keys(ctx, vk, rp, tk, mt, vf, g.max_path_metric)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        assert np.array_equal(rank, order), "rank of input vertex i"
        for name in ARRAYS:
            assert np.array_equal(K.export(name), G.export(name)), name
        res = ctx.run(K, roots, 0)
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.MAP, mask_words_=res.first_hop_mask.shape[2])
        assert np.array_equal(res.dist, ref.dist) and np.array_equal(res.hops, ref.hops) and np.array_equal(res.first_hop_mask, ref.mask)
    finally:
        K.free(); G.free()


@pytest.mark.parametrize("seed", range(6))
def test_random_lsdbs_shuffled_with_absent_targets(spf_ctx, seed):
    g = synth.random_lsdb(150 + 40 * seed, 10, 3.0, 8100 + seed, metric_hi=9)
    _check(spf_ctx, g, seed, 25, np.arange(10, 40, dtype=np.uint32))


def test_isis_100k_shuffled(spf_ctx):
    g = synth.isis_100k()
    _check(spf_ctx, g, 5, 1000, np.asarray([0, 50000, 99999], np.uint32))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "isis", "*.json")))[:12], ids=lambda p: os.path.basename(p)[:-5])
def test_reference_fixtures_by_lan_id_keys(spf_ctx, path):
    """The recorded IS-IS LSDBs: keys = !pseudonode << 56 | LAN id (holo-isis/src/spf.rs:96-100), vertices in LSDB order."""
    inst = H.Instance.from_vector(json.load(open(path)))
    for level in inst.config.levels():
        g = H.LevelGraph(inst, level, 0, False)
        if g.n == 0:
            continue
        key = lambda vid: (int(bool(vid[0])) << 56) | int.from_bytes(bytes(vid[1]) + bytes([vid[2]]), "big")     # noqa: E731
        vkey = np.asarray([key(v) for v in g.vids], np.uint64)
        assert (np.diff(vkey.astype(object)) > 0).all(), "key order = VertexId order"
        order = np.random.default_rng(3).permutation(g.n)
        rows = [vkey[g.col[g.row_ptr[v]:g.row_ptr[v + 1]]] for v in order]
        rp = np.zeros(g.n + 1, np.uint32); rp[1:] = np.cumsum([len(r) for r in rows])
        K, rank = E.SpfGraph.from_keys(spf_ctx, vkey[order], rp, np.concatenate(rows) if len(rows) else np.zeros(0, np.uint64),
                                       np.concatenate([g.metric[g.row_ptr[v]:g.row_ptr[v + 1]] for v in order]), g.vflags[order], g.max_path_metric)
        G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        try:
            assert np.array_equal(rank, order)
            for name in ARRAYS:
                assert np.array_equal(K.export(name), G.export(name)), name
        finally:
            K.free(); G.free()


def test_duplicate_keys_are_refused(spf_ctx):
    with pytest.raises(E.HspfError) as ei:
        E.SpfGraph.from_keys(spf_ctx, [5, 9, 5], [0, 1, 2, 2], [9, 5], [1, 1], [0, 0, 0], 0xFE000000)
    assert ei.value.code == -1
