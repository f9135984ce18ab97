"""splice_rows (the numpy twin of hspf_graph_patch's row replacement) on CPU."""
import numpy as np

from holo_amd import synth
from holo_amd.engine import splice_rows


def test_splice_rows_replaces_whole_rows():
    g = synth.random_lsdb(30, 4, 3.0, 9)
    rng = np.random.default_rng(0)
    vs = np.sort(rng.choice(g.n, 7, replace=False))
    cols = [rng.integers(0, g.n, rng.integers(0, 6)).astype(np.uint32) for _ in vs]
    mets = [rng.integers(1, 9, len(c)).astype(np.uint32) for c in cols]
    nf = rng.integers(0, 8, len(vs)).astype(np.uint8)
    rp, col, met, vf = splice_rows(g.row_ptr, g.col, g.metric, g.vflags, vs, cols, mets, nf)
    assert rp[0] == 0 and rp[-1] == len(col) == len(met)
    j = 0
    for u in range(g.n):
        a, b = rp[u], rp[u + 1]
        if j < len(vs) and vs[j] == u:
            assert np.array_equal(col[a:b], cols[j]) and np.array_equal(met[a:b], mets[j]) and vf[u] == nf[j]
            j += 1
        else:
            oa, ob = g.row_ptr[u], g.row_ptr[u + 1]
            assert np.array_equal(col[a:b], g.col[oa:ob]) and np.array_equal(met[a:b], g.metric[oa:ob])
            assert vf[u] == g.vflags[u]


def test_splice_rows_first_and_last_vertex_and_empty_rows():
    rp = np.array([0, 2, 2, 5, 6], np.uint32); col = np.arange(6, dtype=np.uint32); met = col + 10
    vf = np.zeros(4, np.uint8)
    nrp, ncol, nmet, nvf = splice_rows(rp, col, met, vf, [0, 3], [np.zeros(0, np.uint32), np.array([1, 2], np.uint32)],
                                       [np.zeros(0, np.uint32), np.array([7, 8], np.uint32)], [4, 2])
    assert nrp.tolist() == [0, 0, 0, 3, 5] and ncol.tolist() == [2, 3, 4, 1, 2] and nmet.tolist() == [12, 13, 14, 7, 8]
    assert nvf.tolist() == [4, 0, 0, 2]
