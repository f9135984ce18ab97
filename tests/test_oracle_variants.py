"""The three variants of the CPU oracle (oracle/spf_oracle.cpp: REF = the reference's shape, linear candidate scan; MAP =
the same ordered map behind an index; HEAP = binary heap) must agree field by field on adversarial LSDBs — in particular
where path costs SATURATE at u32::MAX (OSPF's saturating add with max path metric 0xFFFFFFFF): a candidate whose distance
is 0xFFFFFFFF is still a candidate.  (Found by the GPU fuzz, graph 54989: MAP used 0xFFFFFFFF as "not on the list", kept a
saturated candidate beside its later, shorter replacement and reported the saturated one; the engine and HEAP had the
reference's answer.)"""
import numpy as np
import pytest

from holo_amd import synth
from oracle import graph_oracle as go

FIELDS = ("dist", "hops", "flags", "pop_rank", "mask", "n_nexthops", "n_parents")


def lsdb(seed):
    rng = np.random.default_rng(seed)
    hi = int(rng.integers(2, 40))
    g = synth.random_lsdb(int(rng.integers(20, 200)), int(rng.integers(0, 8)), float(rng.uniform(1.2, 2.2)), 31_000 + seed,
                          metric_lo=1, metric_hi=hi, max_path=0xFFFFFFFF,
                          p_oneway=0.03, p_parallel=float(rng.choice([0.0, 0.2])), p_overload=0.03, p_noexpand=0.02,
                          lan_size=int(rng.choice([3, 8, 20])))
    g.metric = (g.metric.astype(np.uint64) << (31 - hi.bit_length() - int(rng.integers(0, 3)))).clip(0, 0xFFFFFFFE).astype(np.uint32)
    return g


def agree(g, roots, flags=0):
    W = 4
    res = {v: go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, flags, v, mask_words_=W)
           for v in (go.REF, go.MAP, go.HEAP)}
    for f in FIELDS:
        assert np.array_equal(getattr(res[go.REF], f), getattr(res[go.MAP], f)), f
        assert np.array_equal(getattr(res[go.REF], f), getattr(res[go.HEAP], f)), f
    return res[go.REF]


@pytest.mark.parametrize("seed", range(24))
def test_variants_agree_where_costs_saturate(seed):
    g = lsdb(seed)
    roots = np.arange(0, g.n, max(1, g.n // 24), dtype=np.uint32)
    agree(g, roots, int(seed % 4))


def test_the_saturating_case_is_exercised():
    hit = 0
    for seed in range(24):
        g = lsdb(seed)
        roots = np.arange(0, g.n, max(1, g.n // 24), dtype=np.uint32)
        r = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=4)
        hit += bool(((r.dist == 0xFFFFFFFF) & (r.flags == 1)).any())
    assert hit >= 8, hit                                 # graphs with vertices IN the SPT at distance u32::MAX


def test_saturated_candidate_is_replaced_by_a_shorter_one():
    """0 -> 1 -> 2 saturates on the way to 2 (pop order reaches it first through the cheap first hop); 0 -> 3 -> 2 is a
    real 0xF0000000.  The reference replaces the saturated candidate (holo-ospf/src/spf.rs:682-700: found by id)."""
    big = 0xF0000000
    links = [(0, 1, 2), (1, 0, 1), (1, 2, 0xFFFFFFFE), (2, 1, 5), (0, 3, big - 1), (3, 0, 7), (3, 2, 1), (2, 3, 9)]
    src = np.array([a for a, _, _ in links]); dst = np.array([b for _, b, _ in links]); met = np.array([c for _, _, c in links])
    row_ptr, col, metric = synth._csr_from_links(4, src, dst, met)
    g = synth.CsrGraph(row_ptr, col, metric, np.zeros(4, np.uint8), 0xFFFFFFFF, "sat", {})
    r = agree(g, np.array([0], np.uint32))
    assert r.dist[0].tolist() == [0, 2, big, big - 1]
    assert r.hops[0].tolist() == [0, 1, 2, 1]
