"""The closed form of the reference's pop order on graphs with zero-cost links (holo_amd/csrc/spf_repair.hip.h, restated in
tests/_dynamic_order_model.py) against the oracle's literal loop: pop_rank, hops and first-hop masks, every root, on small
tie-heavy LSDBs where a third to a half of the router links cost 0 (nested groups, zero-cost cycles, LANs, overloaded and
non-expandable vertices, parallel links, IS-IS and OSPF next-hop rules)."""
import numpy as np
import pytest

from holo_amd import synth
from oracle import graph_oracle as go
import _dynamic_order_model as M


def _check(g, roots, run_flags):
    W = max(1, go.mask_words(g.row_ptr, g.col, g.metric, g.vflags, roots))
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, run_flags, go.MAP, mask_words_=W)
    dyn = 0
    for i, r in enumerate(roots):
        out = M.dynamic_order(g.row_ptr, g.col, g.metric, g.vflags, int(r), ref.dist[i], ignore_ovl=bool(run_flags & go.RUN_IGNORE_OVERLOAD),
                              net_nexthops=bool(run_flags & go.RUN_NET_NEXTHOPS), words=W)
        R, pos, hops, mask, rank = out
        assert np.array_equal(rank, ref.pop_rank[i]), (g.name, int(r))
        assert np.array_equal(hops, ref.hops[i]), (g.name, int(r))
        assert np.array_equal(mask, ref.mask[i]), (g.name, int(r))
        dyn += any(R[v] != v for v in range(g.n))
    return dyn


@pytest.mark.parametrize("seed", range(24))
def test_closed_form_order_equals_the_literal_loop(seed):
    hi = 1 + seed % 3                                          # costs 0..1, 0..2, 0..3: half / a third / a quarter of the links cost 0
    g = synth.random_lsdb(60 + 7 * (seed % 5), 6 if seed % 2 else 0, 2.6 + 0.2 * (seed % 4), 9000 + seed, metric_hi=hi, zero_cost_router_links=True)
    roots = np.arange(g.n, dtype=np.uint32)
    flags = (0, go.RUN_NET_NEXTHOPS, go.RUN_IGNORE_OVERLOAD)[seed % 3]
    dyn = _check(g, roots, flags)
    assert dyn > 0                                             # roots with a non-static order did occur


def test_a_group_that_outgrows_the_walk_list_is_reported():
    g = synth.random_lsdb(80, 0, 3.0, 9100, metric_hi=1, zero_cost_router_links=True)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, np.arange(g.n, dtype=np.uint32), 0, go.MAP)
    small = [M.dynamic_order(g.row_ptr, g.col, g.metric, g.vflags, r, ref.dist[r], heap_cap=1) for r in range(g.n)]
    assert any(x is None for x in small)
