"""Differential test of the OSPFv2 host twin on random areas (tests/_random_ospf.py), CPU only: holo_amd.ospf with the
oracle engine behind it against the literal restatement of run_area / calc_nexthops / update_rib_intra_area
(oracle/ospf_ref.py, pinned to the reference's recorded RIBs): the intra-area RIB and the whole SPT of the area."""
import pytest

from holo_amd import ospf as HO
from oracle import ospf_ref as RO
from _oracle_engine import OracleEngine
from _random_ospf import make


def check(vec, eng):
    areas = [HO.Area.from_vector(a) for a in vec["areas"]]
    assert HO.compute_spf_intra_area(vec["router_id"], areas, vec["max_paths"], eng) == RO.intra_area_rib(vec)
    ref = RO.run_area(vec, vec["areas"][0])
    spt = HO.run_area(vec["router_id"], areas[0], eng)
    if ref is None:
        assert spt is None
        return
    assert set(spt) == set(ref[0])
    for vid, vx in ref[0].items():
        assert (spt[vid].distance, spt[vid].hops) == (vx.distance, vx.hops), vid
        assert spt[vid].nexthops == vx.nexthops, vid


@pytest.mark.parametrize("block", range(8))
def test_random_areas_rib_and_spt(block):
    eng = OracleEngine()
    for seed in range(block * 40, block * 40 + 40):
        check(make(seed), eng)


@pytest.mark.parametrize("block", range(4))
def test_random_areas_with_zero_cost_links_rib_and_spt(block):
    """A third of the p2p metrics at 0: dynamic pop orders (the twin orders the SPT by the engine's pop ranks)."""
    eng = OracleEngine()
    for seed in range(7000 + block * 40, 7000 + block * 40 + 40):
        check(make(seed, zero=True), eng)
