"""Differential test of the OSPFv3 host twin on random areas (tests/_random_ospfv3.py), CPU only: holo_amd.ospfv3 with
the oracle engine behind it against the literal restatement (oracle/ospfv3_ref.py): intra-area RIB and whole SPT."""
import pytest

from holo_amd import ospfv3 as H3
from oracle import ospfv3_ref as R3
from _oracle_engine import OracleEngine
from _random_ospfv3 import make


def check(vec, eng):
    areas = [H3.Area3.from_vector(a) for a in vec["areas"]]
    assert H3.compute_spf_intra_area(vec["router_id"], areas, vec["max_paths"], eng, vec["af"]) == R3.intra_area_rib(vec)
    ref = R3.run_area(vec, vec["areas"][0])
    spt = H3.run_area(vec["router_id"], areas[0], eng, vec["af"])
    if ref is None:
        assert spt is None
        return
    assert set(spt) == set(ref[0])
    for vid, vx in ref[0].items():
        assert (spt[vid].distance, spt[vid].hops) == (vx.distance, vx.hops), vid
        assert spt[vid].nexthops == vx.nexthops, vid


@pytest.mark.parametrize("block", range(8))
def test_random_ospfv3_areas_rib_and_spt(block):
    eng = OracleEngine()
    for seed in range(block * 40, block * 40 + 40):
        check(make(seed), eng)
