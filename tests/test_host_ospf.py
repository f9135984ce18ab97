"""Host logic of holo_amd.ospf (LSA walk -> CSR, per-slot calc_nexthops, intra-area route build)
on CPU, engine replaced by the CPU oracle behind the same interface; answers = the reference's
recorded intra-area routes (ordered next hops included)."""
import glob
import json
import os

import pytest

from holo_amd import ospf as HO
from oracle import ospf_ref as RO
from _oracle_engine import OracleEngine

GOLD = os.path.join(os.path.dirname(__file__), "golden")
OSPF = sorted(glob.glob(os.path.join(GOLD, "ospfv2", "*.json"))) + sorted(glob.glob(os.path.join(GOLD, "ospfv2_steps", "*.json")))


def check_ospf_vector(vec, engine):
    areas = [HO.Area.from_vector(a) for a in vec["areas"]]
    got = HO.compute_spf_intra_area(vec["router_id"], areas, vec["max_paths"], engine)
    # 1. against the literal restatement (every vector, virtual links included)
    assert got == RO.intra_area_rib(vec)
    for a, area in zip(vec["areas"], areas):
        ref = RO.run_area(vec, a)
        spt = HO.run_area(vec["router_id"], area, engine)
        if ref is None:
            assert spt is None
            continue
        assert set(spt) == set(ref[0])
        for vid, vx in ref[0].items():
            assert (spt[vid].distance, spt[vid].hops) == (vx.distance, vx.hops)
            assert spt[vid].nexthops == vx.nexthops
    # 2. against the reference's own recorded answer
    if not vec["has_vlinks"]:
        want = sorted([r for r in vec["rib"] if r["type"] == "intra-area"], key=lambda r: RO._net_key(r["prefix"]))
        assert got == want


@pytest.mark.parametrize("path", OSPF, ids=[os.path.basename(p)[:-5] for p in OSPF])
def test_run_area_and_intra_area_rib(path):
    check_ospf_vector(json.load(open(path)), OracleEngine())


OSPF_STEPS = [p for p in sorted(glob.glob(os.path.join(GOLD, "ospfv2_steps", "*.json"))) if "ibus_routes" in json.load(open(p))]


def wire_rows(vec, intra_rows):
    """The RIB after the step as the wire step sees it: this path's intra-area rows + the rows of the calculations
    outside it (inter-area, external), which the vectors record."""
    return intra_rows + [r for r in vec["rib"] if r["type"] != "intra-area"]


@pytest.mark.parametrize("path", OSPF_STEPS, ids=[os.path.basename(p)[:-5] for p in OSPF_STEPS])
def test_update_global_rib_reproduces_recorded_ibus_messages(path):
    """SPF + intra-area route build of the host twin, then the wire step: the RouteIpAdd / RouteIpDel messages the
    reference recorded on the ibus for the step, in order (7 of the 11 steps hold intra-area routes only; the other four
    take their inter-area rows from the recording: that calculation is outside this path)."""
    vec = json.load(open(path))
    want = [{k: m[k] for k in m if k != "distance"} for m in vec["ibus_routes"]]
    areas = [HO.Area.from_vector(a) for a in vec["areas"]]
    rows = HO.compute_spf_intra_area(vec["router_id"], areas, vec["max_paths"], OracleEngine())
    got = HO.update_global_rib(wire_rows(vec, rows), vec["rib_before"], vec["ifindex"])
    assert got == want
    assert got == RO.update_global_rib(wire_rows(vec, RO.intra_area_rib(vec)), vec["rib_before"], vec["ifindex"])


# ---- OSPFv3 -----------------------------------------------------------------------------------------
from holo_amd import ospfv3 as H3        # noqa: E402
from oracle import ospfv3_ref as R3      # noqa: E402

OSPF3 = sorted(glob.glob(os.path.join(GOLD, "ospfv3", "*.json")))


def check_ospfv3_vector(vec, engine):
    areas = [H3.Area3.from_vector(a) for a in vec["areas"]]
    got = H3.compute_spf_intra_area(vec["router_id"], areas, vec["max_paths"], engine, vec["af"])
    assert got == R3.intra_area_rib(vec)
    for a, area in zip(vec["areas"], areas):
        ref = R3.run_area(vec, a)
        spt = H3.run_area(vec["router_id"], area, engine, vec["af"])
        if ref is None:
            assert spt is None
            continue
        assert set(spt) == set(ref[0])
        for vid, vx in ref[0].items():
            assert (spt[vid].distance, spt[vid].hops) == (vx.distance, vx.hops)
            assert spt[vid].nexthops == vx.nexthops
    if not vec["has_vlinks"]:
        want = sorted([r for r in vec["rib"] if r["type"] == "intra-area"], key=lambda r: R3._net_key(r["prefix"]))
        assert got == want


@pytest.mark.parametrize("path", OSPF3, ids=[os.path.basename(p)[:-5] for p in OSPF3])
def test_ospfv3_run_area_and_intra_area_rib(path):
    check_ospfv3_vector(json.load(open(path)), OracleEngine())
