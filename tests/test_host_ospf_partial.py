"""SpfComputation::{Full, Partial} dispatch of compute_spf (holo-ospf/src/spf.rs:48-60, 489-584, route.rs:200-237) on random
OSPFv3 areas, CPU only: after a FULL run, Intra-Area-Prefix-LSAs are changed (metrics, prefixes added / removed, NU bit,
LSAs withdrawn); the twin's PARTIAL run — stored SPTs, no engine call — must give (a) what the literal restatement of
update_rib_partial gives and (b) what a full run on the new LSDB gives (the topology did not change)."""
import copy
import random

import pytest

from holo_amd import ospfv3 as H3
from oracle import ospfv3_ref as R3
from oracle.ospf_ref import ip
from _oracle_engine import OracleEngine
from _random_ospfv3 import make


def mutate_iaps(vec, rng):
    """Changes some Intra-Area-Prefix-LSAs of a vector; returns the trigger list [{"new":..., "old":...}]."""
    trig = []
    for area in vec["areas"]:
        for lsa in list(area["iaps"]):
            if rng.random() < 0.5:
                continue
            old = copy.deepcopy(lsa)
            what = rng.choice(["metric", "drop-prefix", "add-prefix", "nu", "withdraw"])
            if what == "metric" and lsa["prefixes"]:
                rng.choice(lsa["prefixes"])["metric"] = rng.randint(0, 40)
            elif what == "drop-prefix" and lsa["prefixes"]:
                lsa["prefixes"].pop(rng.randrange(len(lsa["prefixes"])))
            elif what == "add-prefix":
                lsa["prefixes"].append({"prefix": f"2001:db8:{rng.randint(1, 40):x}::/64", "metric": rng.randint(0, 30), "options": []})
            elif what == "nu" and lsa["prefixes"]:
                p = rng.choice(lsa["prefixes"])
                p["options"] = [] if "nu-bit" in p["options"] else ["nu-bit"]
            elif what == "withdraw":
                area["iaps"].remove(lsa)
                new = dict(old, prefixes=[])                 # the MaxAge copy that triggers the run carries no prefixes to add
                trig.append({"new": {"function": "intra-area-prefix", "prefixes": new["prefixes"]},
                             "old": {"function": "intra-area-prefix", "prefixes": old["prefixes"]}})
                continue
            trig.append({"new": {"function": "intra-area-prefix", "prefixes": lsa["prefixes"]},
                         "old": {"function": "intra-area-prefix", "prefixes": old["prefixes"]}})
    return trig


@pytest.mark.parametrize("block", range(6))
def test_partial_run_equals_restatement_and_full_run(block):
    eng = OracleEngine()
    for seed in range(block * 30, block * 30 + 30):
        vec = make(seed)
        rng = random.Random(seed)
        st = H3.SpfState(vec["router_id"], vec["max_paths"], eng, vec["af"])
        areas = [H3.Area3.from_vector(a) for a in vec["areas"]]
        rows0 = st.run(areas)
        assert rows0 == R3.intra_area_rib(vec)
        runs = st.engine_runs
        # the restatement's state after the full run
        ordered = sorted(vec["areas"], key=lambda a: ip(a["area_id"]))
        spts = [(R3.run_area(vec, a) or (None,))[0] for a in ordered]
        rib = {}
        for a, spt in zip(ordered, spts):
            if spt is not None:
                R3.update_rib_intra_area(rib, a, spt, vec["max_paths"])
        for step in range(3):
            trig = mutate_iaps(vec, rng)
            kind, partial = R3.spf_computation_type(trig)
            assert (kind, partial) == H3.spf_computation_type(trig, 3) and kind == "partial"
            ordered = sorted(vec["areas"], key=lambda a: ip(a["area_id"]))
            rib = R3.update_rib_partial_intra(rib, partial["intra"], list(zip(ordered, spts)), vec["max_paths"])
            rows = st.run([H3.Area3.from_vector(a) for a in vec["areas"]], trig)
            assert st.engine_runs == runs, "a partial run must not call the engine"
            assert rows == R3.rows_of(rib)
            assert rows == R3.intra_area_rib(vec), "partial run == full run on the new LSDB (topology unchanged)"


def test_classification_full_vs_partial():
    iap = {"function": "intra-area-prefix", "prefixes": [{"prefix": "2001:db8:1::/64", "metric": 1, "options": []}]}
    for fn in ("router", "network", "link", "router-info"):
        assert H3.spf_computation_type([{"new": iap, "old": None}, {"new": {"function": fn}, "old": None}], 3) == ("full", None)
    kind, partial = H3.spf_computation_type([{"new": {"function": "inter-area-prefix"}, "old": None}], 3)
    assert kind == "partial" and partial["intra"] == set()
    # OSPFv2: Router-/Network-LSAs and the SR opaque LSAs are full runs; summaries / externals are partial with no intra part
    for fn in H3.FULL_FUNCTIONS_V2:
        assert H3.spf_computation_type([{"new": {"function": fn}, "old": None}], 2) == ("full", None)
    assert H3.spf_computation_type([{"new": {"function": "summary-network"}, "old": None}], 2) == ("partial", {"intra": set()})


def test_ospfv2_dispatch_full_runs_the_engine_partial_keeps_spt_routers_and_rib():
    """OSPFv2 (ospfv2/spf.rs:99-170): Router- / Network-LSA changes are Full computations — every area through the
    engine, SPTs / router tables / TransitCapability / intra-area RIB rebuilt and equal to the literal loop's —;
    summary and external LSA changes are Partial ones with an EMPTY intra-area part: no engine call, nothing of the
    path's state moves."""
    import glob
    import json
    import os
    from holo_amd import ospf as HO
    from oracle import ospf_ref as RO
    eng = OracleEngine()
    gold = os.path.join(os.path.dirname(__file__), "golden", "ospfv2")
    for path in sorted(glob.glob(os.path.join(gold, "topo3-*_rt*.json")))[:8] + sorted(glob.glob(os.path.join(gold, "topo2-*_rt1.json"))):
        vec = json.load(open(path))
        areas = [HO.Area.from_vector(a) for a in vec["areas"]]
        st = HO.SpfState(vec["router_id"], vec["max_paths"], eng)
        rows = st.run(areas)
        assert rows == RO.intra_area_rib(vec) and st.engine_runs == len(areas)
        for a in vec["areas"]:
            side = {}
            RO.run_area(vec, a, side)
            assert st.routers[a["area_id"]] == side["routers"] and st.transit_capability[a["area_id"]] == side["transit_capability"]
        runs, spts, routers = st.engine_runs, dict(st.spts), dict(st.routers)
        for fn in ("summary-network", "summary-router", "as-external"):
            assert st.run(areas, [{"new": {"function": fn}, "old": None}]) == rows
            assert st.engine_runs == runs and st.spts == spts and st.routers == routers
        assert st.run(areas, [{"new": {"function": "router"}, "old": None}]) == rows        # a Full one: the engine again
        assert st.engine_runs == runs + len(areas)


def test_root_lsa_missing_keeps_the_router_table_and_spt_of_the_area():
    """run_area returns at SpfRootNotFound BEFORE `routers.clear()` and without touching the area's SPT
    (holo-ospf/src/spf.rs:596-620): only TransitCapability has been reset by then.  (the twin used to empty both.)"""
    import copy
    import glob
    import json
    import os
    from holo_amd import ospf as HO
    eng = OracleEngine()
    gold = os.path.join(os.path.dirname(__file__), "golden", "ospfv2")
    vec = json.load(open(sorted(glob.glob(os.path.join(gold, "topo2-1_rt1.json")))[0]))
    st = HO.SpfState(vec["router_id"], vec["max_paths"], eng)
    before = list(st.run([HO.Area.from_vector(a) for a in vec["areas"]]))
    aid = vec["areas"][0]["area_id"]
    routers, spt = dict(st.routers[aid]), st.spts[aid]
    assert routers and spt is not None
    gone = copy.deepcopy(vec)
    gone["areas"][0]["routers"] = [r for r in gone["areas"][0]["routers"] if r["adv_rtr"] != vec["router_id"]]
    rows = st.run([HO.Area.from_vector(a) for a in gone["areas"]])
    assert st.transit_capability[aid] is False
    assert st.routers[aid] == routers and st.spts[aid] is spt, "the previous run's router table and SPT stay"
    # ... and update_rib_full still folds the area from that SPT (holo-ospf/src/route.rs:157-160; Ospfv2::intra_area_networks
    # walks area.state.spt): the routes of the previous run stay in the RIB (the twin used to drop them)
    assert rows == before and rows
    # a router that never had an SPT for the area contributes nothing
    fresh = HO.SpfState(vec["router_id"], vec["max_paths"], eng)
    if len(vec["areas"]) == 1:
        assert fresh.run([HO.Area.from_vector(a) for a in gone["areas"]]) == []


def test_ospfv3_root_lsa_missing_keeps_the_routes_of_the_stored_spt():
    import copy
    import glob
    import json
    import os
    from holo_amd import ospfv3 as H3
    eng = OracleEngine()
    gold = os.path.join(os.path.dirname(__file__), "golden", "ospfv3")
    vec = json.load(open(sorted(glob.glob(os.path.join(gold, "topo1-1_rt1.json")))[0]))
    st = H3.SpfState(vec["router_id"], vec["max_paths"], eng, vec["af"])
    before = st.run([H3.Area3.from_vector(a) for a in vec["areas"]])
    gone = copy.deepcopy(vec)
    gone["areas"][0]["routers"] = [r for r in gone["areas"][0]["routers"] if r["adv_rtr"] != vec["router_id"]]
    assert before and st.run([H3.Area3.from_vector(a) for a in gone["areas"]]) == before
