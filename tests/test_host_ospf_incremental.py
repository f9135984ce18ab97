"""OSPFv2 LSDB -> CSR kept up to date from the changed LSAs (holo_amd.ospf.AreaGraph.refresh / GraphCache,
SURVEY.md §8f-1) on CPU: every reference step test is replayed as "topology snapshot, then the LSDB after the step";
the cached graphs must equal from-scratch ones and the SPF on them must give the reference's recorded routes."""
import glob
import json
import os
import re

import numpy as np
import pytest

from holo_amd import ospf as HO
from oracle import ospf_ref as RO
from _oracle_engine import OracleEngine

GOLD = os.path.join(os.path.dirname(__file__), "golden")
STEPS = sorted(glob.glob(os.path.join(GOLD, "ospfv2_steps", "*.json")))


def base_of(step_vec):
    topo, rt = re.search(r"snapshot (topo[\d-]+)/(rt\d+)", step_vec["source"]).groups()
    return json.load(open(os.path.join(GOLD, "ospfv2", f"{topo}_{rt}.json")))


def intra(vec):
    return sorted([r for r in vec["rib"] if r["type"] == "intra-area"], key=lambda r: RO._net_key(r["prefix"]))


def replay_ospf_step(step, eng):
    base = base_of(step)
    cache = HO.GraphCache()
    areas0 = [HO.Area.from_vector(a) for a in base["areas"]]
    got0 = HO.compute_spf_intra_area(base["router_id"], areas0, base["max_paths"], eng, cache)
    assert got0 == RO.intra_area_rib(base)
    areas1 = [HO.Area.from_vector(a) for a in step["areas"]]
    old = {a.area_id: a for a in areas0}
    trig = {a.area_id: HO.changed_vertex_ids(old[a.area_id], a) for a in areas1 if a.area_id in old}
    got1 = HO.compute_spf_intra_area(step["router_id"], areas1, step["max_paths"], eng, cache, trig)
    assert got1 == RO.intra_area_rib(step)
    if not step["has_vlinks"]:
        assert got1 == intra(step)
    for a in areas1:
        g, fresh = cache.graphs[a.area_id], HO.AreaGraph(a)
        assert g.vids == fresh.vids and g.link_pos == fresh.link_pos and g.link_ref == fresh.link_ref
        for name in ("row_ptr", "col", "metric", "vflags"):
            assert np.array_equal(getattr(g, name), getattr(fresh, name)), (a.area_id, name)
            assert np.array_equal(getattr(g._dev[1], name), getattr(fresh, name)), ("device mirror", name)
            if hasattr(g._dev[1], "export"):                          # the real engine: what sits in HBM
                assert np.array_equal(g._dev[1].export(name), getattr(fresh, name)), ("device", name)
    for g in cache.graphs.values():
        if g._dev is not None:
            g._dev[1].free()


@pytest.mark.parametrize("path", STEPS, ids=[os.path.basename(p)[:-5] for p in STEPS])
def test_step_replayed_through_the_graph_cache(path):
    replay_ospf_step(json.load(open(path)), OracleEngine())


def test_some_steps_really_are_row_patches():
    patched = 0
    for path in STEPS:
        step = json.load(open(path)); base = base_of(step)
        old = {a["area_id"]: HO.Area.from_vector(a) for a in base["areas"]}
        for a in step["areas"]:
            if a["area_id"] not in old:
                continue
            new = HO.Area.from_vector(a)
            ch = HO.changed_vertex_ids(old[a["area_id"]], new)
            g = HO.AreaGraph(old[a["area_id"]])
            if ch and g.refresh(new, ch):
                patched += 1
    assert patched >= 2
