"""LSDB -> CSR kept up to date from the changed LSPs (holo_amd.isis.LevelGraph.refresh / GraphCache)
on CPU: every reference step test is replayed as "topology snapshot, then the LSDB of the step" — the graphs built
for the snapshot are brought forward with row patches (or rebuilt when a vertex appears / disappears or the
configuration changes) and must equal graphs derived from scratch; the SPF on them must give the RIB the reference
recorded after the step."""
import glob
import json
import os
import re

import numpy as np
import pytest

from holo_amd import isis as H
from oracle import isis_ref as R
from _oracle_engine import OracleEngine

GOLD = os.path.join(os.path.dirname(__file__), "golden")
STEPS = sorted(glob.glob(os.path.join(GOLD, "isis_steps", "*.json")))


def base_of(step_vec):
    topo, rt = re.search(r"snapshot (topo[\d-]+)/(rt\d+)", step_vec["source"]).groups()
    return json.load(open(os.path.join(GOLD, "isis", f"{topo}_{rt}.json")))


def replay_isis_step(step, eng):
    base = base_of(step)
    cache = H.GraphCache()
    inst0 = H.Instance.from_vector(base)
    want0 = sorted(base["rib"], key=lambda r: R._net_key(r["prefix"]))
    assert H.compute_spf(inst0, eng, cache) == want0
    built0 = cache.rebuilt
    inst1 = H.Instance.from_vector(step)
    trig = {level: H.changed_lan_ids(inst0.lsdb.get(level) or H.Lsdb(), inst1.lsdb.get(level) or H.Lsdb())
            for level in (1, 2)}
    want1 = sorted(step["rib"], key=lambda r: R._net_key(r["prefix"]))
    assert H.compute_spf(inst1, eng, cache, trig) == want1
    # whatever mixture of patches and rebuilds happened, the cached graphs are the from-scratch ones
    for (level, mt_id, hc), g in cache.graphs.items():
        if level not in inst1.config.levels() or not inst1.config.is_topology_enabled(mt_id or 0):
            continue
        fresh = H.LevelGraph(inst1, level, mt_id, hc)
        assert g.vids == fresh.vids
        for name in ("row_ptr", "col", "metric", "vflags"):
            assert np.array_equal(getattr(g, name), getattr(fresh, name)), (level, mt_id, name)
            if g._dev is not None:
                assert np.array_equal(getattr(g._dev[1], name), getattr(fresh, name)), ("device mirror", name)
                if hasattr(g._dev[1], "export"):                      # the real engine: what sits in HBM
                    assert np.array_equal(g._dev[1].export(name), getattr(fresh, name)), ("device", name)
    assert cache.patched + cache.rebuilt - built0 >= 1
    for g in cache.graphs.values():
        if g._dev is not None:
            g._dev[1].free()


@pytest.mark.parametrize("path", STEPS, ids=[os.path.basename(p)[:-5] for p in STEPS])
def test_step_replayed_through_the_graph_cache(path):
    replay_isis_step(json.load(open(path)), OracleEngine())


def test_some_steps_really_are_row_patches():
    """At least the overload / metric steps keep the vertex set: they must go through refresh(), not a rebuild."""
    patched = 0
    for path in STEPS:
        step = json.load(open(path)); base = base_of(step)
        inst0, inst1 = H.Instance.from_vector(base), H.Instance.from_vector(step)
        for level in inst0.config.levels():
            if level not in inst0.lsdb or level not in inst1.lsdb:
                continue
            g = H.LevelGraph(inst0, level, H.MT_STANDARD)
            ch = H.changed_lan_ids(inst0.lsdb[level], inst1.lsdb[level])
            if ch and g.refresh(inst1, ch):
                patched += 1
                fresh = H.LevelGraph(inst1, level, H.MT_STANDARD)
                assert np.array_equal(g.col, fresh.col) and np.array_equal(g.metric, fresh.metric)
                assert np.array_equal(g.vflags, fresh.vflags) and np.array_equal(g.row_ptr, fresh.row_ptr)
    assert patched >= 3


def test_refresh_refuses_vertex_set_and_config_changes():
    base = json.load(open(os.path.join(GOLD, "isis", "topo1-1_rt1.json")))
    inst = H.Instance.from_vector(base)
    level = inst.config.levels()[0]
    g = H.LevelGraph(inst, level, H.MT_STANDARD)
    before = (g.row_ptr.copy(), g.col.copy())
    some = next(iter(inst.lsdb[level].iter()))
    inst.lsdb[level].insert(H.Lsp(b"\x99" * 6, 0, 0))                    # a new system appears
    assert g.refresh(inst, [(b"\x99" * 6, 0)]) is False
    assert np.array_equal(g.row_ptr, before[0]) and np.array_equal(g.col, before[1])
    inst = H.Instance.from_vector(base)
    inst.config.metric_type[level] = "standard" if inst.config.metric_type[level] != "standard" else "wide"
    assert g.refresh(inst, [some.lan_id]) is False
