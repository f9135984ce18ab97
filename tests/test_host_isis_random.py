"""Differential test of the IS-IS host twin on random instances (tests/_random_isis.py), CPU only: holo_amd.isis with
the oracle engine behind it against the literal restatement of compute_spt / compute_routes (oracle/isis_ref.py, itself
pinned to the reference's recorded RIBs): the local RIB, and for every system as root the whole SPT (distance, hops,
next hops with interface and addresses, parents, first / second hops in pop order, is_on_path)."""
import pytest

from holo_amd import isis as H
from oracle import isis_ref as R
from _oracle_engine import OracleEngine
from _random_isis import make
from test_host_isis import check_spts_against_ref


@pytest.mark.parametrize("block", range(8))
def test_random_instances_local_rib_and_all_roots(block):
    eng = OracleEngine()
    for seed in range(block * 25, block * 25 + 25):
        vec = make(seed)
        inst = H.Instance.from_vector(vec)
        assert H.compute_spf(inst, eng) == R.local_rib(vec), seed
        if seed % 5 == 0:
            check_spts_against_ref(vec, H.Instance.from_vector(vec), eng)


@pytest.mark.parametrize("block", range(4))
def test_random_instances_with_zero_metrics_local_rib_and_all_roots(block):
    """A third of the link metrics at 0 (legal: holo-isis/src/spf.rs:629-704 adds them like any metric): pop orders are dynamic,
    `first_hops` / `second_hops` lists and the slot replay follow the engine's pop ranks (HSPF_RF_EXACT -> HSPF_RUN_POP_RANK)."""
    eng = OracleEngine()
    for seed in range(5000 + block * 25, 5000 + block * 25 + 25):
        vec = make(seed, zero=True)
        inst = H.Instance.from_vector(vec)
        assert H.compute_spf(inst, eng) == R.local_rib(vec), seed
        if seed % 3 == 0:
            check_spts_against_ref(vec, H.Instance.from_vector(vec), eng)


@pytest.mark.parametrize("block", range(4))
def test_random_two_level_instances_local_rib(block):
    """level-all instances (tests/_random_isis.py make_two_level): one SPT and route set per level, L1 over L2 in the merge
    (holo-isis/src/route.rs:185-249), the default route of L1 routers towards attached L2 routers."""
    from _random_isis import make_two_level
    eng = OracleEngine()
    both = 0
    for seed in range(block * 50, block * 50 + 50):
        vec = make_two_level(seed)
        want = R.local_rib(vec)
        assert H.compute_spf(H.Instance.from_vector(vec), eng) == want, seed
        both += len({r["level"] for r in want}) > 1
    assert both >= 20


def test_random_long_chains_beyond_the_maximum_path_metric():
    """22-30 routers in a line with narrow metrics near 63 (tests/_random_isis.py make_long): path metrics cross
    MAX_PATH_METRIC_STANDARD = 1023 (holo-isis/src/spf.rs:637-641) and the routers beyond stay off the SPT."""
    from _random_isis import make_long
    eng = OracleEngine()
    cut = 0
    for seed in range(100):
        vec = make_long(seed)
        want = R.local_rib(vec)
        assert H.compute_spf(H.Instance.from_vector(vec), eng) == want, seed
        cut += sum(1 for r in want if r["prefix"].endswith("/32")) < len(vec["lsdb"]["2"])
    assert cut >= 40


@pytest.mark.parametrize("block", range(3))
def test_random_multi_topology_instances_local_rib(block):
    """MT IPv6-unicast instances (tests/_random_isis.py make_mt): the standard topology carries IPv4 only, topology 2 its own
    links, metrics, prefixes and overload / attached bits; one SPT per topology, one RIB."""
    from _random_isis import make_mt
    eng = OracleEngine()
    v6 = 0
    for seed in range(block * 50, block * 50 + 50):
        vec = make_mt(seed)
        want = R.local_rib(vec)
        assert H.compute_spf(H.Instance.from_vector(vec), eng) == want, seed
        v6 += any(":" in r["prefix"] and r["nexthops"] for r in want)
    assert v6 >= 20


def mutate(vec, rng):
    """A few LSP-level changes of the kind the protocol produces: overload bit flips, metric changes, a neighbour
    dropped, a fragment purged (lifetime 0)."""
    import copy
    v = copy.deepcopy(vec)
    lsps = v["lsdb"]["2"]
    for _ in range(int(rng.integers(1, 4))):
        l = lsps[int(rng.integers(0, len(lsps)))]
        what = int(rng.integers(0, 4))
        if what == 0:
            l["flags"] = [f for f in l["flags"] if f != "ol"] if "ol" in l["flags"] else l["flags"] + ["ol"]
        elif what == 1:
            for key in ("is_reach", "ext_is_reach"):
                if l[key]:
                    i = int(rng.integers(0, len(l[key])))
                    l[key][i] = [l[key][i][0], int(rng.integers(1, 60))]
        elif what == 2:
            for key in ("is_reach", "ext_is_reach"):
                if l[key]:
                    l[key].pop(int(rng.integers(0, len(l[key]))))
        else:
            l["lifetime"] = 0
    return v


@pytest.mark.parametrize("block", range(4))
def test_random_lsp_changes_through_the_graph_cache(block):
    """LSDB -> CSR kept current from the changed LSPs (GraphCache / LevelGraph.refresh): after random LSP changes the
    cached graph equals one derived from scratch and the RIB equals the literal restatement's."""
    import numpy as np
    eng = OracleEngine()
    patched = 0
    for seed in range(500 + block * 30, 500 + block * 30 + 30):
        rng = np.random.default_rng(seed)
        vec = make(seed)
        cache = H.GraphCache()
        inst = H.Instance.from_vector(vec)
        assert H.compute_spf(inst, eng, cache) == R.local_rib(vec)
        for step in range(3):
            vec2 = mutate(vec, rng)
            inst2 = H.Instance.from_vector(vec2)
            trig = {2: H.changed_lan_ids(inst.lsdb.get(2) or H.Lsdb(), inst2.lsdb.get(2) or H.Lsdb())}
            before = cache.patched
            assert H.compute_spf(inst2, eng, cache, trig) == R.local_rib(vec2), (seed, step)
            patched += cache.patched - before
            for (level, mt_id, hc), g in cache.graphs.items():
                fresh = H.LevelGraph(inst2, level, mt_id, hc)
                assert g.vids == fresh.vids
                for name in ("row_ptr", "col", "metric", "vflags"):
                    assert np.array_equal(getattr(g, name), getattr(fresh, name)), (seed, step, name)
            vec, inst = vec2, inst2
    assert patched > 20           # most changes keep the vertex set: they must go through refresh(), not a rebuild
