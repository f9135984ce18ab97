"""Host logic of holo_amd.isis (LSDB -> CSR, first-hop slot replay, resolve_nexthop, route build,
L1/L2 merge) on CPU: the engine is replaced by the CPU oracle behind the same interface
(tests/_oracle_engine.py), the answers are the reference's recorded local RIBs."""
import glob
import json
import os

import numpy as np
import pytest

from holo_amd import isis as H
from oracle import isis_ref as R
from _oracle_engine import OracleEngine

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ISIS = sorted(glob.glob(os.path.join(GOLD, "isis", "*.json")))
ISIS_STEPS = sorted(glob.glob(os.path.join(GOLD, "isis_steps", "*.json")))


@pytest.mark.parametrize("path", ISIS + ISIS_STEPS, ids=[os.path.basename(p)[:-5] for p in ISIS + ISIS_STEPS])
def test_compute_spf_reproduces_reference_local_rib(path):
    vec = json.load(open(path))
    inst = H.Instance.from_vector(vec)
    want = sorted(vec["rib"], key=lambda r: R._net_key(r["prefix"]))
    assert H.compute_spf(inst, OracleEngine()) == want


ISIS_WIRE = [p for p in ISIS_STEPS if "summary" not in os.path.basename(p)]


@pytest.mark.parametrize("path", ISIS_WIRE, ids=[os.path.basename(p)[:-5] for p in ISIS_WIRE])
def test_update_global_rib_reproduces_recorded_ibus_messages(path):
    """SPF + route build of the host twin, then the wire step (update_global_rib): the RouteIpAdd / RouteIpDel messages
    the reference recorded for the step, in order."""
    vec = json.load(open(path))
    inst = H.Instance.from_vector(vec)
    want = [{k: m[k] for k in m if k != "distance"} for m in vec["ibus_routes"]]
    got = H.update_global_rib(H.compute_spf(inst, OracleEngine()), vec["rib_before"], vec["ifindex"])
    assert got == want and got == R.update_global_rib(R.local_rib(vec), vec["rib_before"], vec["ifindex"])


def check_spts_against_ref(vec, inst, engine):
    """Every system as root in one batched run (the flooding::manet::init_cache shape), local and
    hop-count variants, against the literal restatement: distance, hops, next-hop system ids and
    first/second-hop lists in pop order."""
    for level in inst.config.levels():
        if level not in inst.lsdb:
            continue                          # instance disabled: empty LSDB, nothing to compute
        systems = sorted({l.system_id for l in inst.lsdb[level].iter()})
        for mt_id, hopcount, local in ((0, False, False), (None, True, False), (0, False, True)):
            roots = systems if not local else [inst.config.system_id]
            spts = H.compute_spts(level, roots, local, mt_id, hopcount, inst, engine)
            for sid, spt in zip(roots, spts):
                ref, order = R.compute_spt(vec, level, sid, local, mt_id, hopcount)
                assert set(spt.vertices) == set(ref)
                for vid, vx in ref.items():
                    got = spt.get(vid)
                    assert (got.distance, got.hops) == (vx.distance, vx.hops)
                    assert ({(n.system_id, n.iface_name, n.ipv4, n.ipv6) for n in got.nexthops}
                            == {(n["system_id"], n["iface"], n["ipv4"], n["ipv6"]) for n in vx.nexthops})
                    assert spt.parents(vid) == vx.parents
                assert [v.id for v in spt.first_hops()] == [v for v in order if v[0] and ref[v].hops == 1]
                assert [v.id for v in spt.second_hops()] == [v for v in order if v[0] and ref[v].hops == 2]
                # Spt::is_on_path (holo-isis/src/spf.rs:261-286), every pair of systems
                def ref_on_path(a, d):
                    stack, seen = [d], set()
                    while stack:
                        c = stack.pop()
                        if c == a:
                            return True
                        if c in seen:
                            continue
                        seen.add(c); stack.extend(ref[c].parents)
                    return False
                routers = [v for v in ref if v[0]]
                for a in routers:
                    for d in routers:
                        assert spt.is_on_path(a[1], d[1]) == ref_on_path(a, d)


@pytest.mark.parametrize("path", ISIS[::3], ids=[os.path.basename(p)[:-5] for p in ISIS[::3]])
def test_batched_roots_match_literal_restatement(path):
    vec = json.load(open(path))
    check_spts_against_ref(vec, H.Instance.from_vector(vec), OracleEngine())


def test_root_without_lsp_is_alone_in_its_spt():
    vec = json.load(open(ISIS[0]))
    inst = H.Instance.from_vector(vec)
    spt = H.compute_spt(inst.config.levels()[0], b"\xaa" * 6, True, 0, False, inst, OracleEngine())
    assert [v.id for v in spt.iter()] == [(True, b"\xaa" * 6, 0)]


def test_manet_init_cache_is_one_batched_run():
    vec = json.load(open(ISIS[8]))
    inst = H.Instance.from_vector(vec)

    class Counting(OracleEngine):
        runs = 0

        def run(self, *a, **k):
            Counting.runs += 1
            return super().run(*a, **k)
    level = inst.config.levels()[0]
    cache = H.manet_init_cache(level, inst, Counting())
    nbrs = {a.system_id for i in inst.interfaces for a in i.adjacencies if a.state == "up"}
    assert set(cache) == nbrs and Counting.runs == 1
    for sid, c in cache.items():
        ref, order = R.compute_spt(vec, level, sid, False, None, True)
        assert list(c.remote_nbr_list) == sorted(v[1] for v in order if v[0] and ref[v].hops == 1)


# ---- the recorded cold-start wire output of every topology router (tests/golden/wire/isis, tools/make_golden_wire.py) -------
import _wire as W        # noqa: E402


@pytest.mark.parametrize("path", W.wire_paths("isis"), ids=[os.path.basename(p)[:-5] for p in W.wire_paths("isis")])
def test_cold_start_messages_reproduce_recorded_ibus_state(path):
    W.check_isis_cold_start(path, OracleEngine())
