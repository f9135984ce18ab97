"""SR Prefix-SID bookkeeping of compute_routes (holo-isis/src/spf.rs:931-946 -> sr.rs:34-94, 165-300), CPU only.

The SPT feeds this step two bits per route update — local = (vertex.hops == 0), last_hop = (vertex.hops == 1) — the rest
is label arithmetic over the LSDB's SR-Capabilities.  `sr.enabled` is off in every conformance fixture of the reference
that records a RIB, so nothing here is pinned to recorded output: the host twin (holo_amd.isis) is compared with the
literal restatement (oracle/isis_ref.py) on random instances, and both with labels worked out by hand from the cited
lines on a three-router chain."""
import copy

import pytest

from holo_amd import isis as H
from oracle import isis_ref as R
from _oracle_engine import OracleEngine
from _random_isis import add_sr, make


def rid(i):
    return f"0000.0000.{i:04x}"


def chain(flags2=(), flags3=(), sid3=None, srgb2=((17000, 1000),), algos3=(0,), cap2_flags=("I", "V")):
    """R1 (local) - R2 - R3, wide metrics 10 + 10; each advertises i.i.i.i/32 with SID index 10 i."""
    def lsp(i, nbrs, flags, sid, srgb, algos=(0,), cap_flags=("I", "V")):
        return {"id": f"{rid(i)}.00-00", "flags": [], "protocols": [204, 142], "is_reach": [],
                "ext_is_reach": [[f"{rid(n)}.00", 10] for n in nbrs], "mt": [], "mt_is_reach": [], "mt_ipv6": [],
                "ipv4_int": [], "ipv4_ext": [], "ext_ipv4": [[f"{i}.{i}.{i}.{i}/32", 0, False]], "ipv6": [],
                "sr_cap": {"flags": list(cap_flags), "srgb": [list(x) for x in srgb]}, "sr_algos": list(algos),
                "prefix_sids": {"ext_ipv4": {"0": sid if sid is not None else {"flags": list(flags), "index": 10 * i}}}}
    lsps = [lsp(1, [2], (), None, ((16000, 1000),)), lsp(2, [1, 3], flags2, None, srgb2, cap_flags=cap2_flags),
            lsp(3, [2], flags3, sid3, ((18000, 1000),), algos3)]
    ifaces = [{"name": "eth0", "type": "point-to-point", "metric": {"1": 10, "2": 10},
               "adjacencies": [{"system_id": rid(2), "usage": "level-2", "state": "up", "ipv4": ["10.0.0.2"], "ipv6": [],
                                "topologies": [0], "area_addrs": ["49.0000"]}]}]
    return {"proto": "isis", "source": "sr chain",
            "config": {"system_id": rid(1), "level_type": "level-2", "metric_type": {"1": "wide", "2": "wide"},
                       "afs": {"ipv4": True, "ipv6": False}, "mt_ipv6_unicast": False, "max_paths": 16, "att_ignore": False,
                       "area_addrs": ["49.0000"], "sr_enabled": True},
            "interfaces": ifaces, "lsdb": {"2": lsps}, "rib": []}


def both(vec):
    got = H.compute_spf(H.Instance.from_vector(vec), OracleEngine())
    want = R.local_rib(vec)
    assert got == want
    return {r["prefix"]: (r["sr_label"], r["nexthop_labels"]) for r in got}


def test_labels_on_a_chain_by_hand():
    rib = both(chain())
    # own prefix: hops 0 = local, no P flag -> no input label; R2's: last hop, no P flag -> implicit null out, input label
    # from the LOCAL SRGB; R3's: two hops away -> R2's SRGB out
    assert rib["1.1.1.1/32"] == (None, [])
    assert rib["2.2.2.2/32"] == (16000 + 20, [R.LABEL_IMPLICIT_NULL])
    assert rib["3.3.3.3/32"] == (16000 + 30, [17000 + 30])
    # P flag on the last hop: no penultimate-hop popping -> the neighbour's own label; with E as well: explicit null
    assert both(chain(flags2=("P",)))["2.2.2.2/32"] == (16020, [17020])
    assert both(chain(flags2=("P", "E")))["2.2.2.2/32"] == (16020, [R.LABEL_EXPLICIT_NULL_V4])
    # E on a prefix two hops away changes nothing (not the last hop)
    assert both(chain(flags3=("P", "E")))["3.3.3.3/32"] == (16030, [17030])
    # absolute label (V/L): input = the label; output on a non-last hop = implicit null (local significance)
    assert both(chain(sid3={"flags": ["V", "L"], "label": 5555}))["3.3.3.3/32"] == (5555, [R.LABEL_IMPLICIT_NULL])
    # index beyond the next hop's SRGB: output label unresolved (stays None), the input label is still set; a second
    # SRGB range continues the index space
    assert both(chain(srgb2=((17000, 25),)))["3.3.3.3/32"] == (16030, [None])
    assert both(chain(srgb2=((17000, 25), (27000, 100))))["3.3.3.3/32"] == (16030, [27000 + 5])
    # the advertiser does not list the SPF algorithm: its Prefix-SID is ignored altogether
    assert both(chain(algos3=(1,)))["3.3.3.3/32"] == (None, [None])
    # next hop without MPLS for IPv4 (no I flag)
    assert both(chain(cap2_flags=("V",)))["3.3.3.3/32"] == (16030, [None])


def test_sr_off_leaves_the_rows_as_recorded():
    vec = chain()
    vec["config"]["sr_enabled"] = False
    got = H.compute_spf(H.Instance.from_vector(vec), OracleEngine())
    assert got == R.local_rib(vec) and all("sr_label" not in r for r in got)


@pytest.mark.parametrize("block", range(6))
def test_random_instances_with_sr_against_the_restatement(block):
    eng = OracleEngine()
    seen = set()
    for seed in range(block * 40, block * 40 + 40):
        vec = add_sr(make(seed), seed)
        got = H.compute_spf(H.Instance.from_vector(vec), eng)
        assert got == R.local_rib(vec), seed
        for r in got:
            seen.add("in" if r["sr_label"] is not None else "noin")
            for l in r["nexthop_labels"]:
                seen.add("null" if l in (0, 2, 3) else ("none" if l is None else "label"))
    assert {"in", "noin", "null", "none", "label"} <= seen


def test_update_global_rib_reinstalls_a_route_whose_only_change_is_an_sr_label():
    """the reference compares whole Nexthop structs (holo-isis/src/route.rs:268-277), SR output label
    included — a neighbour's SRGB change re-sends the route although metric, address and interface are the same; a
    device-side 'unchanged' verdict (metric + next-hop mask) must not suppress it."""
    old = [{"prefix": "10.0.0.0/24", "metric": 20, "level": 2, "nexthops": [["10.1.1.2", "eth0"]], "sr_label": 16005, "nexthop_labels": [17005]}]
    new = [{"prefix": "10.0.0.0/24", "metric": 20, "level": 2, "nexthops": [["10.1.1.2", "eth0"]], "sr_label": 16005, "nexthop_labels": [18005]}]
    for f in (H.update_global_rib, R.update_global_rib):
        assert f(new, old, {"eth0": 3}) == [{"op": "add", "prefix": "10.0.0.0/24", "metric": 20, "nexthops": [[3, "10.1.1.2"]]}]
        assert f(old, old, {"eth0": 3}) == []
    assert H.update_global_rib(new, old, {"eth0": 3}, unchanged=["10.0.0.0/24"]) != []
