#!/usr/bin/env python3
"""tools/make_golden_wire.py — the reference's recorded COLD-START wire output as golden vectors.

Every topology router directory of the reference's conformance fixtures holds `output/ibus.jsonl`: the messages the
instance put on the ibus while the recorded topology converged (subscriptions, then `RouteIpAdd` as routes were
installed, `RouteIpDel` when one went away).  The step tests' ibus outputs are extracted by tools/make_golden.py /
make_golden_ospf.py; this script takes the 132 topology recordings (IS-IS 38, OSPFv2 50 — topo1-3 and topo2-4 have no
ibus recording —, OSPFv3 44) and reduces each to the state the route manager ends up with:

    final[prefix] = the last RouteIpAdd of the prefix, unless a RouteIpDel followed it

i.e. metric, distance, tag and the next-hop set (ifindex, address, labels) of every route the reference had installed
when the recording ended — what `update_global_rib` from an empty RIB must produce for the final LSDB (the order of
messages over time depends on LSA / LSP arrival; the order of a one-shot run is the RIB's key order and is checked
against the restatements' own rule).  Interface indices come from the `InterfaceUpd` events of `events.jsonl`.

Only test DATA is extracted (no reference source).  Runs where /root/reference exists:

    python tools/make_golden_wire.py
"""
from __future__ import annotations

import glob
import json
import os
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

BASES = {
    "isis": "holo-isis/tests/conformance/topologies",
    "ospfv2": "holo-ospf/tests/conformance/ospfv2/topologies",
    "ospfv3": "holo-ospf/tests/conformance/ospfv3/topologies",
}


def _nexthop(n):
    """holo-utils/src/southbound.rs Nexthop: Address{ifindex, addr, labels} | Interface{ifindex} | Special(..) | Recursive{..}."""
    if "Address" in n:
        a = n["Address"]
        return [a["ifindex"], a["addr"], list(a.get("labels", []))]
    if "Interface" in n:
        return [n["Interface"]["ifindex"], None, []]
    raise ValueError(f"next-hop kind not expected on this path: {n}")


def wire_vector(rt_dir: str, proto: str) -> dict:
    final, order, n_add, n_del = {}, 0, 0, 0
    for line in open(os.path.join(rt_dir, "output", "ibus.jsonl")):
        m = json.loads(line)
        if "RouteIpAdd" in m:
            a = m["RouteIpAdd"]
            assert a["protocol"] == proto, (rt_dir, a["protocol"])
            n_add += 1
            final[a["prefix"]] = {"prefix": a["prefix"], "metric": a["metric"], "distance": a["distance"], "tag": a["tag"],
                                  "nexthops": [_nexthop(n) for n in a["nexthops"]], "last_msg": order}
        elif "RouteIpDel" in m:
            n_del += 1
            final.pop(m["RouteIpDel"]["prefix"], None)
        order += 1
    ifx = {}
    for line in open(os.path.join(rt_dir, "events.jsonl")):
        ev = json.loads(line)
        ev = ev.get("Ibus", ev)
        if isinstance(ev, dict) and "InterfaceUpd" in ev:
            ifx[ev["InterfaceUpd"]["ifname"]] = ev["InterfaceUpd"]["ifindex"]
    rows = sorted(final.values(), key=lambda r: r["last_msg"])
    for r in rows:
        del r["last_msg"]
    return {"source": os.path.relpath(os.path.join(rt_dir, "output", "ibus.jsonl"), REF), "proto": proto,
            "ifindex": ifx, "n_route_add": n_add, "n_route_del": n_del, "final": rows}


def make_wire():
    total = {}
    for proto, base in BASES.items():
        out = os.path.join(OUT, "wire", proto)
        os.makedirs(out, exist_ok=True)
        n = 0
        for f in sorted(glob.glob(os.path.join(REF, base, "topo*", "rt*", "output", "ibus.jsonl"))):
            rt = os.path.dirname(os.path.dirname(f))
            v = wire_vector(rt, proto)
            name = f"{os.path.basename(os.path.dirname(rt))}_{os.path.basename(rt)}.json"
            json.dump(v, open(os.path.join(out, name), "w"), separators=(",", ":"), sort_keys=True)
            n += 1
        total[proto] = n
        print(f"wire/{proto}: {n} vectors -> {out}")
    return total


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    make_wire()
