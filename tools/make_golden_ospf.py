"""OSPFv2 half of tools/make_golden.py: vectors from
holo-ospf/tests/conformance/ospfv2/topologies/<topo>/<rt>/ (config.json + output/northbound-state.json).

Per area only what `run_area` / `update_rib_intra_area` read: Router- and Network-LSAs, the area's
interfaces with their neighbours (router id + source address), plus the recorded `local-rib`."""
from __future__ import annotations

import glob
import json
import os

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def _proto(doc, key):
    for p in doc["ietf-routing:routing"]["control-plane-protocols"]["control-plane-protocol"]:
        if key in p:
            return p[key]
    raise KeyError(key)


def _strip(s):
    return s.split(":", 1)[1] if s.startswith("ietf-ospf:") else s


def ospfv2_vector(rt_dir: str) -> dict:
    cfg_doc = json.load(open(os.path.join(rt_dir, "config.json")))
    st_doc = json.load(open(os.path.join(rt_dir, "output", "northbound-state.json")))
    cfg = _proto(cfg_doc, "ietf-ospf:ospf")
    st = _proto(st_doc, "ietf-ospf:ospf")
    # Interface arena slot (generational_arena::Index, first component of NexthopKey,
    # holo-ospf/src/route.rs:92-98) is RUNTIME state of the recorded run: it is not in config.json
    # nor in the state dump (same config order gives eth-rt4-1 < eth-sw1 in topo2-1..2-3 and the
    # opposite in topo2-4).  Default: name order; where the recorded ECMP routes show another
    # relative order between two interfaces, that observed order is taken as the slot order (it is
    # an INPUT of the path — `iface_idx` — that happens to be visible only through the answer).
    order, iftype, has_vlinks = {}, {}, False
    names = []
    for a in cfg.get("areas", {}).get("area", []):
        for i in a.get("interfaces", {}).get("interface", []):
            names.append(i["name"])
            iftype[i["name"]] = i.get("interface-type", "broadcast")
    names = sorted(set(names))
    before = set()
    for r in st.get("local-rib", {}).get("route", []):
        seq = []
        for n in r.get("next-hops", {}).get("next-hop", []):
            if n.get("outgoing-interface") and (not seq or seq[-1] != n["outgoing-interface"]):
                seq.append(n["outgoing-interface"])
        before |= {(x, y) for x, y in zip(seq, seq[1:])}
    placed = []
    while len(placed) < len(names):
        nxt = next(n for n in names if n not in placed
                   and not any((m, n) in before for m in names if m not in placed and m != n))
        placed.append(nxt)
    order = {n: k for k, n in enumerate(placed)}
    for a in cfg.get("areas", {}).get("area", []):
        if a.get("virtual-links", {}).get("virtual-link"):
            has_vlinks = True
    areas = []
    for a in st.get("areas", {}).get("area", []):
        routers, networks = [], []
        for t in a.get("database", {}).get("area-scope-lsa-type", []):
            for l in t["area-scope-lsas"]["area-scope-lsa"]:
                body = l["ospfv2"]["body"]
                hdr = l["ospfv2"]["header"]
                if "router" in body:
                    links = []
                    for k in body["router"].get("links", {}).get("link", []):
                        links.append({"type": _strip(k["type"]), "id": k["link-id"], "data": k["link-data"],
                                      "metric": int(k["topologies"]["topology"][0]["metric"])})
                    bits = [_strip(b) for b in body["router"].get("router-bits", {}).get("rtr-lsa-bits", [])]
                    routers.append({"adv_rtr": hdr["adv-router"], "lsa_id": hdr["lsa-id"], "bits": bits,
                                    "links": links})
                elif "network" in body:
                    networks.append({"lsa_id": hdr["lsa-id"], "adv_rtr": hdr["adv-router"],
                                     "mask": body["network"]["network-mask"],
                                     "attached": body["network"]["attached-routers"]["attached-router"]})
        ifaces = []
        for i in a.get("interfaces", {}).get("interface", []):
            ifaces.append({"name": i["name"], "type": iftype.get(i["name"], "broadcast"),
                           "index": order.get(i["name"], 1000 + len(ifaces)), "state": i.get("state"),
                           "neighbors": [{"router_id": n["neighbor-router-id"], "src": n["address"]}
                                         for n in i.get("neighbors", {}).get("neighbor", [])]})
        if a.get("virtual-links", {}).get("virtual-link"):
            has_vlinks = True
        areas.append({"area_id": a["area-id"], "routers": routers, "networks": networks, "interfaces": ifaces})
    rib = []
    for r in st.get("local-rib", {}).get("route", []):
        nhs = [[n.get("next-hop"), n.get("outgoing-interface")]
               for n in r.get("next-hops", {}).get("next-hop", [])]
        rib.append({"prefix": r["prefix"], "metric": int(r["metric"]), "type": r["route-type"], "nexthops": nhs})
    return {"source": os.path.relpath(rt_dir, REF), "proto": "ospfv2", "router_id": st["router-id"],
            "max_paths": int(cfg.get("spf-control", {}).get("paths", 16)), "has_vlinks": has_vlinks,
            "areas": areas, "rib": rib}


def make_ospfv2():
    base = os.path.join(REF, "holo-ospf/tests/conformance/ospfv2/topologies")
    out = os.path.join(OUT, "ospfv2")
    os.makedirs(out, exist_ok=True)
    n = 0
    for rt in sorted(glob.glob(os.path.join(base, "topo*", "rt*"))):
        v = ospfv2_vector(rt)
        name = f"{os.path.basename(os.path.dirname(rt))}_{os.path.basename(rt)}.json"
        json.dump(v, open(os.path.join(out, name), "w"), separators=(",", ":"), sort_keys=True)
        n += 1
    print(f"ospfv2: {n} vectors -> {out}")


if __name__ == "__main__":
    make_ospfv2()
