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


def _vlinks(area_state: dict) -> list:
    """The virtual links an area's operational state shows: `cost` is what update_virtual_link copied out of the TRANSIT
    area's router table (`area.state.routers[endpoint].metric`, holo-ospf/src/area.rs:304-333, filled by run_area at
    holo-ospf/src/spf.rs:627-643), and the link is only up when that entry exists and carries the ABR flag — the one
    place where the recorded fixtures expose run_area's router table."""
    return [{"transit_area": v["transit-area-id"], "router_id": v["router-id"], "cost": int(v["cost"]), "state": v.get("state")}
            for v in area_state.get("virtual-links", {}).get("virtual-link", []) if "cost" in v]


def ospfv2_vector(rt_dir: str) -> dict:
    cfg_doc = json.load(open(os.path.join(rt_dir, "config.json")))
    st_doc = json.load(open(os.path.join(rt_dir, "output", "northbound-state.json")))
    cfg = _proto(cfg_doc, "ietf-ospf:ospf")
    st = _proto(st_doc, "ietf-ospf:ospf")
    # Interface arena slot (generational_arena::Index, first component of NexthopKey, holo-ospf/src/route.rs:92-98):
    # see _iface_order below — interface-NAME order, derived from config.json alone, except where the recorded run
    # demonstrably had another one.
    iftype, has_vlinks = {}, False
    for a in cfg.get("areas", {}).get("area", []):
        for i in a.get("interfaces", {}).get("interface", []):
            iftype[i["name"]] = i.get("interface-type", "broadcast")
    order, order_source = _iface_order(cfg, st)
    for a in cfg.get("areas", {}).get("area", []):
        if a.get("virtual-links", {}).get("virtual-link"):
            has_vlinks = True
    areas, vlinks = [], []
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
                                    "links": links, "maxage": "holo-ospf-dev:maxage" in hdr})
                elif "network" in body:
                    networks.append({"lsa_id": hdr["lsa-id"], "adv_rtr": hdr["adv-router"],
                                     "mask": body["network"]["network-mask"],
                                     "attached": body["network"]["attached-routers"]["attached-router"],
                                     "maxage": "holo-ospf-dev:maxage" in hdr})
        ifaces = []
        for i in a.get("interfaces", {}).get("interface", []):
            ifaces.append({"name": i["name"], "type": iftype.get(i["name"], "broadcast"),
                           "index": order.get(i["name"], 1000 + len(ifaces)), "state": i.get("state"),
                           "neighbors": [{"router_id": n["neighbor-router-id"], "src": n["address"]}
                                         for n in i.get("neighbors", {}).get("neighbor", [])]})
        if a.get("virtual-links", {}).get("virtual-link"):
            has_vlinks = True
        vlinks += _vlinks(a)
        areas.append({"area_id": a["area-id"], "routers": routers, "networks": networks, "interfaces": ifaces})
    rib = []
    for r in st.get("local-rib", {}).get("route", []):
        nhs = [[n.get("next-hop"), n.get("outgoing-interface")]
               for n in r.get("next-hops", {}).get("next-hop", [])]
        rib.append({"prefix": r["prefix"], "metric": int(r["metric"]), "type": r["route-type"], "nexthops": nhs})
    out = {"source": os.path.relpath(rt_dir, REF), "proto": "ospfv2", "router_id": st["router-id"],
           "max_paths": int(cfg.get("spf-control", {}).get("paths", 16)), "has_vlinks": has_vlinks,
           "iface_slot_order": order_source, "areas": areas, "rib": rib}
    if vlinks:
        out["vlinks"] = vlinks
    return out


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


# --------------------------------------------------------------------------------------------
# OSPFv3  (fixtures exist although `mod ospfv3` is commented out upstream,
#          holo-ospf/tests/conformance/mod.rs:7-8)
# --------------------------------------------------------------------------------------------

def _iface_order(cfg, st):
    """Interface arena slots (`iface_idx`, the first key of a route's next-hop map): an INPUT of the path that is runtime
    state of the recorded session — neither config.json nor the state dump carries it.  Rule used: interfaces are
    inserted in NAME order (the northbound tree hands the `interface` list entries over sorted by key).  The rule needs
    nothing from the answer, and the recorded ECMP routes CHECK it: over the 107 OSPFv2 / OSPFv3 topology routers they
    show 80 ordered interface pairs, 78 of them in name order.  The two that are not (ospfv2 topo2-4 rt2 / rt3, the
    segment-routing topology: eth-sw1 before eth-rt4-* / eth-rt5-*) contradict config order in topo2-1..2-3 as well,
    i.e. no rule derived from the fixture's inputs reproduces both: for those two routers the observed order is taken
    (`iface_slot_order: "recorded"`) and their ECMP ORDER is not independently pinned; everywhere else it is
    (`"name"`; tests/test_oracle_golden.py::test_ospf_interface_slot_order_rule).  Returns (slot of every name, source)."""
    names = sorted({i["name"] for a in cfg.get("areas", {}).get("area", [])
                    for i in a.get("interfaces", {}).get("interface", [])})
    before = set()
    for r in st.get("local-rib", {}).get("route", []):
        seq = []
        for n in r.get("next-hops", {}).get("next-hop", []):
            if n.get("outgoing-interface") and (not seq or seq[-1] != n["outgoing-interface"]):
                seq.append(n["outgoing-interface"])
        before |= {(x, y) for x, y in zip(seq, seq[1:])}
    by_name = {n: k for k, n in enumerate(names)}
    if all(by_name[x] < by_name[y] for x, y in before if x in by_name and y in by_name):
        return by_name, "name"
    placed = []
    while len(placed) < len(names):
        nxt = next(n for n in names if n not in placed
                   and not any((m, n) in before for m in names if m not in placed and m != n))
        placed.append(nxt)
    return {n: k for k, n in enumerate(placed)}, "recorded"


def ospfv3_vector(rt_dir: str) -> dict:
    cfg_doc = json.load(open(os.path.join(rt_dir, "config.json")))
    st_doc = json.load(open(os.path.join(rt_dir, "output", "northbound-state.json")))
    cfg = _proto(cfg_doc, "ietf-ospf:ospf")
    st = _proto(st_doc, "ietf-ospf:ospf")
    order, order_source = _iface_order(cfg, st)
    iftype, has_vlinks = {}, False
    for a in cfg.get("areas", {}).get("area", []):
        for i in a.get("interfaces", {}).get("interface", []):
            iftype[i["name"]] = i.get("interface-type", "broadcast")
        if a.get("virtual-links", {}).get("virtual-link"):
            has_vlinks = True
    areas, vlinks = [], []
    for a in st.get("areas", {}).get("area", []):
        routers, networks, iaps = [], [], []
        for t in a.get("database", {}).get("area-scope-lsa-type", []):
            for l in t["area-scope-lsas"]["area-scope-lsa"]:
                body, hdr = l["ospfv3"]["body"], l["ospfv3"]["header"]
                if "router" in body:
                    b = body["router"]
                    routers.append({"adv_rtr": hdr["adv-router"], "lsa_id": int(hdr["lsa-id"]),
                                    "options": [_strip(o) for o in b.get("lsa-options", {}).get("lsa-options", [])],
                                    "bits": [_strip(x) for x in b.get("router-bits", {}).get("rtr-lsa-bits", [])],
                                    "links": [{"type": _strip(k["type"]), "iface_id": int(k["interface-id"]),
                                               "nbr_iface_id": int(k["neighbor-interface-id"]),
                                               "nbr_router_id": k["neighbor-router-id"], "metric": int(k["metric"])}
                                              for k in b.get("links", {}).get("link", [])]})
                elif "network" in body:
                    networks.append({"adv_rtr": hdr["adv-router"], "lsa_id": int(hdr["lsa-id"]),
                                     "attached": body["network"]["attached-routers"]["attached-router"]})
                elif "intra-area-prefix" in body:
                    b = body["intra-area-prefix"]
                    iaps.append({"adv_rtr": hdr["adv-router"], "lsa_id": int(hdr["lsa-id"]),
                                 "ref_type": _strip(b["referenced-ls-type"]), "ref_lsa_id": int(b["referenced-link-state-id"]),
                                 "ref_adv_rtr": b["referenced-adv-router"],
                                 "prefixes": [{"prefix": p["prefix"], "metric": int(p.get("metric", 0)),
                                               "options": [_strip(o) for o in p.get("prefix-options", {}).get("prefix-options", [])]}
                                              for p in b.get("prefixes", {}).get("prefix", [])]})
        ifaces = []
        for i in a.get("interfaces", {}).get("interface", []):
            links = []
            for t in i.get("database", {}).get("link-scope-lsa-type", []):
                for l in t["link-scope-lsas"]["link-scope-lsa"]:
                    b = l["ospfv3"]["body"]
                    if "link" in b:
                        links.append({"adv_rtr": l["ospfv3"]["header"]["adv-router"], "lsa_id": int(l["ospfv3"]["header"]["lsa-id"]),
                                      "lladdr": b["link"]["link-local-interface-address"]})
            ifaces.append({"name": i["name"], "type": iftype.get(i["name"], "broadcast"),
                           "index": order.get(i["name"], 1000 + len(ifaces)), "iface_id": int(i.get("interface-id", 0)),
                           "neighbors": [{"router_id": n["neighbor-router-id"], "src": n["address"]}
                                         for n in i.get("neighbors", {}).get("neighbor", [])],
                           "link_lsas": links})
        if a.get("virtual-links", {}).get("virtual-link"):
            has_vlinks = True
        vlinks += _vlinks(a)
        areas.append({"area_id": a["area-id"], "routers": routers, "networks": networks, "iaps": iaps, "interfaces": ifaces})
    rib = []
    for r in st.get("local-rib", {}).get("route", []):
        nhs = [[n.get("next-hop"), n.get("outgoing-interface")] for n in r.get("next-hops", {}).get("next-hop", [])]
        rib.append({"prefix": r["prefix"], "metric": int(r["metric"]), "type": r["route-type"], "nexthops": nhs})
    af = "ipv4" if any("." in r["prefix"].split("/")[0] and ":" not in r["prefix"] for r in rib) else "ipv6"
    out = {"source": os.path.relpath(rt_dir, REF), "proto": "ospfv3", "router_id": st["router-id"], "af": af,
           "max_paths": int(cfg.get("spf-control", {}).get("paths", 16)), "has_vlinks": has_vlinks,
           "iface_slot_order": order_source, "areas": areas, "rib": rib}
    if vlinks:
        out["vlinks"] = vlinks
    return out


def make_ospfv3():
    base = os.path.join(REF, "holo-ospf/tests/conformance/ospfv3/topologies")
    out = os.path.join(OUT, "ospfv3")
    os.makedirs(out, exist_ok=True)
    n = 0
    for rt in sorted(glob.glob(os.path.join(base, "topo*", "rt*"))):
        try:
            v = ospfv3_vector(rt)
        except Exception as e:      # noqa: BLE001
            print("skip", rt, repr(e)[:100])
            continue
        name = f"{os.path.basename(os.path.dirname(rt))}_{os.path.basename(rt)}.json"
        json.dump(v, open(os.path.join(out, name), "w"), separators=(",", ":"), sort_keys=True)
        n += 1
    print(f"ospfv3: {n} vectors -> {out}")


def make_ospfv2_steps():
    """OSPFv2 step tests (holo-ospf/tests/conformance/ospfv2/<name>/): like the IS-IS ones, only the
    steps after which the reference demonstrably re-ran SPF (route (re)installations on the ibus in
    the same step as the last recorded state)."""
    import re
    import shutil
    import tempfile
    base = os.path.join(REF, "holo-ospf/tests/conformance/ospfv2")
    src = open(os.path.join(base, "mod.rs")).read()
    out = os.path.join(OUT, "ospfv2_steps")
    os.makedirs(out, exist_ok=True)
    n = 0
    for name, topo, rt in re.findall(r'run_test::<Instance<Ospfv2>>\(\s*"([^"]+)",\s*"([^"]+)",\s*"([^"]+)"', src):
        d = os.path.join(base, name)
        states = sorted(glob.glob(os.path.join(d, "*-output-northbound-state.json")))
        if not states:
            continue
        ibus = states[-1].replace("-output-northbound-state.json", "-output-ibus.jsonl")
        if not os.path.exists(ibus) or "RouteIp" not in open(ibus).read():
            continue
        if glob.glob(os.path.join(d, "*-input-northbound-config-change.json")):
            continue          # config edits (interface types, areas ...) are not replayed here
        tmp = tempfile.mkdtemp()
        try:
            os.makedirs(os.path.join(tmp, "output"))
            shutil.copy(os.path.join(base, "topologies", topo, rt, "config.json"), os.path.join(tmp, "config.json"))
            shutil.copy(states[-1], os.path.join(tmp, "output", "northbound-state.json"))
            try:
                v = ospfv2_vector(tmp)
            except Exception as e:      # noqa: BLE001
                print("skip", name, repr(e)[:80])
                continue
        finally:
            shutil.rmtree(tmp)
        v["source"] = f"holo-ospf/tests/conformance/ospfv2/{name} (snapshot {topo}/{rt}, state {os.path.basename(states[-1])})"
        # the wire step after the path (update_global_rib, holo-ospf/src/route.rs:856-916): the route
        # messages this step put on the ibus, the whole local RIB before the step, the announced interface indices
        from make_golden import _ibus_routes
        prev = states[-2] if len(states) > 1 else os.path.join(base, "topologies", topo, rt, "output", "northbound-state.json")
        pst = _proto(json.load(open(prev)), "ietf-ospf:ospf")
        v["rib_before"] = [{"prefix": r["prefix"], "metric": int(r["metric"]), "type": r["route-type"],
                            "nexthops": [[n.get("next-hop"), n.get("outgoing-interface")] for n in r.get("next-hops", {}).get("next-hop", [])]}
                           for r in pst.get("local-rib", {}).get("route", [])]
        v["ibus_routes"] = _ibus_routes(ibus)
        ifx = {}
        for evf in [os.path.join(base, "topologies", topo, rt, "events.jsonl")] + sorted(glob.glob(os.path.join(d, "*-input-ibus.jsonl"))):
            for line in open(evf):
                ev = json.loads(line)
                ev = ev.get("Ibus", ev)
                if isinstance(ev, dict) and "InterfaceUpd" in ev:
                    ifx[ev["InterfaceUpd"]["ifname"]] = ev["InterfaceUpd"]["ifindex"]
        v["ifindex"] = ifx
        json.dump(v, open(os.path.join(out, f"{name}.json"), "w"), separators=(",", ":"), sort_keys=True)
        n += 1
    print(f"ospfv2 step tests: {n} vectors -> {out}")
