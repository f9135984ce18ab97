"""oracle/ospfv3_ref.py — literal CPU restatement of holo-ospf's OSPFv3 SPF path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pure Python on the LSDB (no CSR); input: a tests/golden/ospfv3 vector (tools/make_golden_ospf.py).
The generic loop is the one of oracle/ospf_ref.py (holo-ospf/src/spf.rs:587-767); this file adds the
version-specific parts:
  holo-ospf/src/ospfv3/spf.rs:38-42     VertexId { Network{router_id, iface_id}, Router{router_id} }
  holo-ospf/src/ospfv3/spf.rs:165-284   Ospfv3::calc_nexthops
  holo-ospf/src/ospfv3/spf.rs:286-346   vertex_lsa_find (router-LSA fragments aggregated, R / V6 bits)
  holo-ospf/src/ospfv3/spf.rs:348-419   vertex_lsa_links
  holo-ospf/src/ospfv3/spf.rs:421-478   intra_area_networks (Intra-Area-Prefix LSAs, NU bit)
  holo-ospf/src/ospfv3/spf.rs:593-612   calc_nexthop_lladdr (neighbour's Link-LSA)
"""
from __future__ import annotations

import ipaddress

from oracle.ospf_ref import NET, RTR, U32_MAX, Vertex, ip, router_route


class AreaDb3:
    def __init__(self, vec: dict, area: dict):
        self.area, self.af = area, vec.get("af", "ipv6")
        frs = {}
        for l in sorted(area["routers"], key=lambda l: (ip(l["adv_rtr"]), l["lsa_id"])):     # iter_by_type_advrtr
            if "r-bit" in l["options"] and (self.af != "ipv6" or "v6-bit" in l["options"]):
                frs.setdefault(ip(l["adv_rtr"]), []).append(l)
        self.routers = frs
        self.networks = {(ip(l["adv_rtr"]), l["lsa_id"]): l for l in area["networks"]}

    def vertex_lsa_find(self, vid):                                  # ospfv3/spf.rs:286-346
        if vid[0] == NET:
            return self.networks.get((vid[1], vid[2]))
        return self.routers.get(vid[1])

    def vertex_lsa_links(self, vid, lsa):                            # ospfv3/spf.rs:348-419
        if vid[0] == NET:
            for r in sorted(ip(a) for a in lsa["attached"]):
                l = self.vertex_lsa_find((RTR, r))
                if l is not None:
                    yield None, (RTR, r), l, 0
            return
        pos = -1
        for frag in lsa:
            for link in frag["links"]:
                pos += 1
                if link["type"] in ("point-to-point-link", "virtual-link"):
                    tid = (RTR, ip(link["nbr_router_id"]))
                else:
                    tid = (NET, ip(link["nbr_router_id"]), link["nbr_iface_id"])
                l = self.vertex_lsa_find(tid)
                if l is not None:
                    yield (pos, link), tid, l, link["metric"]


def _lladdr(iface, nbr_router_id: int, nbr_iface_id: int):          # ospfv3/spf.rs:593-612
    for l in iface["link_lsas"]:
        if ip(l["adv_rtr"]) == nbr_router_id and l["lsa_id"] == nbr_iface_id:
            return l["lladdr"]
    return None


def _akey(a):
    return -1 if a is None else int(ipaddress.ip_address(a))


def calc_nexthops_v3(db: AreaDb3, parent: Vertex, parent_link, dest_id, dest_lsa):
    out = {}
    ifaces = db.area["interfaces"]
    if parent.id[0] == RTR:
        _, plink = parent_link
        iface = next((i for i in ifaces if i["iface_id"] == plink["iface_id"]), None)   # get_by_ifindex
        if iface is None:
            return None
        if iface["type"] == "virtual-link":
            return out
        if dest_id[0] == RTR:
            addr = _lladdr(iface, ip(plink["nbr_router_id"]), plink["nbr_iface_id"])
            if addr is None:
                return None
            out[(iface["index"], _akey(addr))] = (iface["name"], addr)
        else:
            out[(iface["index"], -1)] = (iface["name"], None)
        return out
    # parent = network directly connecting the root to the destination router
    plsa = parent.lsa
    link = next((k for frag in dest_lsa for k in frag["links"]
                 if ip(k["nbr_router_id"]) == ip(plsa["adv_rtr"]) and k["nbr_iface_id"] == plsa["lsa_id"]), None)
    if link is None or not parent.nexthops:
        return None
    first = min(parent.nexthops)
    iface = next(i for i in ifaces if i["index"] == first[0])
    addr = _lladdr(iface, dest_id[1], link["iface_id"])
    if addr is None:
        return None
    out[(first[0], _akey(addr))] = (iface["name"], addr)
    return out


def run_area(vec: dict, area: dict, side: dict = None):
    """`side`: as oracle/ospf_ref.py::run_area — area.state.routers and transit_capability (holo-ospf/src/spf.rs:627-643)."""
    db = AreaDb3(vec, area)
    if side is not None:
        side["routers"], side["transit_capability"] = {}, False
    root_id = (RTR, ip(vec["router_id"]))
    root_lsa = db.vertex_lsa_find(root_id)
    if root_lsa is None:
        return None
    spt, order = {}, []
    cand = {(0, root_id): Vertex(root_id, root_lsa, 0, 0)}
    while cand:
        key = min(cand)
        vertex = cand.pop(key)
        spt[vertex.id] = vertex
        order.append(vertex.id)
        if side is not None and vertex.id[0] == RTR:
            r = router_route(area["area_id"], vertex)
            side["routers"][vertex.id[1]] = r
            if "vlink-end-bit" in r["flags"]:
                side["transit_capability"] = True
        for parent_link, lid, llsa, cost in db.vertex_lsa_links(vertex.id, vertex.lsa):
            if not any(b == vertex.id for _, b, _, _ in db.vertex_lsa_links(lid, llsa)):
                continue
            if lid in spt:
                continue
            distance = min(vertex.distance + cost, U32_MAX)
            hops = min(vertex.hops + (1 if lid[0] == RTR else 0), 0xFFFF)
            ex = next((k for k, c in cand.items() if c.id == lid), None)
            if ex is not None:
                if distance < cand[ex].distance:
                    del cand[ex]
                elif distance > cand[ex].distance:
                    continue
            cv = cand.setdefault((distance, lid), Vertex(lid, llsa, distance, hops))
            nh = calc_nexthops_v3(db, vertex, parent_link, lid, cv.lsa) if vertex.hops == 0 else dict(vertex.nexthops)
            if nh is not None:
                cv.nexthops.update(nh)
    return spt, order


def intra_area_networks(area: dict, spt):                              # ospfv3/spf.rs:421-478
    for lsa in sorted(area["iaps"], key=lambda l: (ip(l["adv_rtr"]), l["lsa_id"])):
        if lsa["ref_type"] == "ospfv3-router-lsa":
            if lsa["ref_lsa_id"] != 0:
                continue
            v = spt.get((RTR, ip(lsa["ref_adv_rtr"])))
        elif lsa["ref_type"] == "ospfv3-network-lsa":
            v = spt.get((NET, ip(lsa["ref_adv_rtr"]), lsa["ref_lsa_id"]))
        else:
            v = None
        if v is None:
            continue
        for p in lsa["prefixes"]:
            if "nu-bit" in p["options"]:
                continue
            yield v, p["prefix"], p["metric"]


def _net_key(p: str):
    n = ipaddress.ip_network(p, strict=False)
    return (n.version, int(n.network_address), n.prefixlen)


def _origin(v: Vertex):
    """vertex.lsa.origin(): LS-ID of the vertex' (first) LSA."""
    return v.lsa["lsa_id"] if v.id[0] == NET else v.lsa[0]["lsa_id"]


def update_rib_intra_area(rib: dict, area: dict, spt, max_paths: int, filter=None):   # holo-ospf/src/route.rs:343-448
    for v, prefix, smetric in intra_area_networks(area, spt):
        key = _net_key(prefix)
        if filter is not None and key not in filter:                     # route.rs:356-362 (partial SPF)
            continue
        metric = min(v.distance + smetric, U32_MAX)
        cur = rib.get(key)
        if cur is not None and metric > cur["metric"]:
            continue
        if v.id[0] == NET and cur is not None:
            if metric < cur["metric"] or (metric == cur["metric"] and _origin(v) > cur["origin"]):
                del rib[key]
            else:
                continue
        new = {"prefix": prefix, "metric": metric, "origin": _origin(v), "nexthops": dict(v.nexthops)}
        cur = rib.get(key)
        if cur is None or new["metric"] < cur["metric"]:
            cur = rib[key] = new
        elif new["metric"] == cur["metric"]:
            cur["nexthops"].update(new["nexthops"])
        if len(cur["nexthops"]) > max_paths:
            cur["nexthops"] = {k: cur["nexthops"][k] for k in sorted(cur["nexthops"])[:max_paths]}


def intra_area_rib(vec: dict):
    rib = {}
    for area in sorted(vec["areas"], key=lambda a: ip(a["area_id"])):
        r = run_area(vec, area)
        if r is not None:
            update_rib_intra_area(rib, area, r[0], vec["max_paths"])
    return [{"prefix": rib[k]["prefix"], "metric": rib[k]["metric"], "type": "intra-area",
             "nexthops": [[rib[k]["nexthops"][n][1], rib[k]["nexthops"][n][0]] for n in sorted(rib[k]["nexthops"])]}
            for k in sorted(rib)]


# ---- SpfComputation::{Full, Partial} (holo-ospf/src/spf.rs:48-60, 489-584) ----------------------------------------------

def spf_computation_type(trigger_lsas):                                  # ospfv3/spf.rs:97-163
    """trigger_lsas: [{"new": lsa, "old": lsa or None}], lsa = {"function": ..., ...}.  ("full", None), or ("partial",
    {"intra": set of prefix keys}) — the inter-area / external members of SpfPartialComputation are outside this path."""
    if any(t["new"]["function"] in ("router", "network", "link", "router-info") for t in trigger_lsas):
        return "full", None
    intra = set()
    for t in trigger_lsas:
        for lsa in (t["new"], t.get("old")):
            if lsa is not None and lsa["function"] == "intra-area-prefix":
                intra.update(_net_key(p["prefix"]) for p in lsa["prefixes"])
    return "partial", {"intra": intra}


def update_rib_partial_intra(rib: dict, intra: set, areas_spts, max_paths: int) -> dict:   # route.rs:200-237, 335-337
    """`rib`: prefix key -> route of the last run; `areas_spts`: [(area, stored SPT)] in area-id order, the SPTs being
    those of the last FULL run (area.state.spt is not touched by a partial run).  Returns the new RIB."""
    partial_rib: dict = {}
    if intra:
        rib = {k: r for k, r in rib.items() if k not in intra}          # extract_if: affected intra-area routes leave the RIB
        for area, spt in areas_spts:                                     # all areas: correct ECMP across areas
            if spt is not None:
                update_rib_intra_area(partial_rib, area, spt, max_paths, intra)
    rib = dict(rib)
    rib.update(partial_rib)                                              # rib.extend(partial_rib)
    return rib


def rows_of(rib: dict):
    return [{"prefix": rib[k]["prefix"], "metric": rib[k]["metric"], "type": "intra-area",
             "nexthops": [[rib[k]["nexthops"][n][1], rib[k]["nexthops"][n][0]] for n in sorted(rib[k]["nexthops"])]}
            for k in sorted(rib)]
