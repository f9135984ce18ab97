"""holo_amd.ospfv3 — host-side mirror of the OSPFv3 half of holo-ospf's SPF path on the HIP engine.

Version-specific parts only (the generic back half is holo_amd.ospf.spt_from_engine):
  AreaGraph3        vertex_lsa_find / vertex_lsa_links  holo-ospf/src/ospfv3/spf.rs:286-419
                    VertexId { Network{router_id, iface_id}, Router{router_id} }  :38-42
  calc_nexthops     Ospfv3::calc_nexthops + calc_nexthop_lladdr   :165-284, 593-612
  intra_area_networks / update_rib_intra_area   :421-478, holo-ospf/src/route.rs:343-448
"""
from __future__ import annotations

import ipaddress
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .ospf import MAX_PATH_METRIC_OSPF, NET, RTR, VF_NETWORK, Vertex, ip, spt_from_engine


@dataclass
class RouterLink3:
    link_type: str                 # point-to-point-link | transit-network-link | virtual-link
    iface_id: int
    nbr_iface_id: int
    nbr_router_id: str
    metric: int


@dataclass
class RouterLsa3:
    adv_rtr: str
    lsa_id: int
    options: List[str]
    links: List[RouterLink3]
    maxage: bool = False
    bits: List[str] = field(default_factory=list)      # router flags (B / E / V ...): area.state.routers, TransitCapability


@dataclass
class NetworkLsa3:
    adv_rtr: str
    lsa_id: int
    attached: List[str]
    maxage: bool = False


@dataclass
class IntraAreaPrefixLsa:
    adv_rtr: str
    lsa_id: int
    ref_type: str
    ref_lsa_id: int
    ref_adv_rtr: str
    prefixes: List[dict]
    maxage: bool = False


@dataclass
class Interface3:
    name: str
    if_type: str
    index: int
    iface_id: int
    link_lsas: List[dict] = field(default_factory=list)     # link-scope LSDB: adv_rtr, lsa_id, lladdr


@dataclass
class Area3:
    area_id: str
    routers: List[RouterLsa3]
    networks: List[NetworkLsa3]
    iaps: List[IntraAreaPrefixLsa]
    interfaces: List[Interface3]

    @classmethod
    def from_vector(cls, a: dict) -> "Area3":
        return cls(a["area_id"],
                   [RouterLsa3(r["adv_rtr"], r["lsa_id"], r["options"],
                               [RouterLink3(k["type"], k["iface_id"], k["nbr_iface_id"], k["nbr_router_id"], k["metric"])
                                for k in r["links"]], bits=list(r.get("bits", []))) for r in a["routers"]],
                   [NetworkLsa3(n["adv_rtr"], n["lsa_id"], n["attached"]) for n in a["networks"]],
                   [IntraAreaPrefixLsa(p["adv_rtr"], p["lsa_id"], p["ref_type"], p["ref_lsa_id"], p["ref_adv_rtr"], p["prefixes"])
                    for p in a["iaps"]],
                   [Interface3(i["name"], i["type"], i["index"], i["iface_id"], i["link_lsas"]) for i in a["interfaces"]])


class AreaGraph3:
    """CSR of one OSPFv3 area.  A router vertex aggregates all its Router-LSA fragments (ascending
    LS-ID) that carry the R bit (and V6 for the IPv6 address family)."""

    def __init__(self, area: Area3, af: str = "ipv6"):
        self.area = area
        frs: Dict[int, List[RouterLsa3]] = {}
        for l in sorted(area.routers, key=lambda l: (ip(l.adv_rtr), l.lsa_id)):
            if not l.maxage and "r-bit" in l.options and (af != "ipv6" or "v6-bit" in l.options):
                frs.setdefault(ip(l.adv_rtr), []).append(l)
        self.routers = frs
        self.networks = {(ip(l.adv_rtr), l.lsa_id): l for l in area.networks if not l.maxage}
        self.vids = sorted([(NET, k[0], k[1]) for k in self.networks] + [(RTR, k) for k in self.routers])
        self.index = {v: i for i, v in enumerate(self.vids)}
        n = len(self.vids)
        row_ptr = np.zeros(n + 1, np.uint32)
        col, met, self.link_ref = [], [], []
        for i, vid in enumerate(self.vids):
            if vid[0] == NET:
                for r in sorted(ip(a) for a in self.networks[(vid[1], vid[2])].attached):
                    j = self.index.get((RTR, r))
                    if j is not None:
                        col.append(j); met.append(0); self.link_ref.append(None)
            else:
                for frag in self.routers[vid[1]]:
                    for link in frag.links:
                        tid = ((RTR, ip(link.nbr_router_id)) if link.link_type in ("point-to-point-link", "virtual-link")
                               else (NET, ip(link.nbr_router_id), link.nbr_iface_id))
                        j = self.index.get(tid)
                        if j is not None:
                            col.append(j); met.append(link.metric); self.link_ref.append(link)
            row_ptr[i + 1] = len(col)
        self.row_ptr = row_ptr
        self.col = np.asarray(col, np.uint32)
        self.metric = np.asarray(met, np.uint32)
        self.vflags = np.asarray([VF_NETWORK if v[0] == NET else 0 for v in self.vids], np.uint8)
        self._dev = None

    def lsa_of(self, v: int):
        vid = self.vids[v]
        return self.networks[(vid[1], vid[2])] if vid[0] == NET else self.routers[vid[1]]

    def device(self, engine):
        if self._dev is None or self._dev[0] is not engine:
            self._dev = (engine, engine.upload(self.row_ptr, self.col, self.metric, self.vflags, MAX_PATH_METRIC_OSPF))
        return self._dev[1]


def _lladdr(iface: Interface3, nbr_router_id: int, nbr_iface_id: int) -> Optional[str]:
    for l in iface.link_lsas:
        if ip(l["adv_rtr"]) == nbr_router_id and l["lsa_id"] == nbr_iface_id:
            return l["lladdr"]
    return None


def _akey(a: Optional[str]) -> int:
    return -1 if a is None else int(ipaddress.ip_address(a))


def calc_nexthops(g: AreaGraph3, parent: Vertex, k: int, dest: tuple, dest_lsa) -> Optional[dict]:
    """Ospfv3::calc_nexthops for a hops == 0 parent and CSR entry k (ospfv3/spf.rs:165-284)."""
    out: Dict[tuple, tuple] = {}
    if parent.id[0] == RTR:
        plink = g.link_ref[k]
        iface = next((i for i in g.area.interfaces if i.iface_id == plink.iface_id), None)
        if iface is None:
            return None
        if iface.if_type == "virtual-link":
            return out
        if dest[0] == RTR:
            addr = _lladdr(iface, ip(plink.nbr_router_id), plink.nbr_iface_id)
            if addr is None:
                return None
            out[(iface.index, _akey(addr))] = (iface.name, addr)
        else:
            out[(iface.index, -1)] = (iface.name, None)
        return out
    plsa = parent.lsa
    link = next((l for frag in dest_lsa for l in frag.links
                 if ip(l.nbr_router_id) == ip(plsa.adv_rtr) and l.nbr_iface_id == plsa.lsa_id), None)
    if link is None or not parent.nexthops:
        return None
    first = min(parent.nexthops)
    iface = next(i for i in g.area.interfaces if i.index == first[0])
    addr = _lladdr(iface, dest[1], link.iface_id)
    if addr is None:
        return None
    out[(first[0], _akey(addr))] = (iface.name, addr)
    return out


def run_area(router_id: str, area: Area3, engine, af: str = "ipv6", graph: Optional[AreaGraph3] = None):
    """holo-ospf/src/spf.rs:587-729 for V = Ospfv3."""
    g = graph or AreaGraph3(area, af)
    root = g.index.get((RTR, ip(router_id)))
    if root is None:
        return None
    return spt_from_engine(g, root, engine, calc_nexthops)


def _net_key(p: str):
    n = ipaddress.ip_network(p, strict=False)
    return (n.version, int(n.network_address), n.prefixlen)


def update_rib_intra_area(rib: dict, area: Area3, spt: Dict[tuple, Vertex], max_paths: int, filter=None):
    """intra_area_networks (ospfv3/spf.rs:421-478) + update_rib_intra_area (route.rs:343-448); `filter` = the prefix
    keys of a partial run (route.rs:356-362)."""
    for lsa in sorted(area.iaps, key=lambda l: (ip(l.adv_rtr), l.lsa_id)):
        if lsa.maxage:
            continue
        if lsa.ref_type == "ospfv3-router-lsa":
            v = spt.get((RTR, ip(lsa.ref_adv_rtr))) if lsa.ref_lsa_id == 0 else None
        elif lsa.ref_type == "ospfv3-network-lsa":
            v = spt.get((NET, ip(lsa.ref_adv_rtr), lsa.ref_lsa_id))
        else:
            v = None
        if v is None:
            continue
        origin = v.lsa.lsa_id if v.id[0] == NET else v.lsa[0].lsa_id
        for p in lsa.prefixes:
            if "nu-bit" in p["options"]:
                continue
            key = _net_key(p["prefix"])
            if filter is not None and key not in filter:
                continue
            metric = min(v.distance + p["metric"], 0xFFFFFFFF)
            cur = rib.get(key)
            if cur is not None and metric > cur["metric"]:
                continue
            if v.id[0] == NET and cur is not None:
                if metric < cur["metric"] or (metric == cur["metric"] and origin > cur["origin"]):
                    del rib[key]
                else:
                    continue
            new = {"prefix": p["prefix"], "metric": metric, "origin": origin, "nexthops": dict(v.nexthops)}
            cur = rib.get(key)
            if cur is None or new["metric"] < cur["metric"]:
                cur = rib[key] = new
            elif new["metric"] == cur["metric"]:
                cur["nexthops"].update(new["nexthops"])
            if len(cur["nexthops"]) > max_paths:
                cur["nexthops"] = {k: cur["nexthops"][k] for k in sorted(cur["nexthops"])[:max_paths]}


def compute_spf_intra_area(router_id: str, areas: List[Area3], max_paths: int, engine, af: str = "ipv6") -> List[dict]:
    rib: dict = {}
    for area in sorted(areas, key=lambda a: ip(a.area_id)):
        spt = run_area(router_id, area, engine, af)
        if spt is not None:
            update_rib_intra_area(rib, area, spt, max_paths)
    return [{"prefix": rib[k]["prefix"], "metric": rib[k]["metric"], "type": "intra-area",
             "nexthops": [[rib[k]["nexthops"][n][1], rib[k]["nexthops"][n][0]] for n in sorted(rib[k]["nexthops"])]}
            for k in sorted(rib)]


# ---- SpfComputation::{Full, Partial} (holo-ospf/src/spf.rs:48-60, 489-584) ----------------------------------------------

FULL_FUNCTIONS_V3 = ("router", "network", "link", "router-info")                     # ospfv3/spf.rs:106-118
FULL_FUNCTIONS_V2 = ("router", "network", "opaque-area-router-info", "opaque-area-ext-prefix", "opaque-area-ext-link",
                     "opaque-as-ext-prefix")                                         # ospfv2/spf.rs:104-119


def spf_computation_type(trigger_lsas, version: int = 3):
    """`V::spf_computation_type`: which changed LSAs need the SPT again.  trigger_lsas: [{"new": {"function": ...,
    "prefixes": [...]}, "old": ... or None}].  Returns ("full", None) or ("partial", {"intra": set of prefix keys}):
    a partial run does NOT touch the SPT — the engine is not called — and, for OSPFv3, re-attaches the prefixes of the
    changed Intra-Area-Prefix-LSAs (old and new version) to the stored SPTs; in OSPFv2 the intra-area information lives
    in Router- / Network-LSAs, so `intra` is always empty there (ospfv2/spf.rs:121-123).  The inter-area / external
    members of SpfPartialComputation belong to route calculations outside this path."""
    full = FULL_FUNCTIONS_V3 if version == 3 else FULL_FUNCTIONS_V2
    if any(t["new"]["function"] in full for t in trigger_lsas):
        return "full", None
    intra = set()
    if version == 3:
        for t in trigger_lsas:
            for lsa in (t["new"], t.get("old")):
                if lsa is not None and lsa["function"] == "intra-area-prefix":
                    intra.update(_net_key(p["prefix"]) for p in lsa["prefixes"])
    return "partial", {"intra": intra}


class SpfState:
    """What compute_spf keeps between runs (holo-ospf/src/spf.rs:489-584): the per-area SPTs of the last FULL run
    (`area.state.spt`) and the intra-area RIB.  `run` dispatches like the reference: a Full computation runs every area
    on the engine and rebuilds the RIB (route::update_rib_full); a Partial one (route::update_rib_partial, route.rs:
    200-237) removes the affected prefixes, re-attaches them from the STORED SPTs over all areas and never calls the
    engine."""

    def __init__(self, router_id: str, max_paths: int, engine, af: str = "ipv6"):
        self.router_id, self.max_paths, self.engine, self.af = router_id, max_paths, engine, af
        self.spts: Dict[str, Optional[Dict[tuple, Vertex]]] = {}
        self.rib: dict = {}
        self.engine_runs = 0

    def run(self, areas: List[Area3], trigger_lsas=None) -> List[dict]:
        kind, partial = ("full", None) if trigger_lsas is None else spf_computation_type(trigger_lsas, 3)
        ordered = sorted(areas, key=lambda a: ip(a.area_id))
        if kind == "full":
            self.rib = {}
            for area in ordered:
                spt = run_area(self.router_id, area, self.engine, self.af)
                self.engine_runs += 1
                if spt is None:
                    # root LSA missing: run_area returns before it touches area.state.spt (holo-ospf/src/spf.rs:596-620) and
                    # update_rib_full still folds the area from that SPT (route.rs:157-160)
                    spt = self.spts.get(area.area_id)
                self.spts[area.area_id] = spt
                if spt is not None:
                    update_rib_intra_area(self.rib, area, spt, self.max_paths)
        elif partial["intra"]:
            intra = partial["intra"]
            rib = {k: r for k, r in self.rib.items() if k not in intra}
            part: dict = {}
            for area in ordered:
                spt = self.spts.get(area.area_id)
                if spt is not None:
                    update_rib_intra_area(part, area, spt, self.max_paths, intra)
            rib.update(part)
            self.rib = rib
        rib = self.rib
        return [{"prefix": rib[k]["prefix"], "metric": rib[k]["metric"], "type": "intra-area",
                 "nexthops": [[rib[k]["nexthops"][n][1], rib[k]["nexthops"][n][0]] for n in sorted(rib[k]["nexthops"])]}
                for k in sorted(rib)]
