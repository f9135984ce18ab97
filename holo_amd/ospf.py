"""holo_amd.ospf — host-side mirror of holo-ospf's OSPFv2 SPF path on top of the HIP engine.

  run_area()               holo-ospf/src/spf.rs:587-729        one SPT per area, root = self
  calc_nexthops()          holo-ospf/src/spf.rs:733-767 + holo-ospf/src/ospfv2/spf.rs:172-353
  intra_area_networks()    holo-ospf/src/ospfv2/spf.rs:462-538
  update_rib_intra_area()  holo-ospf/src/route.rs:343-448 (+ route_update :918-965)

The LSA walk (`vertex_lsa_find` / `vertex_lsa_links`, ospfv2/spf.rs:355-460) is done once per LSDB
generation into the CSR of include/holo_spf_hip.h (AreaGraph); the SPT loop runs on the GPU
(HSPF_RUN_NET_NEXTHOPS semantics: a network reached from a hops == 0 parent owns a first-hop
slot); `calc_nexthops`, which needs Interface / Neighbor objects, is evaluated here once per
first-hop slot and OR-ed through the per-vertex slot masks.  No CPU SPT loop lives here.
"""
from __future__ import annotations

import ipaddress
from dataclasses import dataclass, field
from typing import Iterable, Dict, List, Optional, Tuple

import numpy as np

from . import engine as E

NET, RTR = 0, 1                       # enum VertexId { Network, Router } derive(Ord), ospfv2/spf.rs:41-45
VF_NETWORK = 1
MAX_PATH_METRIC_OSPF = 0xFFFFFFFF     # u32 saturating add, holo-ospf/src/spf.rs:672

VertexId = Tuple[int, int]


def ip(s: str) -> int:
    return int(ipaddress.IPv4Address(s))


@dataclass
class RouterLink:                     # LsaRouterLink
    link_type: str                    # point-to-point-link | transit-network-link | stub-network-link | virtual-link
    link_id: str
    link_data: str
    metric: int


@dataclass
class RouterLsa:
    adv_rtr: str
    links: List[RouterLink]
    bits: List[str] = field(default_factory=list)
    maxage: bool = False


@dataclass
class NetworkLsa:
    lsa_id: str
    adv_rtr: str
    mask: str
    attached: List[str]
    maxage: bool = False


@dataclass
class Neighbor:
    router_id: str
    src: str


@dataclass
class Interface:
    name: str
    if_type: str = "broadcast"        # point-to-point | broadcast | point-to-multipoint | virtual-link
    index: int = 0                    # arena slot: first component of NexthopKey (route.rs:92-98)
    neighbors: List[Neighbor] = field(default_factory=list)
    addrs: List[str] = field(default_factory=list)


@dataclass
class Area:
    area_id: str
    routers: List[RouterLsa]
    networks: List[NetworkLsa]
    interfaces: List[Interface]

    @classmethod
    def from_vector(cls, a: dict) -> "Area":
        return cls(a["area_id"],
                   [RouterLsa(r["adv_rtr"], [RouterLink(k["type"], k["id"], k["data"], k["metric"]) for k in r["links"]],
                              r.get("bits", []), bool(r.get("maxage"))) for r in a["routers"]],
                   [NetworkLsa(n["lsa_id"], n["adv_rtr"], n["mask"], n["attached"], bool(n.get("maxage"))) for n in a["networks"]],
                   [Interface(i["name"], i["type"], i["index"], [Neighbor(n["router_id"], n["src"]) for n in i["neighbors"]],
                              i.get("addrs", [])) for i in a["interfaces"]])


@dataclass
class Vertex:                          # holo-ospf/src/spf.rs:38-46
    id: VertexId
    lsa: object
    distance: int
    hops: int
    nexthops: Dict[tuple, tuple] = field(default_factory=dict)   # (iface idx, addr|-1) -> (iface name, addr|None)


class AreaGraph:
    """CSR of one area's Router-/Network-LSAs.  Vertex index = rank in VertexId order (all networks,
    then all routers, numeric).  link_pos / link_ref keep, per CSR entry, what Ospfv2::calc_nexthops
    needs from `SpfLink.parent` (the position among the non-stub links BEFORE the existence filter,
    ospfv2/spf.rs:439-456, and the link itself)."""

    def __init__(self, area: Area):
        self.area = area
        self.routers, self.networks = self._live_lsas(area)
        self.vids: List[VertexId] = sorted([(NET, k) for k in self.networks] + [(RTR, k) for k in self.routers])
        self.index = {v: i for i, v in enumerate(self.vids)}
        n = len(self.vids)
        row_ptr = np.zeros(n + 1, np.uint32)
        col, met, self.link_pos, self.link_ref = [], [], [], []
        for i, vid in enumerate(self.vids):
            c, m, pos, ref = self._row(vid)
            col += c; met += m; self.link_pos += pos; self.link_ref += ref
            row_ptr[i + 1] = len(col)
        self.row_ptr = row_ptr
        self.col = np.asarray(col, np.uint32)
        self.metric = np.asarray(met, np.uint32)
        self.vflags = np.asarray([VF_NETWORK if v[0] == NET else 0 for v in self.vids], np.uint8)
        self._dev = None

    @staticmethod
    def _live_lsas(area: Area):
        routers: Dict[int, RouterLsa] = {ip(l.adv_rtr): l for l in area.routers if not l.maxage}
        nets: Dict[int, NetworkLsa] = {}
        # vertex_lsa_find for a network: FIRST Network-LSA in (adv_rtr, lsa_id) order whose LS-ID
        # matches, then dropped if MaxAge (ospfv2/spf.rs:362-373)
        for l in sorted(area.networks, key=lambda l: (ip(l.adv_rtr), ip(l.lsa_id))):
            nets.setdefault(ip(l.lsa_id), l)
        return routers, {k: l for k, l in nets.items() if not l.maxage}

    def _row(self, vid: VertexId):
        """vertex_lsa_links (ospfv2/spf.rs:389-460) of one vertex against the current vertex set."""
        col, met, lpos, ref = [], [], [], []
        if vid[0] == NET:
            for r in sorted(ip(a) for a in self.networks[vid[1]].attached):
                j = self.index.get((RTR, r))
                if j is not None:
                    col.append(j); met.append(0); lpos.append(-1); ref.append(None)
            return col, met, lpos, ref
        pos = -1
        for link in self.routers[vid[1]].links:
            if link.link_type in ("point-to-point-link", "virtual-link"):
                tid = (RTR, ip(link.link_id))
            elif link.link_type == "transit-network-link":
                tid = (NET, ip(link.link_id))
            else:
                continue
            pos += 1
            j = self.index.get(tid)
            if j is not None:
                col.append(j); met.append(link.metric); lpos.append(pos); ref.append(link)
        return col, met, lpos, ref

    def refresh(self, area: Area, changed: Iterable[VertexId]) -> bool:
        """Bring the graph forward to `area` (a later state of the same area's LSDB) when only the LSAs of the
        `changed` vertices were re-originated (the reference's SpfTriggerLsa list, holo-ospf/src/spf.rs:120-139):
        their rows are rebuilt and replaced on the device with hspf_graph_patch.  False (nothing touched) when a
        vertex appeared or vanished (new LSA, MaxAge): the caller builds a new AreaGraph."""
        routers, networks = self._live_lsas(area)
        if set(routers) != set(self.routers) or set(networks) != set(self.networks):
            return False
        self.area, self.routers, self.networks = area, routers, networks
        vs = sorted({self.index[v] for v in changed if v in self.index})
        if not vs:
            return True
        rows, aux = [], {}
        for i in vs:
            c, m, lpos, ref = self._row(self.vids[i])
            rows.append((np.asarray(c, np.uint32), np.asarray(m, np.uint32)))
            aux[i] = (lpos, ref)
        # per-entry side tables of calc_nexthops, spliced like the CSR
        npos, nref = [], []
        for u in range(len(self.vids)):
            if u in aux:
                npos += aux[u][0]; nref += aux[u][1]
            else:
                a, b = int(self.row_ptr[u]), int(self.row_ptr[u + 1])
                npos += self.link_pos[a:b]; nref += self.link_ref[a:b]
        flags = [int(self.vflags[i]) for i in vs]
        if self._dev is not None:
            self._dev[1].patch(vs, rows, flags)
        self.row_ptr, self.col, self.metric, self.vflags = E.splice_rows(
            self.row_ptr, self.col, self.metric, self.vflags, vs, [r[0] for r in rows], [r[1] for r in rows],
            np.asarray(flags, np.uint8))
        self.link_pos, self.link_ref = npos, nref
        return True

    def lsa_of(self, v: int):
        vid = self.vids[v]
        return self.networks[vid[1]] if vid[0] == NET else self.routers[vid[1]]

    def device(self, engine):
        if self._dev is None or self._dev[0] is not engine:
            self._dev = (engine, engine.upload(self.row_ptr, self.col, self.metric, self.vflags, MAX_PATH_METRIC_OSPF))
        return self._dev[1]


def calc_nexthops(g: AreaGraph, parent: Vertex, k: int, dest: VertexId, dest_lsa) -> Optional[dict]:
    """Ospfv2::calc_nexthops for a hops == 0 parent and CSR entry k (ospfv2/spf.rs:172-353).
    None = Err(SpfNexthopCalcError), which the reference logs and skips (spf.rs:717-718)."""
    out: Dict[tuple, tuple] = {}
    if parent.id[0] == RTR:
        pos = g.link_pos[k]
        cands = [i for i in sorted(g.area.interfaces, key=lambda i: i.name) if len(i.neighbors) > 0]
        if pos >= len(cands):
            return None
        iface = cands[pos]
        if iface.if_type == "virtual-link":
            return out
        if dest[0] == RTR:
            if iface.if_type in ("point-to-point", "virtual-link"):
                nbr = next((n for n in iface.neighbors if ip(n.router_id) == dest[1]), None)
                if nbr is None:
                    return None
                out[(iface.index, ip(nbr.src))] = (iface.name, nbr.src)
            elif iface.if_type == "point-to-multipoint":
                for link in dest_lsa.links:
                    if any(ipaddress.IPv4Address(link.link_data) in ipaddress.ip_network(a, strict=False)
                           for a in iface.addrs):
                        out[(iface.index, ip(link.link_data))] = (iface.name, link.link_data)
            if not out:
                return None
        else:
            out[(iface.index, -1)] = (iface.name, None)
        return out
    try:
        net = ipaddress.ip_network((parent.lsa.lsa_id, parent.lsa.mask), strict=False)
    except ValueError:
        return None
    link = next((l for l in dest_lsa.links if ipaddress.IPv4Address(l.link_data) in net), None)
    if link is None or not parent.nexthops:
        return None
    first_key = min(parent.nexthops)
    out[(first_key[0], ip(link.link_data))] = (parent.nexthops[first_key][0], link.link_data)
    return out


def spt_from_engine(g, root: int, engine, calc, res=None, slots_out: Optional[dict] = None) -> Dict[tuple, Vertex]:
    """Version-generic back half of run_area: one engine run (HSPF_RUN_NET_NEXTHOPS), then every
    first-hop slot is expanded ONCE through the version's `calc(g, parent_vertex, k, dest_vid,
    dest_lsa)` (= V::calc_nexthops for a hops == 0 parent, holo-ospf/src/spf.rs:747-760) and the
    per-slot sets are OR-ed through the per-vertex masks (= the inheritance of :761-766)."""
    G = g.device(engine)
    if res is None:                       # `res`: tables of a run the caller already made (device-resident path)
        res = engine.run(G, np.asarray([root], np.uint32), E.RUN_NET_NEXTHOPS)
    dist, hops = res.dist[0], res.hops[0]
    in_spt = (res.flags[0] & E.RF_IN_SPT) != 0
    mask = res.first_hop_mask[0]
    hv, hb, _tot = G.slot_table(root)
    hv, hb = [int(x) for x in hv], [int(x) for x in hb]
    spt: Dict[tuple, Vertex] = {}
    slot_cache: Dict[int, Optional[dict]] = {}

    def slots_of(v: int):
        for w in range(mask.shape[1]):
            m = int(mask[v, w])
            while m:
                b = (m & -m).bit_length() - 1
                m &= m - 1
                yield w * 64 + b

    def vertex(v: int) -> Vertex:
        vid = g.vids[v]
        vx = spt.get(vid)
        if vx is not None:
            return vx
        vx = Vertex(vid, g.lsa_of(v), int(dist[v]), int(hops[v]))
        for s in slots_of(v):
            nh = resolve_slot(s)
            if nh:
                vx.nexthops.update(nh)
        spt[vid] = vx
        return vx

    def resolve_slot(s: int) -> Optional[dict]:
        if s in slot_cache:
            return slot_cache[s]
        i = int(np.searchsorted(hb, s, side="right")) - 1
        p, k = hv[i], int(g.row_ptr[hv[i]]) + (s - hb[i])
        t = int(g.col[k])
        slot_cache[s] = calc(g, vertex(p), k, g.vids[t], g.lsa_of(t))
        return slot_cache[s]

    # distance order guarantees parents (hops == 0 networks) are materialised before children
    for v in sorted(np.nonzero(in_spt)[0].tolist(), key=lambda v: (int(dist[v]), v)):
        vertex(v)
    if slots_out is not None:             # slot -> next hops, for consumers of masks that are not vertices
        slots_out.update(slot_cache)
    return spt


def run_area(router_id: str, area: Area, engine, graph: Optional[AreaGraph] = None):
    """holo-ospf/src/spf.rs:587-729 -> dict VertexId -> Vertex (the area's SPT), or None when the
    root's Router-LSA is missing (Error::SpfRootNotFound is logged and the run returns, :605-610)."""
    g = graph or AreaGraph(area)
    root = g.index.get((RTR, ip(router_id)))
    if root is None:
        return None
    return spt_from_engine(g, root, engine, calc_nexthops)


def routers_table(area_id: str, spt: Dict[tuple, Vertex]):
    """The two side outputs of run_area's loop (holo-ospf/src/spf.rs:627-643), from the SPT the engine gave: the area's
    "router" routing table — RouteRtr{area, IntraArea, options, flags, distance, next hops} per router vertex, what
    area::update_virtual_link (area.rs:304-333), the ASBR lookups of the external route calculation and the Type-4
    summaries read — and TransitCapability (some router of the SPT is a virtual-link endpoint).  A vertex's next hops are
    final when it is popped, so the table built from the finished SPT is the one the loop builds pop by pop.  Version
    generic: an OSPFv3 router vertex is its list of Router-LSA fragments, the FIRST one gives flags and options
    (ospfv3/spf.rs:70-80).  Returns (dict router id -> route dict, transit_capability)."""
    routers, transit = {}, False
    for vid, v in spt.items():
        if vid[0] != RTR:
            continue
        lsa = v.lsa[0] if isinstance(v.lsa, list) else v.lsa
        flags = sorted(getattr(lsa, "bits", []))
        routers[vid[1]] = {"area_id": area_id, "path_type": "intra-area", "options": sorted(getattr(lsa, "options", [])),
                           "flags": flags, "metric": v.distance, "nexthops": dict(v.nexthops)}
        transit = transit or "vlink-end-bit" in flags
    return routers, transit


def intra_area_networks(spt: Dict[VertexId, Vertex]):          # ospfv2/spf.rs:462-538
    for vid in sorted(spt):
        v = spt[vid]
        if vid[0] == NET:
            try:
                n = ipaddress.ip_network((v.lsa.lsa_id, v.lsa.mask), strict=False)
            except ValueError:
                continue
            yield v, str(n), 0, ip(v.lsa.lsa_id)
        else:
            for link in v.lsa.links:
                if link.link_type != "stub-network-link":
                    continue
                try:
                    n = ipaddress.ip_network((link.link_id, link.link_data), strict=False)
                except ValueError:
                    continue
                yield v, str(n), link.metric, ip(v.lsa.adv_rtr)


def _net_key(p: str):
    n = ipaddress.ip_network(p, strict=False)
    return (int(n.network_address), n.prefixlen)


def update_rib_intra_area(rib: dict, spt: Dict[VertexId, Vertex], max_paths: int):   # route.rs:343-448
    for v, prefix, smetric, origin_id in intra_area_networks(spt):
        key = _net_key(prefix)
        metric = min(v.distance + smetric, 0xFFFFFFFF)
        cur = rib.get(key)
        if cur is not None and metric > cur["metric"]:
            continue
        if v.id[0] == NET and cur is not None:
            if metric < cur["metric"] or (metric == cur["metric"] and origin_id > cur["origin"]):
                del rib[key]
            else:
                continue
        new = {"prefix": prefix, "metric": metric, "origin": origin_id, "connected": v.hops == 0,
               "nexthops": dict(v.nexthops)}
        cur = rib.get(key)                                        # route_update, route.rs:918-965
        if cur is None or new["metric"] < cur["metric"]:
            cur = rib[key] = new
        elif new["metric"] == cur["metric"]:
            cur["nexthops"].update(new["nexthops"])
        if len(cur["nexthops"]) > max_paths:
            cur["nexthops"] = {k: cur["nexthops"][k] for k in sorted(cur["nexthops"])[:max_paths]}


def changed_vertex_ids(old: Area, new: Area) -> List[VertexId]:
    """Vertices whose Router-/Network-LSA differs between two states of an area (the SpfTriggerLsa list the
    reference accumulates as LSAs are installed, holo-ospf/src/spf.rs:120-139)."""
    ra, rb = ({ip(l.adv_rtr): l for l in a.routers} for a in (old, new))
    na, nb = ({(ip(l.adv_rtr), ip(l.lsa_id)): l for l in a.networks} for a in (old, new))
    out = {(RTR, k) for k in set(ra) | set(rb) if ra.get(k) != rb.get(k)}
    out |= {(NET, k[1]) for k in set(na) | set(nb) if na.get(k) != nb.get(k)}
    return sorted(out)


class GraphCache:
    """Area graphs kept on the device across SPF runs and patched from the changed LSAs."""

    def __init__(self):
        self.graphs: Dict[str, AreaGraph] = {}
        self.rebuilt = 0
        self.patched = 0

    def get(self, area: Area, trigger: Optional[Iterable[VertexId]] = None) -> AreaGraph:
        g = self.graphs.get(area.area_id)
        if g is not None and trigger is not None and g.refresh(area, trigger):
            self.patched += 1
            return g
        if g is not None and g._dev is not None:
            g._dev[1].free()
        g = self.graphs[area.area_id] = AreaGraph(area)
        self.rebuilt += 1
        return g


def compute_spf_intra_area(router_id: str, areas: List[Area], max_paths: int, engine,
                           cache: Optional[GraphCache] = None,
                           trigger: Optional[Dict[str, Iterable[VertexId]]] = None) -> List[dict]:
    """The SPT + intra-area part of compute_spf (holo-ospf/src/spf.rs:489-584, route.rs:146-160):
    areas in area-id order, one run_area each; rows like the YANG `local-rib` list.  With a GraphCache the area
    graphs persist on the device and `trigger[area_id]` (changed vertices) turns LSDB -> CSR into row patches."""
    rib: dict = {}
    for area in sorted(areas, key=lambda a: ip(a.area_id)):
        graph = None
        if cache is not None:
            graph = cache.get(area, None if trigger is None else trigger.get(area.area_id, ()))
        spt = run_area(router_id, area, engine, graph)
        if spt is not None:
            update_rib_intra_area(rib, spt, max_paths)
    rows = []
    for key in sorted(rib):
        r = rib[key]
        rows.append({"prefix": r["prefix"], "metric": r["metric"], "type": "intra-area",
                     "nexthops": [[r["nexthops"][k][1], r["nexthops"][k][0]] for k in sorted(r["nexthops"])]})
    return rows


class SpfState:
    """What compute_spf keeps between runs for OSPFv2 (holo-ospf/src/spf.rs:489-584): per area the SPT of the last FULL
    run (`area.state.spt`), its "router" routing table and TransitCapability (`area.state.routers`,
    `area.state.transit_capability`, :627-643), and the intra-area RIB.  `run` dispatches like the reference
    (`Ospfv2::spf_computation_type`, ospfv2/spf.rs:99-170): a change of a Router- / Network-LSA (or of the SR opaque
    LSAs) is a Full computation — every area goes through the engine —; Type-3 / Type-4 / Type-5 changes are Partial
    ones whose intra-area part is EMPTY in OSPFv2 (:124-126): the engine is not called, SPTs, router tables and the
    intra-area RIB stay as they are (what the partial run recomputes — inter-area and external routes — reads
    `routers` and the SPTs, and lies outside this path)."""

    def __init__(self, router_id: str, max_paths: int, engine):
        from . import ospfv3 as _v3                      # the version-generic classification lives with SpfState of OSPFv3
        self._classify = _v3.spf_computation_type
        self.router_id, self.max_paths, self.engine = router_id, max_paths, engine
        self.cache = GraphCache()
        self.spts: Dict[str, Optional[Dict[tuple, Vertex]]] = {}
        self.routers: Dict[str, dict] = {}
        self.transit_capability: Dict[str, bool] = {}
        self.rows: List[dict] = []
        self.engine_runs = 0

    def run(self, areas: List[Area], trigger_lsas=None, trigger_vertices: Optional[Dict[str, Iterable[VertexId]]] = None) -> List[dict]:
        kind, _partial = ("full", None) if trigger_lsas is None else self._classify(trigger_lsas, 2)
        if kind != "full":
            return self.rows
        rib: dict = {}
        for area in sorted(areas, key=lambda a: ip(a.area_id)):
            graph = self.cache.get(area, None if trigger_vertices is None else trigger_vertices.get(area.area_id, ()))
            spt = run_area(self.router_id, area, self.engine, graph)
            self.engine_runs += 1
            if spt is None:
                # root LSA missing: run_area logs SpfRootNotFound and returns BEFORE `routers.clear()` and without touching
                # `area.state.spt` (holo-ospf/src/spf.rs:596-620) — only transit_capability has been reset (:598); the
                # router table and the SPT of the area keep what the previous run left
                self.transit_capability[area.area_id] = False
                self.spts.setdefault(area.area_id, None)
                self.routers.setdefault(area.area_id, {})
                # update_rib_full folds EVERY area (holo-ospf/src/route.rs:157-160), this one from the SPT it still holds
                # (Ospfv2::intra_area_networks walks area.state.spt, ospfv2/spf.rs:462-469): its routes stay
                if self.spts[area.area_id] is not None:
                    update_rib_intra_area(rib, self.spts[area.area_id], self.max_paths)
                continue
            self.spts[area.area_id] = spt
            self.routers[area.area_id], self.transit_capability[area.area_id] = routers_table(area.area_id, spt)
            update_rib_intra_area(rib, spt, self.max_paths)
        self.rows = [{"prefix": rib[k]["prefix"], "metric": rib[k]["metric"], "type": "intra-area",
                      "nexthops": [[rib[k]["nexthops"][n][1], rib[k]["nexthops"][n][0]] for n in sorted(rib[k]["nexthops"])]}
                     for k in sorted(rib)]
        return self.rows


# ---- the wire step after the path: update_global_rib --------------------------------------------------

def update_global_rib(new_rows: List[dict], old_rows: List[dict], ifindex: Dict[str, int]) -> List[dict]:
    """holo-ospf/src/route.rs:856-916 + ibus::tx::route_install / route_uninstall (holo-ospf/src/ibus/tx.rs:32-77) on rows
    shaped like the YANG `local-rib` list (compute_spf_intra_area's rows, plus whatever inter-area / external rows the
    calculations outside this path produced: the comparison does not look at the route type).  New RIB in
    BTreeMap<IpNetwork, _> order: the prefix leaves the old RIB; equal metric and equal next-hop set -> nothing to send
    (:875-885); otherwise a RouteIpAdd unless the route is CONNECTED or has no next hops (:887-901) — a route is
    CONNECTED iff none of its next hops carries an address (the vertex is a hops-0 network: `Ospfv2::calc_nexthops`
    yields (iface, None), ospfv2/spf.rs:296-302) —, next hops as the BTreeSet of Nexthop::Address { ifindex, addr } the
    message carries; then a RouteIpDel for every installed route the old RIB still holds (:908-914).  Messages in emission
    order: what the reference recorded on the ibus for its step tests (tests/test_host_ospf.py)."""
    import ipaddress

    def installed(r) -> bool:
        return any(a is not None for a, _ in r["nexthops"])

    def nh_set(r):
        return sorted((str(a), str(i)) for a, i in r["nexthops"])

    def same(o, r) -> bool:
        # :875-879 — metric, tag, SR label and the next-hop set (inter-area / external rows carry `tag`, SR-enabled
        # instances `sr_label`; a row without the key has None, like the reference's Option)
        return (o["metric"] == r["metric"] and o.get("tag") == r.get("tag") and o.get("sr_label") == r.get("sr_label")
                and nh_set(o) == nh_set(r))
    old = {_net_key(r["prefix"]): r for r in old_rows}
    msgs: List[dict] = []
    for r in sorted(new_rows, key=lambda r: _net_key(r["prefix"])):
        o = old.pop(_net_key(r["prefix"]), None)
        if o is not None and same(o, r):
            continue
        if installed(r):
            # (a route whose next hops are partly interface-only keeps the addressed ones in the Address variant; the
            # reference's BTreeSet<Nexthop> would order Interface variants after them, never seen on an installed OSPF route)
            nhs = sorted(((ifindex[ifname], addr) for addr, ifname in r["nexthops"] if addr is not None),
                         key=lambda t: (t[0], int(ipaddress.ip_address(t[1]))))
            msg = {"op": "add", "prefix": r["prefix"], "metric": r["metric"], "nexthops": [list(t) for t in nhs]}
            if r.get("tag") is not None:
                msg["tag"] = r["tag"]
            msgs.append(msg)
    for k in sorted(old):
        if installed(old[k]):
            msgs.append({"op": "del", "prefix": old[k]["prefix"]})
    return msgs
