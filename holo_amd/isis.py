"""holo_amd.isis — host-side mirror of holo-isis' SPF path on top of the HIP engine.

Same names, argument meaning and failure behaviour as the reference functions it stands in for:

  compute_spt()      holo-isis/src/spf.rs:527-709      one SPT per (level, root, local, mt_id, mode)
  compute_spts()     holo-isis/src/flooding/manet.rs:47-69   the batched-roots caller: ONE engine run
  compute_routes()   holo-isis/src/spf.rs:840-949      RIB from an SPT
  compute_spf()      holo-isis/src/spf.rs:719-836      per-level orchestration + L1/L2 merge
                                                       (holo-isis/src/route.rs:185-249)

What runs where: the LSDB walk (`vertex_edges`, spf.rs:1013-1128) is done ONCE per LSDB generation
into the CSR of include/holo_spf_hip.h (LevelGraph); the SPT loop itself — distances, hops, ECMP
first-hop sets for every root — runs on the GPU through the C ABI; everything that needs
Interface / Adjacency objects (resolve_nexthop, spf.rs:956-1010) and the prefix attachment stays
here and consumes dist / hops / first_hop_mask, exactly as INTEGRATION.md describes for the Rust
side.  There is no CPU SPT loop in this module: without an engine it cannot compute anything.
"""
from __future__ import annotations

import ipaddress
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import engine as E

MAX_PATH_METRIC_STANDARD = 1023          # holo-isis/src/spf.rs:45
MAX_PATH_METRIC_WIDE = 0xFE000000        # holo-isis/src/spf.rs:47
MAX_LINK_METRIC_WIDE = 0x00FFFFFF        # holo-isis/src/spf.rs:49
MT_STANDARD, MT_IPV6_UNICAST = 0, 2
NLPID_IPV4, NLPID_IPV6 = 0xCC, 0x8E

VF_NETWORK, VF_NO_TRANSIT, VF_NO_EXPAND = 1, 2, 4

LanId = Tuple[bytes, int]                # (system id, pseudonode number)
VertexId = Tuple[bool, bytes, int]       # (non_pseudonode, system id, pseudonode): derive(Ord) order


def lan_id_from_str(s: str) -> LanId:
    a, b, c, pn = s.split(".")
    return bytes.fromhex(a + b + c), int(pn, 16)


def system_id_from_str(s: str) -> bytes:
    return bytes.fromhex(s.replace(".", ""))


def vertex_id(lan_id: LanId) -> VertexId:
    """holo-isis/src/spf.rs:96-100, 301-317."""
    return (lan_id[1] == 0, lan_id[0], lan_id[1])


# ---- LSDB model (the fields of Lsp / LspTlvs the path reads) ---------------------------------

@dataclass
class Lsp:
    system_id: bytes
    pseudonode: int
    fragment: int
    seqno: int = 1
    rem_lifetime: int = 1200
    overload: bool = False
    att: bool = False
    protocols_supported: Optional[List[int]] = None
    mt_flags: Dict[int, Tuple[bool, bool]] = field(default_factory=dict)   # mt -> (overload, att)
    is_reach: List[Tuple[LanId, int]] = field(default_factory=list)        # TLV 2
    ext_is_reach: List[Tuple[LanId, int]] = field(default_factory=list)    # TLV 22
    mt_is_reach: List[Tuple[int, LanId, int]] = field(default_factory=list)  # TLV 222
    ipv4_internal: List[Tuple[str, int]] = field(default_factory=list)     # TLV 128
    ipv4_external: List[Tuple[str, int]] = field(default_factory=list)     # TLV 130
    ext_ipv4: List[Tuple[str, int, bool]] = field(default_factory=list)    # TLV 135 (+X flag)
    ipv6: List[Tuple[str, int, bool]] = field(default_factory=list)        # TLV 236
    mt_ipv6: List[Tuple[int, str, int, bool]] = field(default_factory=list)  # TLV 237
    # Segment routing (Router Capability TLV 242 and the Prefix-SID sub-TLVs of algorithm SPF, kept beside the entries:
    # prefix_sids[kind][i] for entry i of ext_ipv4 / ipv6 / mt_ipv6): only compute_routes' SR step reads them
    sr_cap: Optional[dict] = None                      # {"flags": ["I", "V"], "srgb": [[first label, range], ...]}
    sr_algos: List[int] = field(default_factory=list)  # 0 = SPF
    prefix_sids: Dict[str, Dict[int, dict]] = field(default_factory=dict)

    def prefix_sid(self, kind: str, i: int) -> Optional[dict]:
        return self.prefix_sids.get(kind, {}).get(i)

    @property
    def lan_id(self) -> LanId:
        return (self.system_id, self.pseudonode)

    def live(self) -> bool:                # spf.rs:1024-1025
        return self.seqno != 0 and self.rem_lifetime != 0

    def overload_bit(self, mt_id: int) -> bool:   # holo-isis/src/packet/pdu.rs:1463-1477
        return self.overload if mt_id == MT_STANDARD else self.mt_flags.get(mt_id, (False, False))[0]

    def att_bit(self, mt_id: int) -> bool:        # pdu.rs:1446-1460
        return self.att if mt_id == MT_STANDARD else self.mt_flags.get(mt_id, (False, False))[1]


@dataclass
class Adjacency:
    system_id: bytes
    level_usage: str                       # "level-1" | "level-2" | "level-all"
    state: str = "up"
    ipv4_addrs: List[str] = field(default_factory=list)
    ipv6_addrs: List[str] = field(default_factory=list)
    topologies: List[int] = field(default_factory=lambda: [0])
    area_addrs: List[str] = field(default_factory=list)
    snpa: object = None                    # anything hashable & unique per adjacency

    def intersects(self, level: int) -> bool:
        return self.level_usage in ("level-all", f"level-{level}")


@dataclass
class Interface:
    name: str
    interface_type: str = "broadcast"      # "broadcast" | "point-to-point"
    metric: Dict[int, int] = field(default_factory=lambda: {1: 10, 2: 10})
    adjacencies: List[Adjacency] = field(default_factory=list)


@dataclass
class InstanceCfg:
    system_id: bytes
    level_type: str = "level-all"
    metric_type: Dict[int, str] = field(default_factory=lambda: {1: "wide", 2: "wide"})
    ipv4_enabled: bool = True
    ipv6_enabled: bool = True
    mt_ipv6_unicast: bool = False
    max_paths: int = 16
    att_ignore: bool = False
    area_addrs: List[str] = field(default_factory=list)
    sr_enabled: bool = False

    def is_af_enabled(self, af: str) -> bool:
        return self.ipv4_enabled if af == "ipv4" else self.ipv6_enabled

    def is_topology_enabled(self, mt_id: int) -> bool:
        return True if mt_id == MT_STANDARD else self.mt_ipv6_unicast

    def levels(self) -> List[int]:
        return {"level-1": [1], "level-2": [2], "level-all": [1, 2]}[self.level_type]


class Lsdb:
    """LSPs of one level ordered by LSP id (holo-isis/src/collections.rs:657-706)."""

    def __init__(self, lsps: Iterable[Lsp] = ()):
        self._by_id: Dict[Tuple[bytes, int, int], Lsp] = {}
        for l in lsps:
            self._by_id[(l.system_id, l.pseudonode, l.fragment)] = l
        self.generation = 0

    def insert(self, lsp: Lsp):
        self._by_id[(lsp.system_id, lsp.pseudonode, lsp.fragment)] = lsp
        self.generation += 1

    def iter(self):
        for k in sorted(self._by_id):
            yield self._by_id[k]

    def iter_for_lan_id(self, lan_id: LanId):
        return [l for k, l in sorted(self._by_id.items()) if k[0] == lan_id[0] and k[1] == lan_id[1]]

    def iter_for_system_id(self, system_id: bytes):
        return [l for k, l in sorted(self._by_id.items()) if k[0] == system_id]

    def zeroth_lsp(self, lan_id: LanId) -> Optional[Lsp]:      # spf.rs:1299-1309
        l = self._by_id.get((lan_id[0], lan_id[1], 0))
        return l if (l is not None and l.live()) else None


@dataclass
class Instance:
    """The slice of InstanceUpView the path reads."""
    config: InstanceCfg
    interfaces: List[Interface]
    lsdb: Dict[int, Lsdb]

    def interfaces_by_name(self) -> List[Interface]:           # collections.rs:258-265
        return sorted(self.interfaces, key=lambda i: i.name)

    def is_l2_attached_to_backbone(self, mt_id: int) -> bool:  # holo-isis/src/instance.rs:577-591
        mine = set(self.config.area_addrs)
        return any(mt_id in a.topologies and a.state == "up" and a.intersects(2)
                   and mine.isdisjoint(a.area_addrs)
                   for i in self.interfaces_by_name() for a in i.adjacencies)

    @classmethod
    def from_vector(cls, vec: dict) -> "Instance":
        """Build from a tests/golden/isis/*.json vector (tools/make_golden.py)."""
        c = vec["config"]
        cfg = InstanceCfg(system_id=system_id_from_str(c["system_id"]), level_type=c["level_type"],
                          metric_type={1: c["metric_type"]["1"], 2: c["metric_type"]["2"]},
                          ipv4_enabled=c["afs"].get("ipv4", True), ipv6_enabled=c["afs"].get("ipv6", True),
                          mt_ipv6_unicast=c["mt_ipv6_unicast"], max_paths=c["max_paths"],
                          att_ignore=c["att_ignore"], area_addrs=list(c["area_addrs"]),
                          sr_enabled=bool(c.get("sr_enabled", False)))
        ifaces = []
        for i in vec["interfaces"]:
            adjs = [Adjacency(system_id_from_str(a["system_id"]), a["usage"], a["state"], list(a["ipv4"]),
                              list(a["ipv6"]), list(a["topologies"]), list(a["area_addrs"]),
                              snpa=(i["name"], a["system_id"], a["usage"]))
                    for a in i["adjacencies"]]
            ifaces.append(Interface(i["name"], i["type"], {1: i["metric"]["1"], 2: i["metric"]["2"]}, adjs))
        lsdb = {}
        for lv, lsps in vec["lsdb"].items():
            out = []
            for l in lsps:
                lan, frag = l["id"].rsplit("-", 1)
                sysid, pn = lan_id_from_str(lan)
                out.append(Lsp(
                    sysid, pn, int(frag, 16), seqno=l.get("seqno", 1), rem_lifetime=l.get("lifetime", 1200),
                    overload="ol" in l["flags"], att="att" in l["flags"], protocols_supported=l["protocols"],
                    mt_flags={m["id"]: ("ol" in m["flags"], "att" in m["flags"]) for m in l["mt"]},
                    is_reach=[(lan_id_from_str(n), m) for n, m in l["is_reach"]],
                    ext_is_reach=[(lan_id_from_str(n), m) for n, m in l["ext_is_reach"]],
                    mt_is_reach=[(t, lan_id_from_str(n), m) for t, n, m in l["mt_is_reach"]],
                    ipv4_internal=[tuple(x) for x in l["ipv4_int"]], ipv4_external=[tuple(x) for x in l["ipv4_ext"]],
                    ext_ipv4=[tuple(x) for x in l["ext_ipv4"]], ipv6=[tuple(x) for x in l["ipv6"]],
                    mt_ipv6=[tuple(x) for x in l["mt_ipv6"]],
                    sr_cap=l.get("sr_cap"), sr_algos=list(l.get("sr_algos", [])),
                    prefix_sids={k: {int(i): sid for i, sid in d.items()} for k, d in l.get("prefix_sids", {}).items()}))
            lsdb[int(lv)] = Lsdb(out)
        return cls(cfg, ifaces, lsdb)


# ---- LSDB -> CSR ------------------------------------------------------------------------------

def vertex_edges(lsp: Lsp, mt_id: Optional[int], hopcount: bool, metric_type: str):
    """Edges one live fragment contributes, in the reference's order (spf.rs:1026-1127)."""
    std_on = metric_type in ("standard", "both")
    wide_on = metric_type in ("wide", "both")

    def cost(nbr: LanId, metric: int) -> int:                  # spf.rs:1131-1146
        return metric if not hopcount else (0 if nbr[1] != 0 else 1)

    if (mt_id is None or mt_id == MT_STANDARD) and std_on:
        for nbr, m in lsp.is_reach:
            yield nbr, cost(nbr, m)
    if ((mt_id is None or mt_id == MT_STANDARD) or lsp.pseudonode != 0) and wide_on:
        for nbr, m in lsp.ext_is_reach:
            if m < MAX_LINK_METRIC_WIDE:
                yield nbr, cost(nbr, m)
    if mt_id is not None and mt_id != MT_STANDARD:
        for t, nbr, m in lsp.mt_is_reach:
            if t == mt_id and m < MAX_LINK_METRIC_WIDE:
                yield nbr, cost(nbr, m)
    if mt_id is None:
        for _t, nbr, m in lsp.mt_is_reach:
            if m < MAX_LINK_METRIC_WIDE:
                yield nbr, cost(nbr, m)


class LevelGraph:
    """CSR form (include/holo_spf_hip.h) of one level's LSDB for one (mt_id, metric mode).

    Vertex index = rank in VertexId order over the LAN ids that own at least one live LSP
    fragment; links to LAN ids without any LSP are not listed (they can never pass the two-way
    check).  `edge_cost[k]` keeps the link cost the reference hands to resolve_nexthop."""

    def __init__(self, instance: Instance, level: int, mt_id: Optional[int], hopcount: bool = False):
        cfg = instance.config
        lsdb = instance.lsdb.get(level) or Lsdb()
        self.level, self.mt_id, self.hopcount = level, mt_id, hopcount
        self.metric_type = cfg.metric_type[level]
        self.cfg_key = self._cfg_key(cfg, level)
        frags = self._live_fragments(lsdb)
        self.vids: List[VertexId] = sorted(vertex_id(k) for k in frags)
        self.index: Dict[VertexId, int] = {v: i for i, v in enumerate(self.vids)}
        n = len(self.vids)
        row_ptr = np.zeros(n + 1, np.uint32)
        col, met = [], []
        vflags = np.zeros(n, np.uint8)
        for i, vid in enumerate(self.vids):
            c, m, f = self._row((vid[1], vid[2]), frags, lsdb, cfg)
            col += c
            met += m
            vflags[i] = f
            row_ptr[i + 1] = len(col)
        self.row_ptr = row_ptr
        self.col = np.asarray(col, np.uint32)
        self.metric = np.asarray(met, np.uint32)
        self.vflags = vflags
        self.max_path_metric = (MAX_PATH_METRIC_STANDARD if self.metric_type == "standard"
                                else MAX_PATH_METRIC_WIDE)                 # spf.rs:637-641
        self.run_flags = E.RUN_IGNORE_OVERLOAD if mt_id is None else 0    # spf.rs:566-574
        self._dev = None

    @staticmethod
    def _cfg_key(cfg: InstanceCfg, level: int):
        """Everything outside the LSDB that a row depends on (vertex_edges' TLV choice, the gates)."""
        return (cfg.metric_type[level], cfg.is_af_enabled("ipv4"), cfg.is_af_enabled("ipv6"))

    @staticmethod
    def _live_fragments(lsdb: Lsdb) -> Dict[LanId, List[Lsp]]:
        frags: Dict[LanId, List[Lsp]] = {}
        for l in lsdb.iter():
            if l.live():
                frags.setdefault(l.lan_id, []).append(l)
        return frags

    def _row(self, lan: LanId, frags, lsdb: Lsdb, cfg: InstanceCfg):
        """Links (in fragment, then TLV order: spf.rs:1013-1128) and gate flags (spf.rs:557-604) of one vertex."""
        col, met = [], []
        for lsp in frags[lan]:
            for nbr, c in vertex_edges(lsp, self.mt_id, self.hopcount, self.metric_type):
                j = self.index.get(vertex_id(nbr))
                if j is not None:
                    col.append(j)
                    met.append(c)
        is_pn = lan[1] != 0
        f = VF_NETWORK if is_pn else 0
        z = lsdb.zeroth_lsp(lan)
        if z is None:
            return col, met, f | VF_NO_EXPAND                           # spf.rs:557-561
        if not is_pn and self.mt_id is not None and z.overload_bit(self.mt_id):
            f |= VF_NO_TRANSIT                                          # spf.rs:568-574
        if self.mt_id is not None and self.mt_id == MT_STANDARD and not is_pn:   # spf.rs:582-604
            ps = z.protocols_supported
            if ps is None or (cfg.is_af_enabled("ipv4") and NLPID_IPV4 not in ps) \
                    or (cfg.is_af_enabled("ipv6") and NLPID_IPV6 not in ps):
                f |= VF_NO_EXPAND
        return col, met, f

    def refresh(self, instance: Instance, changed: Iterable[LanId]) -> bool:
        """Incremental re-derivation after the LSPs of `changed` LAN ids were re-originated, purged or aged out
        (the reference's `trigger_lsps`, holo-isis/src/spf.rs:144,735): only their rows are rebuilt and — when the
        graph is on the device — replaced there by hspf_graph_patch.  Returns False, leaving everything untouched,
        when the change cannot be expressed as row replacements (a vertex appeared or vanished, or the
        configuration the rows depend on changed): the caller builds a new LevelGraph."""
        cfg = instance.config
        lsdb = instance.lsdb.get(self.level) or Lsdb()
        if self._cfg_key(cfg, self.level) != self.cfg_key:
            return False
        frags = self._live_fragments(lsdb)
        if len(frags) != len(self.vids) or any(vertex_id(k) not in self.index for k in frags):
            return False
        vs = sorted({self.index[vertex_id(lan)] for lan in changed if vertex_id(lan) in self.index})
        if not vs:
            return True
        rows, flags = [], []
        for i in vs:
            vid = self.vids[i]
            c, m, f = self._row((vid[1], vid[2]), frags, lsdb, cfg)
            rows.append((np.asarray(c, np.uint32), np.asarray(m, np.uint32)))
            flags.append(f)
        if self._dev is not None:
            self._dev[1].patch(vs, rows, flags)
        self.row_ptr, self.col, self.metric, self.vflags = E.splice_rows(
            self.row_ptr, self.col, self.metric, self.vflags, vs, [r[0] for r in rows], [r[1] for r in rows],
            np.asarray(flags, np.uint8))
        return True

    @property
    def n(self) -> int:
        return len(self.vids)

    def device(self, engine):
        if self._dev is None or self._dev[0] is not engine:
            g = engine.upload(self.row_ptr, self.col, self.metric, self.vflags, self.max_path_metric)
            self._dev = (engine, g)
        return self._dev[1]

    def links_back(self, t: int, v: int) -> bool:
        return bool((self.col[self.row_ptr[t]:self.row_ptr[t + 1]] == v).any())


# ---- SPT ------------------------------------------------------------------------------------------

@dataclass
class VertexNexthop:                       # holo-isis/src/spf.rs:107-114
    system_id: bytes
    iface_name: Optional[str] = None
    ipv4: Optional[str] = None
    ipv6: Optional[str] = None


@dataclass
class Vertex:                              # holo-isis/src/spf.rs:78-88
    id: VertexId
    distance: int
    hops: int
    nexthops: List[VertexNexthop] = field(default_factory=list)


class Spt:
    """holo-isis/src/spf.rs:67-73, 224-297."""

    def __init__(self):
        self.vertices: Dict[VertexId, Vertex] = {}
        self._pop_order: List[VertexId] = []
        self._parents: Optional[Dict[VertexId, List[VertexId]]] = None
        self._parent_src = None            # (graph, dist row, hops row, in-SPT row, rank_key) for the lazy rebuild

    def contains(self, vid: VertexId) -> bool:
        return vid in self.vertices

    def get(self, vid: VertexId) -> Optional[Vertex]:
        return self.vertices.get(vid)

    def iter(self):
        for k in sorted(self.vertices):
            yield self.vertices[k]

    def first_hops(self):
        return [self.vertices[v] for v in self._pop_order if v[0] and self.vertices[v].hops == 1]

    def second_hops(self):
        return [self.vertices[v] for v in self._pop_order if v[0] and self.vertices[v].hops == 2]

    def parents(self, vid: VertexId) -> List[VertexId]:
        """`Vertex.parents` (spf.rs:85, 677): every relaxation that reached the vertex at its final
        distance, in the order they happened (parents in pop order, their links in LSP order,
        parallel links repeated).  Rebuilt on first use from the tight links of the graph."""
        if self._parents is None:
            self._parents = {}
            if self._parent_src is not None:
                g, dist, hops, in_spt, rank_key = self._parent_src
                ignore_ovl = g.mt_id is None
                members = sorted(np.nonzero(in_spt)[0].tolist(), key=rank_key)
                for u in members:
                    f = g.vflags[u]
                    if f & VF_NO_EXPAND:
                        continue
                    if hops[u] != 0 and not (f & VF_NETWORK) and not ignore_ovl and (f & VF_NO_TRANSIT):
                        continue
                    for k in range(int(g.row_ptr[u]), int(g.row_ptr[u + 1])):
                        t = int(g.col[k])
                        if not in_spt[t] or rank_key(t) <= rank_key(u) or not g.links_back(t, u):
                            continue
                        if min(int(dist[u]) + int(g.metric[k]), 0xFFFFFFFF) == int(dist[t]):
                            self._parents.setdefault(g.vids[t], []).append(g.vids[u])
        return self._parents.get(vid, [])

    def is_on_path(self, ancestor: bytes, descendant: bytes) -> bool:
        """spf.rs:261-286: does `ancestor` appear on ANY shortest path from `descendant` to the root."""
        a, d = vertex_id((ancestor, 0)), vertex_id((descendant, 0))
        if a not in self.vertices or d not in self.vertices:
            return False
        stack, seen = [d], set()
        while stack:
            cur = stack.pop()
            if cur == a:
                return True
            if cur in seen:
                continue
            seen.add(cur)
            stack.extend(self.parents(cur))
        return False


def resolve_nexthop(nexthop: VertexNexthop, level: int, mt_id: int, parent_is_pseudonode: bool,
                    target_system_id: bytes, link_cost: int, used_adjs: set, ifaces: Sequence[Interface]):
    """holo-isis/src/spf.rs:956-1010 (ifaces already in name order)."""
    want = "broadcast" if parent_is_pseudonode else "point-to-point"
    for iface in ifaces:
        if iface.interface_type != want:
            continue
        adj = None
        if want == "broadcast":
            adj = next((a for a in iface.adjacencies
                        if a.level_usage == f"level-{level}" and a.system_id == target_system_id), None)
            if adj is not None and (mt_id not in adj.topologies or adj.state != "up"):
                adj = None
        else:
            if iface.metric[level] != link_cost:
                continue
            a = iface.adjacencies[0] if iface.adjacencies else None
            if (a is not None and mt_id in a.topologies and a.intersects(level)
                    and a.system_id == target_system_id and a.state == "up"):
                adj = a
        if adj is None or adj.snpa in used_adjs:
            continue
        used_adjs.add(adj.snpa)
        nexthop.iface_name = iface.name
        nexthop.ipv4 = adj.ipv4_addrs[0] if adj.ipv4_addrs else None
        nexthop.ipv6 = adj.ipv6_addrs[0] if adj.ipv6_addrs else None
        return


def _slot_nexthops(g: LevelGraph, G, root: int, dist: np.ndarray, hops: np.ndarray, in_spt: np.ndarray,
                   rank_key, local: bool, level: int, instance: Instance) -> Dict[int, VertexNexthop]:
    """Replays the relaxations made from hops == 0 vertices, in the reference's order, to give every
    first-hop slot its VertexNexthop (spf.rs:680-701).  resolve_nexthop is order dependent through
    `used_adjs`, and the reference calls it for EVERY relaxation that is not `Ordering::Greater`
    at that moment — also for candidates a later, shorter path replaces — so the replay evaluates
    the candidate-list state at each of those moments from the final distances."""
    hv, hb, _total = G.slot_table(root)
    parents = [(int(p), int(b)) for p, b in zip(hv, hb) if in_spt[p] and hops[p] == 0]
    parents.sort(key=lambda pb: rank_key(pb[0]))
    ifaces = instance.interfaces_by_name()
    used_adjs: set = set()
    out: Dict[int, VertexNexthop] = {}
    max_path = g.max_path_metric
    ignore_ovl = g.mt_id is None

    def expandable(u: int) -> bool:
        f = g.vflags[u]
        if f & VF_NO_EXPAND:
            return False
        if hops[u] != 0 and not (f & VF_NETWORK) and not ignore_ovl and (f & VF_NO_TRANSIT):
            return False
        return True

    def cand_before(t: int, p: int, upto_k: int) -> int:
        """Distance of t on the candidate list just before link `upto_k` of p is processed."""
        best = None
        pk = rank_key(p)
        for u in set(int(x) for x in g.col[g.row_ptr[t]:g.row_ptr[t + 1]]):   # two-way => u lists t
            if not in_spt[u] or not expandable(u):
                continue
            uk = rank_key(u)
            if uk > pk:
                continue
            for k in range(int(g.row_ptr[u]), int(g.row_ptr[u + 1])):
                if int(g.col[k]) != t:
                    continue
                if u == p and k >= upto_k:
                    break
                d = min(int(dist[u]) + int(g.metric[k]), 0xFFFFFFFF)
                if d <= max_path and (best is None or d < best):
                    best = d
        return best

    for p, base in parents:
        if not expandable(p):
            continue
        for k in range(int(g.row_ptr[p]), int(g.row_ptr[p + 1])):
            t = int(g.col[k])
            if not g.links_back(t, p):
                continue
            if in_spt[t] and rank_key(t) < rank_key(p):          # already on the SPT
                continue
            d = min(int(dist[p]) + int(g.metric[k]), 0xFFFFFFFF)
            if d > max_path:
                continue
            cur = cand_before(t, p, k)
            if cur is not None and d > cur:                       # Ordering::Greater
                continue
            if g.vflags[t] & VF_NETWORK:                          # pseudonode target: no next hop
                continue
            nh = VertexNexthop(system_id=g.vids[t][1])
            if local and g.mt_id is not None:
                resolve_nexthop(nh, level, g.mt_id, bool(g.vflags[p] & VF_NETWORK), g.vids[t][1],
                                int(g.metric[k]), used_adjs, ifaces)
            out[base + (k - int(g.row_ptr[p]))] = nh
    return out


def compute_spts(level: int, root_system_ids: Sequence[bytes], local: bool, mt_id: Optional[int],
                 hopcount: bool, instance: Instance, engine, graph: Optional[LevelGraph] = None) -> List[Spt]:
    """All SPTs of one level/topology for a list of roots with ONE engine run — the shape of
    flooding::manet::init_cache (holo-isis/src/flooding/manet.rs:47-69)."""
    g = graph or LevelGraph(instance, level, mt_id, hopcount)
    spts: List[Optional[Spt]] = [None] * len(root_system_ids)
    roots, where = [], []
    for i, sid in enumerate(root_system_ids):
        rv = vertex_id((sid, 0))
        r = g.index.get(rv)
        if r is None:
            # Root without any LSP: it is inserted into the SPT and not expanded (spf.rs:552-561).
            s = Spt()
            s.vertices[rv] = Vertex(rv, 0, 0)
            s._pop_order = [rv]
            spts[i] = s
        else:
            roots.append(r)
            where.append(i)
    if not roots:
        return spts  # type: ignore[return-value]
    G = g.device(engine)
    res = engine.run(G, np.asarray(roots, np.uint32), g.run_flags)
    W = res.first_hop_mask.shape[2]
    # Roots the engine had to run through its sequential kernel (zero-cost plateaus: the pop order
    # is not the static (distance, id) order) are re-run once more for their exact pop ranks.
    exact_j = [j for j in range(len(roots)) if (res.flags[j] & E.RF_EXACT).any()]
    pop_rank = {}
    if exact_j:
        rr = engine.run(G, np.asarray([roots[j] for j in exact_j], np.uint32), g.run_flags | E.RUN_POP_RANK)
        for q, j in enumerate(exact_j):
            pop_rank[j] = rr.pop_rank[q]
    for j, (r, i) in enumerate(zip(roots, where)):
        dist, hops = res.dist[j], res.hops[j]
        in_spt = (res.flags[j] & E.RF_IN_SPT) != 0
        if j in pop_rank:
            pr = pop_rank[j]
            rank_key = lambda v, pr=pr: (int(pr[v]), 0, 0, 0)           # noqa: E731
        elif g.hopcount:
            # Hop-count graphs (spf.rs:1138-1145): links into pseudonodes cost 0, so a pseudonode is
            # put on the candidate list by the lowest-numbered router of its own distance that lists
            # it and, sorting before every router, is popped right after that router.
            def rank_key(v, dist=dist, in_spt=in_spt, g=g, cache={}):   # noqa: B006
                k = cache.get(v)
                if k is None:
                    k = (int(dist[v]), v, 0, 0)
                    if g.vflags[v] & VF_NETWORK:
                        acts = [int(u) for u in g.col[g.row_ptr[v]:g.row_ptr[v + 1]]
                                if in_spt[u] and dist[u] == dist[v] and not (g.vflags[u] & VF_NO_EXPAND)
                                and g.links_back(int(u), v)]
                        if acts:
                            k = (int(dist[v]), min(acts), 1, v)
                    cache[v] = k
                return k
        else:
            rank_key = lambda v, dist=dist: (int(dist[v]), v, 0, 0)     # noqa: E731  static order
        slot_nh = _slot_nexthops(g, G, r, dist, hops, in_spt, rank_key, local, level, instance)
        s = Spt()
        members = np.nonzero(in_spt)[0]
        for v in members.tolist():
            vx = Vertex(g.vids[v], int(dist[v]), int(hops[v]))
            for w in range(W):
                m = int(res.first_hop_mask[j, v, w])
                while m:
                    b = (m & -m).bit_length() - 1
                    m &= m - 1
                    nh = slot_nh.get(w * 64 + b)
                    if nh is not None:
                        vx.nexthops.append(nh)
            s.vertices[vx.id] = vx
        s._pop_order = [g.vids[v] for v in sorted(members.tolist(), key=rank_key)]
        s._parent_src = (g, dist, hops, in_spt, rank_key)
        spts[i] = s
    return spts  # type: ignore[return-value]


def compute_spt(level: int, root_system_id: bytes, local: bool, mt_id: Optional[int], hopcount: bool,
                instance: Instance, engine, graph: Optional[LevelGraph] = None) -> Spt:
    """holo-isis/src/spf.rs:527-709."""
    return compute_spts(level, [root_system_id], local, mt_id, hopcount, instance, engine, graph)[0]


@dataclass
class NeighborCache:                       # holo-isis/src/flooding/manet.rs:30-35
    spt_hopcount: Spt
    remote_nbr_list: Dict[bytes, str]


def manet_init_cache(level: int, instance: Instance, engine, flooding_algo_of=None) -> Dict[bytes, NeighborCache]:
    """flooding::manet::init_cache (holo-isis/src/flooding/manet.rs:39-97): one hop-count SPT per Up
    adjacency — here ONE batched engine run instead of a sequential loop — and, per neighbour, its
    remote neighbour list (the first hops of that SPT with the flooding algorithm each advertises;
    `flooding_algo_of(system_id)` supplies the sub-TLV value, default zero-pruner as in :84)."""
    nbrs: List[bytes] = []
    for iface in instance.interfaces_by_name():
        for adj in iface.adjacencies:
            if adj.state == "up" and adj.system_id not in nbrs:
                nbrs.append(adj.system_id)
    spts = compute_spts(level, nbrs, False, None, True, instance, engine)
    out = {}
    for sid, spt in zip(nbrs, spts):
        rnl = {v.id[1]: (flooding_algo_of(v.id[1]) if flooding_algo_of else "zero-pruner") for v in spt.first_hops()}
        out[sid] = NeighborCache(spt, dict(sorted(rnl.items())))
    return out


def flood_reduction_hash(lsp_id: Tuple[bytes, int, int]) -> int:
    """flood_reduction_hash (holo-isis/src/flooding/manet.rs:190-194): Fletcher-16 of the 8-byte LSP id with
    the fragment number shifted right by 3 (`fletcher` crate, calc_fletcher16: two running sums mod 255 over
    the bytes, result = sum2 << 8 | sum1).  Pinned by the reference's unit test vectors
    (manet.rs:205-232, from draft-ietf-lsr-distoptflood-12 section 1.2.3) in tests/test_host_manet.py."""
    system_id, pseudonode, fragment = lsp_id
    s1 = s2 = 0
    for b in bytes(system_id) + bytes([pseudonode & 0xFF, (fragment & 0xFF) >> 3]):
        s1 = (s1 + b) % 255
        s2 = (s2 + s1) % 255
    return (s2 << 8) | s1


def reflood_list(cache: Dict[bytes, NeighborCache], local_system_id: bytes, tn: bytes,
                 lsp_id: Tuple[bytes, int, int]) -> List[bytes]:
    """flooding::manet::reflood_list (holo-isis/src/flooding/manet.rs:99-173): the neighbours of transmitting
    neighbour `tn` that this router has to re-flood LSP `lsp_id` to.  Every query is answered from the
    hop-count SPT of the batched run in manet_init_cache: second hops in pop order, is_on_path over the tight
    parent links.  Returns the BTreeSet order (ascending system id)."""
    c = cache.get(tn)
    if c is None or not c.remote_nbr_list:
        return []
    spt = c.spt_hopcount
    originator = lsp_id[0]
    thl = sorted({v.id[1] for v in spt.second_hops()                        # :120-132
                  if v.id[1] != originator and not spt.is_on_path(v.id[1], originator)})
    rnl = list(c.remote_nbr_list.items())                                    # BTreeMap order
    rnum = len(rnl)
    n0 = flood_reduction_hash(lsp_id) % rnum                                 # :135-139
    out: List[bytes] = []
    for k in range(rnum):                                                    # circular from index N, :143-170
        if not thl:
            break
        sid, algo = rnl[(n0 + k) % rnum]
        if sid == local_system_id:
            out = [t for t in thl if spt.is_on_path(sid, t)]
            break
        if algo != "modified-manet":
            continue
        thl = [t for t in thl if not spt.is_on_path(sid, t)]
    return sorted(out)


# ---- flooding::manet with the ancestor queries answered from device-built bit sets ------------------

@dataclass
class DeviceNeighborCache:
    """What reflood_list needs of one neighbour's hop-count SPT, taken from hspf_ancestors_device instead of a walk
    over parent links: the remote-neighbour list with each member's bit in the level-1 sets, the second hops with (a)
    their own level-2 bit and (b) the set of first hops above them, and the level-2 set of every vertex (row of this
    root) for "which second hops lie on a shortest path to the LSP originator"."""
    remote_nbr_list: Dict[bytes, str]
    rnl_bit: Dict[bytes, int]                 # first-hop router -> bit index in the level-1 sets
    second: List[Tuple[bytes, int, int]]      # second-hop routers: (system id, level-2 bit, level-1 ancestor set as int)
    anc2_of: Dict[bytes, int]                 # router system id -> its level-2 ancestor set as int (0 when not in the SPT)
    fallback: Optional[NeighborCache] = None  # root handled by the sequential kernel: host walk


def _bits(row) -> int:
    x = 0
    for w, word in enumerate(row.tolist()):
        x |= int(word) << (64 * w)
    return x


def manet_init_cache_device(level: int, instance: Instance, engine, flooding_algo_of=None,
                            device="cuda:0") -> Dict[bytes, DeviceNeighborCache]:
    """flooding::manet::init_cache (holo-isis/src/flooding/manet.rs:39-97) with everything reflood_list asks of the
    SPTs computed on the device: ONE batched hop-count run (results stay in HBM), then hspf_ancestors_device for levels
    1 and 2.  What comes back to the host is the remote-neighbour list, the second hops and the bit sets."""
    import torch
    nbrs: List[bytes] = []
    for iface in instance.interfaces_by_name():
        for adj in iface.adjacencies:
            if adj.state == "up" and adj.system_id not in nbrs:
                nbrs.append(adj.system_id)
    g = LevelGraph(instance, level, None, True)
    out: Dict[bytes, DeviceNeighborCache] = {}
    have = [(sid, g.index.get(vertex_id((sid, 0)))) for sid in nbrs]
    for sid, r in have:
        if r is None:                        # root without any LSP: alone in its SPT (spf.rs:552-561), empty RNL
            out[sid] = DeviceNeighborCache({}, {}, [], {})
    run = [(sid, r) for sid, r in have if r is not None]
    if not run:
        return out
    roots = np.asarray([r for _, r in run], np.uint32)
    R, n = len(roots), g.n
    G = g.device(engine)
    dev = torch.device(device)
    dist = torch.empty((R, n), dtype=torch.int32, device=dev)
    hops = torch.empty((R, n), dtype=torch.int16, device=dev)
    flags = torch.empty((R, n), dtype=torch.int16, device=dev)
    engine.run_device(G, roots, g.run_flags, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr())
    levels = {}
    for L in (1, 2):
        W = 1
        while True:
            rank = torch.empty((R, n), dtype=torch.int32, device=dev)
            cnt = torch.empty((R,), dtype=torch.int32, device=dev)
            anc = torch.empty((R, n, W), dtype=torch.int64, device=dev)
            rc = engine.ancestors_device(G, roots, g.run_flags, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(),
                                         flags_ptr=flags.data_ptr(), level=L, n_words=W, level_rank_ptr=rank.data_ptr(),
                                         level_count_ptr=cnt.data_ptr(), anc_ptr=anc.data_ptr())
            c = cnt.cpu().numpy().view(np.uint32)
            if rc == 0:
                break
            W = int(max((int(x) + 63) // 64 for x in c if x != 0xFFFFFFFF))
        levels[L] = (rank.cpu().numpy().view(np.uint32), c, anc.cpu().numpy().view(np.uint64))
    routers = [(v, g.vids[v][1]) for v in range(n) if g.vids[v][0]]          # (vertex, system id) of router vertices
    host_cache = None
    for j, (sid, r) in enumerate(run):
        if levels[1][1][j] == 0xFFFFFFFF:                                     # dynamic pop order: the host walk
            if host_cache is None:
                host_cache = manet_init_cache(level, instance, engine, flooding_algo_of)
            hc = host_cache[sid]
            out[sid] = DeviceNeighborCache(hc.remote_nbr_list, {}, [], {}, hc)
            continue
        rank1, rank2 = levels[1][0][j], levels[2][0][j]
        anc1, anc2 = levels[1][2][j], levels[2][2][j]
        rnl_bit = {s2: int(rank1[v]) for v, s2 in routers if rank1[v] != 0xFFFFFFFF}
        rnl = {s2: (flooding_algo_of(s2) if flooding_algo_of else "zero-pruner") for s2 in rnl_bit}
        second = [(s2, int(rank2[v]), _bits(anc1[v])) for v, s2 in routers if rank2[v] != 0xFFFFFFFF]
        anc2_of = {s2: _bits(anc2[v]) for v, s2 in routers}
        out[sid] = DeviceNeighborCache(dict(sorted(rnl.items())), rnl_bit, second, anc2_of)
    return out


def reflood_list_device(cache: Dict[bytes, DeviceNeighborCache], local_system_id: bytes, tn: bytes,
                        lsp_id: Tuple[bytes, int, int]) -> List[bytes]:
    """flooding::manet::reflood_list (holo-isis/src/flooding/manet.rs:99-173) on the device-built sets: every
    Spt::is_on_path of the reference is one bit test here."""
    c = cache.get(tn)
    if c is None or not c.remote_nbr_list:
        return []
    if c.fallback is not None:
        return reflood_list({tn: c.fallback}, local_system_id, tn, lsp_id)
    originator = lsp_id[0]
    on_path_to_orig = c.anc2_of.get(originator, 0)                           # level-2 routers above (or equal to) the originator
    thl = sorted((s2, a1) for s2, bit2, a1 in c.second                       # :120-132
                 if s2 != originator and not (on_path_to_orig >> bit2) & 1)
    rnl = list(c.remote_nbr_list.items())
    rnum = len(rnl)
    n0 = flood_reduction_hash(lsp_id) % rnum
    out: List[bytes] = []
    for k in range(rnum):
        if not thl:
            break
        sid, algo = rnl[(n0 + k) % rnum]
        b = c.rnl_bit[sid]
        if sid == local_system_id:
            out = [t for t, a1 in thl if (a1 >> b) & 1]
            break
        if algo != "modified-manet":
            continue
        thl = [(t, a1) for t, a1 in thl if not (a1 >> b) & 1]
    return sorted(out)


def should_flood(iface: Interface, reflood: Sequence[bytes]) -> bool:      # manet.rs:176-186
    rs = set(reflood)
    return any(a.state == "up" and a.system_id in rs for a in iface.adjacencies)


# ---- routes ------------------------------------------------------------------------------------------

def _addr_key(a: str):
    ip = ipaddress.ip_address(a)
    return (ip.version, int(ip))


def _net_key(p: str):
    n = ipaddress.ip_network(p, strict=False)
    return (n.version, int(n.network_address), n.prefixlen)


@dataclass
class Route:                               # holo-isis/src/route.rs:27-37
    prefix: str
    metric: int
    level: int
    external: bool
    connected: bool
    nexthops: Dict[tuple, Tuple[str, str, bytes, Optional[int]]]   # addr key -> (addr, iface, system id, SR output label)
    prefix_sid: Optional[dict] = None       # the Prefix-SID of the network the route was CREATED from (route.rs:78-103)
    sr_label: Optional[int] = None          # SR input label


def vertex_networks_sr(instance: Instance, level: int, mt_id: int, lan_id: LanId, att_bit: bool,
                       l2_attached: bool, ipv4_enabled: bool, ipv6_enabled: bool):
    """holo-isis/src/spf.rs:1149-1296: (prefix, metric, external, Prefix-SID of algorithm SPF or None)."""
    cfg = instance.config
    metric_type = cfg.metric_type[level]
    std_on = metric_type in ("standard", "both")
    wide_on = metric_type in ("wide", "both")
    for lsp in instance.lsdb[level].iter_for_lan_id(lan_id):
        if not lsp.live():
            continue
        if att_bit and level == 1 and (cfg.level_type == "level-1" or not l2_attached):
            if ipv4_enabled:
                yield "0.0.0.0/0", 0, False, None
            if ipv6_enabled:
                yield "::/0", 0, False, None
        if mt_id == MT_STANDARD and ipv4_enabled:
            if std_on:
                for p, m in lsp.ipv4_internal:
                    yield p, m, False, None
                for p, m in lsp.ipv4_external:
                    yield p, m, True, None
            if wide_on:
                for i, (p, m, x) in enumerate(lsp.ext_ipv4):
                    if m <= MAX_PATH_METRIC_WIDE:
                        yield p, m, x, lsp.prefix_sid("ext_ipv4", i)
        if ipv6_enabled:
            it = ([(p, m, x, lsp.prefix_sid("mt_ipv6", i)) for i, (t, p, m, x) in enumerate(lsp.mt_ipv6) if t == MT_IPV6_UNICAST]
                  if mt_id == MT_IPV6_UNICAST else
                  [(p, m, x, lsp.prefix_sid("ipv6", i)) for i, (p, m, x) in enumerate(lsp.ipv6)])
            for p, m, x, sid in it:
                if m <= MAX_PATH_METRIC_WIDE:
                    yield p, m, x, sid


def vertex_networks(instance: Instance, level: int, mt_id: int, lan_id: LanId, att_bit: bool,
                    l2_attached: bool, ipv4_enabled: bool, ipv6_enabled: bool):
    """(prefix, metric, external) of holo-isis/src/spf.rs:1149-1296 (what route derivation on the device needs)."""
    for p, m, x, _sid in vertex_networks_sr(instance, level, mt_id, lan_id, att_bit, l2_attached, ipv4_enabled, ipv6_enabled):
        yield p, m, x


def _build_nexthops(vertex: Vertex, prefix: str):          # route.rs:118-142
    v6 = ":" in prefix
    out = {}
    for nh in vertex.nexthops:
        addr = nh.ipv6 if v6 else nh.ipv4
        if addr is not None:
            out[_addr_key(addr)] = (addr, nh.iface_name, nh.system_id, None)
    return out


# ---- SR Prefix-SID bookkeeping of compute_routes (holo-isis/src/spf.rs:931-946, sr.rs:34-94, 165-300) ---------------------
# The SPT feeds it two bits per route update: local = (vertex.hops == 0), last_hop = (vertex.hops == 1).

LABEL_IMPLICIT_NULL, LABEL_EXPLICIT_NULL_V4, LABEL_EXPLICIT_NULL_V6 = 3, 0, 2


def _sr_cap(lsdb: "Lsdb", system_id: bytes) -> Optional[dict]:
    for lsp in lsdb.iter_for_system_id(system_id):
        if lsp.live() and lsp.sr_cap is not None:
            return lsp.sr_cap
    return None


def sr_index_to_label(index: int, srgbs) -> Optional[int]:             # sr.rs:270-300
    for first, rng in srgbs:
        if index >= rng:
            index -= rng
            continue
        return first + index
    return None


def prefix_sid_update(instance: Instance, level: int, adv_rtr: LanId, prefix: str, route: Route, local: bool,
                      last_hop: bool):
    """sr.rs:34-94: SR input label of the route, output label of every next hop; a label that cannot be resolved stays."""
    sid = route.prefix_sid
    lsdb = instance.lsdb[level]
    if sid is None or not any(lsp.live() and 0 in lsp.sr_algos for lsp in lsdb.iter_for_lan_id(adv_rtr)):
        return
    flags = sid["flags"]
    v6 = ":" in prefix
    # input label (sr.rs:165-205)
    if local and ("P" not in flags or "E" in flags):
        route.sr_label = None
    elif "index" in sid:
        cap = _sr_cap(lsdb, instance.config.system_id)
        label = sr_index_to_label(sid["index"], cap["srgb"]) if cap is not None else None
        if label is not None:
            route.sr_label = label
    else:
        route.sr_label = sid["label"]
    # output labels (sr.rs:208-267)
    for k, (addr, iface, system_id, old) in list(route.nexthops.items()):
        if last_hop and "P" not in flags:
            label = LABEL_IMPLICIT_NULL
        else:
            cap = _sr_cap(lsdb, system_id)
            if cap is None or ("V" if v6 else "I") not in cap["flags"]:
                continue
            if last_hop and "E" in flags:
                label = LABEL_EXPLICIT_NULL_V6 if v6 else LABEL_EXPLICIT_NULL_V4
            elif "index" in sid:
                label = sr_index_to_label(sid["index"], cap["srgb"])
                if label is None:
                    continue
            else:
                label = sid["label"] if last_hop else LABEL_IMPLICIT_NULL
        route.nexthops[k] = (addr, iface, system_id, label)


def compute_routes(level: int, mt_id: int, instance: Instance, spt: Spt, rib: Dict[tuple, Route]):
    """holo-isis/src/spf.rs:840-949."""
    cfg = instance.config
    l2_attached = instance.is_l2_attached_to_backbone(mt_id)
    ipv4_enabled = cfg.is_af_enabled("ipv4") and mt_id == MT_STANDARD
    ipv6_enabled = cfg.is_af_enabled("ipv6") and (
        (not cfg.is_topology_enabled(MT_IPV6_UNICAST)) if mt_id == MT_STANDARD else True)
    lsdb = instance.lsdb.get(level) or Lsdb()
    for vertex in spt.iter():
        lan = (vertex.id[1], vertex.id[2])
        z = lsdb.zeroth_lsp(lan)
        if z is None:
            continue
        att = (not cfg.att_ignore) and z.att_bit(mt_id) and not z.overload_bit(mt_id)
        for prefix, metric, external, sid in vertex_networks_sr(instance, level, mt_id, lan, att, l2_attached,
                                                                ipv4_enabled, ipv6_enabled):
            key = _net_key(prefix)
            route_metric = vertex.distance + metric
            cur = rib.get(key)
            if cur is None or route_metric < cur.metric:
                cur = rib[key] = Route(prefix, route_metric, level, external, vertex.hops == 0,
                                       _build_nexthops(vertex, prefix), prefix_sid=sid)
            elif route_metric == cur.metric:
                cur.nexthops.update(_build_nexthops(vertex, prefix))
            else:
                continue
            if len(cur.nexthops) > cfg.max_paths:
                cur.nexthops = {k: cur.nexthops[k] for k in sorted(cur.nexthops)[:cfg.max_paths]}
            if cfg.sr_enabled and cur.prefix_sid is not None:                       # spf.rs:931-946
                prefix_sid_update(instance, level, lan, prefix, cur, vertex.hops == 0, vertex.hops == 1)


class GraphCache:
    """Device graphs kept across SPF runs, one per (level, topology, metric mode), brought up to date from the
    changed LSPs instead of being re-derived from the whole LSDB."""

    def __init__(self):
        self.graphs: Dict[tuple, LevelGraph] = {}
        self.rebuilt = 0
        self.patched = 0

    def get(self, instance: Instance, level: int, mt_id: Optional[int], hopcount: bool = False,
            trigger_lsps: Optional[Iterable[LanId]] = None) -> LevelGraph:
        """`trigger_lsps` = LAN ids whose LSPs changed since the previous call for this instance (None: unknown,
        rebuild)."""
        key = (level, mt_id, hopcount)
        g = self.graphs.get(key)
        if g is not None and trigger_lsps is not None and g.refresh(instance, trigger_lsps):
            self.patched += 1
            return g
        if g is not None and g._dev is not None:
            g._dev[1].free()
        g = self.graphs[key] = LevelGraph(instance, level, mt_id, hopcount)
        self.rebuilt += 1
        return g


def changed_lan_ids(old: Lsdb, new: Lsdb) -> List[LanId]:
    """LAN ids with an LSP fragment that differs between two LSDB snapshots (what the reference accumulates in
    `trigger_lsps` as LSPs are installed, holo-isis/src/lsdb.rs)."""
    a = {(l.system_id, l.pseudonode, l.fragment): l for l in old.iter()}
    b = {(l.system_id, l.pseudonode, l.fragment): l for l in new.iter()}
    return sorted({(k[0], k[1]) for k in set(a) | set(b) if a.get(k) != b.get(k)})


def compute_spf(instance: Instance, engine, cache: Optional[GraphCache] = None,
                trigger_lsps: Optional[Dict[int, Iterable[LanId]]] = None) -> List[dict]:
    """Full SPF of every configured level and topology (holo-isis/src/spf.rs:719-836) followed by
    the L1/L2 merge of holo-isis/src/route.rs:185-249; returns the rows of the YANG `local-rib`.
    With a GraphCache the level graphs persist on the device between calls and `trigger_lsps[level]` (changed LAN
    ids) turns the LSDB -> CSR step into row patches."""
    cfg = instance.config
    per_level: Dict[int, Dict[tuple, Route]] = {}
    for level in cfg.levels():
        if level not in instance.lsdb:
            instance.lsdb[level] = Lsdb()
        rib: Dict[tuple, Route] = {}
        for mt_id in (MT_STANDARD, MT_IPV6_UNICAST):
            if cfg.is_topology_enabled(mt_id):
                graph = None
                if cache is not None:
                    graph = cache.get(instance, level, mt_id, False,
                                      None if trigger_lsps is None else trigger_lsps.get(level, ()))
                spt = compute_spt(level, cfg.system_id, True, mt_id, False, instance, engine, graph)
                compute_routes(level, mt_id, instance, spt, rib)
        per_level[level] = rib
    merged: Dict[tuple, Route] = {}
    for level in (2, 1):
        merged.update(per_level.get(level, {}))
    rows = []
    for key in sorted(merged):
        r = merged[key]
        row = {"prefix": r.prefix, "metric": r.metric, "level": r.level,
               "nexthops": [[r.nexthops[k][0], r.nexthops[k][1]] for k in sorted(r.nexthops)]}
        if cfg.sr_enabled:                                        # the SR columns only where SR is on
            row["sr_label"] = r.sr_label
            row["nexthop_labels"] = [r.nexthops[k][3] for k in sorted(r.nexthops)]
        rows.append(row)
    return rows


# ---- the wire step after the path: update_global_rib, holo-isis/src/route.rs:254-312 ------------------

def _nh_key(row: dict):
    """Next hops as the reference compares them (whole Nexthop structs, holo-isis/src/route.rs:270-272): address,
    interface and SR output label (rows of an SR-enabled instance carry `nexthop_labels`)."""
    labels = row.get("nexthop_labels") or [None] * len(row["nexthops"])
    return sorted((tuple(nh), -1 if lb is None else lb) for nh, lb in zip(row["nexthops"], labels))


def update_global_rib(new_rows: List[dict], old_rows: List[dict], ifindex: Dict[str, int],
                      unchanged: Optional[Iterable[str]] = None) -> List[dict]:
    """The RouteIpAdd / RouteIpDel messages a new local RIB puts on the ibus, in emission order (rows as compute_spf
    returns them).  `unchanged`: prefixes a device-side diff (hspf_routes_diff_device through holo_amd.routes) has already
    found identical — they are skipped without comparing next hops on the host; everything else is compared here.
    Same rules as the reference: unchanged routes are not re-sent (:268-277), CONNECTED / next-hop-less routes are not
    installed (:283-287), old routes that were installed and are gone are withdrawn (:303-310)."""
    import ipaddress
    skip = set(unchanged or ())
    old = {_net_key(r["prefix"]): r for r in old_rows}
    msgs: List[dict] = []
    for r in sorted(new_rows, key=lambda r: _net_key(r["prefix"])):
        o = old.pop(_net_key(r["prefix"]), None)
        # (a device-side diff sees metric and next-hop masks, not SR labels: with labels in play the host compares)
        labelled = any(lb is not None for row in (r, o or {}) for lb in (row.get("nexthop_labels") or ()))
        if o is not None and ((r["prefix"] in skip and not labelled) or (o["metric"] == r["metric"] and _nh_key(o) == _nh_key(r))):
            continue
        if r["nexthops"]:
            nhs = sorted(((ifindex[ifname], addr) for addr, ifname in r["nexthops"]),
                         key=lambda t: (t[0], ipaddress.ip_address(t[1]).version, int(ipaddress.ip_address(t[1]))))
            msgs.append({"op": "add", "prefix": r["prefix"], "metric": r["metric"], "nexthops": [list(t) for t in nhs]})
    for k in sorted(old):
        if old[k]["nexthops"]:
            msgs.append({"op": "del", "prefix": old[k]["prefix"]})
    return msgs
