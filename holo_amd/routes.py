"""holo_amd.routes — route derivation with the prefix attachment on the GPU.

`compute_routes` of holo-isis (holo-isis/src/spf.rs:840-949) walks the SPT in VertexId order and,
per advertised prefix, keeps the smallest `distance + metric`, merging next hops on ties.  That is a
segmented min-reduce with a mask OR: here the prefix table is built once per LSDB generation
(PrefixTable, root independent), and `hspf_routes_device` evaluates it for every root of a batched run
straight from the device-resident dist / flags / first-hop-mask tables — the tables never travel to
the host.  What needs addresses (Nexthop objects, max-paths truncation by address) stays on the host.

torch is used for device buffers only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import engine as E
from . import isis as I


@dataclass
class PrefixTable:
    prefixes: List[str]                # sorted by (family, address, length) = BTreeMap<IpNetwork, _> order
    pfx_ptr: np.ndarray                # u32 [P+1]
    pfx_vertex: np.ndarray             # u32 [n_entries]  ascending vertex index inside a prefix
    pfx_metric: np.ndarray             # u32 [n_entries]
    external: np.ndarray               # bool [n_entries]

    @classmethod
    def build(cls, instance: I.Instance, level: int, mt_id: int, g: I.LevelGraph) -> "PrefixTable":
        """Every (vertex, prefix, metric) the reference's vertex_networks() would yield
        (holo-isis/src/spf.rs:1149-1296), vertices in VertexId order, LSP order inside a vertex."""
        cfg = instance.config
        l2_attached = instance.is_l2_attached_to_backbone(mt_id)
        v4 = cfg.is_af_enabled("ipv4") and mt_id == I.MT_STANDARD
        v6 = cfg.is_af_enabled("ipv6") and ((not cfg.is_topology_enabled(I.MT_IPV6_UNICAST))
                                             if mt_id == I.MT_STANDARD else True)
        lsdb = instance.lsdb.get(level) or I.Lsdb()
        rows = []
        for v, vid in enumerate(g.vids):
            lan = (vid[1], vid[2])
            z = lsdb.zeroth_lsp(lan)
            if z is None:
                continue                                                     # spf.rs:866-869
            att = (not cfg.att_ignore) and z.att_bit(mt_id) and not z.overload_bit(mt_id)
            for prefix, metric, external in I.vertex_networks(instance, level, mt_id, lan, att, l2_attached, v4, v6):
                rows.append((I._net_key(prefix), prefix, v, metric, external))
        keys = sorted({r[0]: r[1] for r in rows}.items())
        pid = {k: i for i, (k, _) in enumerate(keys)}
        order = sorted(range(len(rows)), key=lambda i: (pid[rows[i][0]], rows[i][2], i))   # stable inside a vertex
        ptr = np.zeros(len(keys) + 1, np.uint32)
        for r in rows:
            ptr[pid[r[0]] + 1] += 1
        ptr = np.cumsum(ptr, dtype=np.uint64).astype(np.uint32)
        return cls([p for _, p in keys], ptr,
                   np.asarray([rows[i][2] for i in order], np.uint32),
                   np.asarray([rows[i][3] for i in order], np.uint32),
                   np.asarray([rows[i][4] for i in order], bool))


@dataclass
class DeviceRoutes:
    """Per (root, prefix) results of hspf_routes_device, copied to the host."""
    best_metric: np.ndarray            # [R, P] u32
    best_entry: np.ndarray             # [R, P] u32
    nexthop_mask: np.ndarray           # [R, P, W] u64


def spt_and_routes_device(engine, g: I.LevelGraph, roots: Sequence[int], table: PrefixTable, device="cuda:0"):
    """One batched engine run + one route-derivation launch, everything device resident; returns
    (SpfResult-like host tables, DeviceRoutes).  The SPT tables are copied back here only because the
    callers below also want them for next-hop resolution / checking."""
    import torch
    roots = np.ascontiguousarray(roots, np.uint32)
    R, n = len(roots), g.n
    G = g.device(engine)
    W = G.mask_words(roots)
    dev = torch.device(device)
    dist = torch.empty((R, n), dtype=torch.int32, device=dev)
    hops = torch.empty((R, n), dtype=torch.int16, device=dev)
    flags = torch.empty((R, n), dtype=torch.int16, device=dev)
    mask = torch.empty((R, n, W), dtype=torch.int64, device=dev)
    stats = engine.run_device(G, roots, g.run_flags, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(),
                              flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
    P = len(table.prefixes)
    bm = torch.empty((R, P), dtype=torch.int32, device=dev)
    be = torch.empty((R, P), dtype=torch.int32, device=dev)
    nm = torch.empty((R, P, W), dtype=torch.int64, device=dev)
    engine.routes_device(n, R, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), table.pfx_ptr, table.pfx_vertex,
                         table.pfx_metric, best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(),
                         nexthop_mask_ptr=nm.data_ptr())
    torch.cuda.synchronize(dev)
    res = E.SpfResult(dist.cpu().numpy().view(np.uint32), hops.cpu().numpy().view(np.uint16),
                      flags.cpu().numpy().view(np.uint16), mask.cpu().numpy().view(np.uint64), None, stats)
    routes = DeviceRoutes(bm.cpu().numpy().view(np.uint32), be.cpu().numpy().view(np.uint32),
                          nm.cpu().numpy().view(np.uint64))
    return res, routes


def compute_spf_device_routes(instance: I.Instance, engine, device="cuda:0") -> List[dict]:
    """compute_spf (holo-isis/src/spf.rs:719-836) with BOTH the SPT and the prefix attachment on the
    GPU; rows of the YANG `local-rib` like holo_amd.isis.compute_spf."""
    cfg = instance.config
    per_level: Dict[int, Dict[tuple, dict]] = {}
    for level in cfg.levels():
        if level not in instance.lsdb:
            instance.lsdb[level] = I.Lsdb()
        rib: Dict[tuple, dict] = {}
        for mt_id in (I.MT_STANDARD, I.MT_IPV6_UNICAST):
            if not cfg.is_topology_enabled(mt_id):
                continue
            g = I.LevelGraph(instance, level, mt_id, False)
            root = g.index.get(I.vertex_id((cfg.system_id, 0)))
            if root is None:
                continue                       # root without LSP: SPT = {root}, zeroth LSP missing -> no routes
            table = PrefixTable.build(instance, level, mt_id, g)
            if not table.prefixes:
                continue
            res, routes = spt_and_routes_device(engine, g, [root], table, device)
            dist, hops = res.dist[0], res.hops[0]
            in_spt = (res.flags[0] & E.RF_IN_SPT) != 0
            G = g.device(engine)
            if (res.flags[0] & E.RF_EXACT).any():
                rr = engine.run(G, np.asarray([root], np.uint32), g.run_flags | E.RUN_POP_RANK)
                pr = rr.pop_rank[0]
                rank_key = lambda v: (int(pr[v]), 0)                          # noqa: E731
            else:
                rank_key = lambda v: (int(dist[v]), v)                        # noqa: E731
            slot_nh = I._slot_nexthops(g, G, root, dist, hops, in_spt, rank_key, True, level, instance)
            for p, prefix in enumerate(table.prefixes):
                metric = int(routes.best_metric[0, p])
                if metric == 0xFFFFFFFF:
                    continue
                v6 = ":" in prefix
                nhs = {}
                for w in range(routes.nexthop_mask.shape[2]):
                    m = int(routes.nexthop_mask[0, p, w])
                    while m:
                        b = (m & -m).bit_length() - 1
                        m &= m - 1
                        nh = slot_nh.get(w * 64 + b)
                        if nh is None:
                            continue
                        addr = nh.ipv6 if v6 else nh.ipv4
                        if addr is not None:
                            nhs[I._addr_key(addr)] = (addr, nh.iface_name)
                key = I._net_key(prefix)
                cur = rib.get(key)
                # two topologies never share a prefix family, but keep compute_routes' compare anyway
                if cur is None or metric < cur["metric"]:
                    rib[key] = {"prefix": prefix, "metric": metric, "level": level, "nexthops": nhs}
                elif metric == cur["metric"]:
                    cur["nexthops"].update(nhs)
                r = rib[key]
                if len(r["nexthops"]) > cfg.max_paths:
                    r["nexthops"] = {k: r["nexthops"][k] for k in sorted(r["nexthops"])[:cfg.max_paths]}
        per_level[level] = rib
    merged: Dict[tuple, dict] = {}
    for level in (2, 1):
        merged.update(per_level.get(level, {}))
    return [{"prefix": merged[k]["prefix"], "metric": merged[k]["metric"], "level": merged[k]["level"],
             "nexthops": [[merged[k]["nexthops"][a][0], merged[k]["nexthops"][a][1]] for a in sorted(merged[k]["nexthops"])]}
            for k in sorted(merged)]
