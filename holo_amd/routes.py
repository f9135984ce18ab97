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
from typing import Dict, List, Sequence

import numpy as np

from . import engine as E
from . import isis as I
from . import ospf as O


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


# ---- the wire step from device tables: diff on the device, ONE packed copy, messages on the host -----

def expand_route_records(records: np.ndarray, prefixes: Sequence[str], slot_nh: Dict[int, object], old_rows: Dict[tuple, dict],
                         ifindex: Dict[str, int], max_paths: int) -> List[dict]:
    """The record stream of hspf_routes_pack (one root) -> the RouteIpAdd / RouteIpDel sequence of update_global_rib
    (holo-isis/src/route.rs:254-312) in the reference's order: installs in prefix order, then the withdrawals of the
    prefixes that have no route any more.  A record is a CANDIDATE: the device compared metric and first-hop slot
    masks, which is finer than the reference's comparison of next-hop sets (two slots may resolve to one adjacency; the
    host truncates to max-paths) — each candidate is confirmed against the old row here, on the few records only."""
    import ipaddress
    adds, dels = [], []
    W = (records.shape[1] - 6) // 2 if len(records) else 0
    for rec in records:
        p, action, metric, entry = int(rec[1]), int(rec[2]), int(rec[3]), int(rec[4])
        prefix = prefixes[p]
        if action == E.DIFF_WITHDRAW:
            if entry == 0xFFFFFFFF:
                dels.append({"op": "del", "prefix": prefix})
            continue
        if action != E.DIFF_INSTALL:
            continue
        v6 = ":" in prefix
        nhs = {}
        for w in range(W):
            m = int(rec[6 + 2 * w]) | (int(rec[7 + 2 * w]) << 32)
            while m:
                b = (m & -m).bit_length() - 1
                m &= m - 1
                nh = slot_nh.get(w * 64 + b)
                if nh is None:
                    continue
                addr = nh.ipv6 if v6 else nh.ipv4
                if addr is not None:
                    nhs[I._addr_key(addr)] = (addr, nh.iface_name)
        keep = [nhs[k] for k in sorted(nhs)[:max_paths]]
        o = old_rows.get(I._net_key(prefix))
        if o is not None and o["metric"] == metric and sorted(map(tuple, o["nexthops"])) == sorted(keep):
            continue                                               # the reference's "unchanged" (:268-277)
        if keep:
            wire = sorted(((ifindex[ifn], a) for a, ifn in keep),
                          key=lambda t: (t[0], ipaddress.ip_address(t[1]).version, int(ipaddress.ip_address(t[1]))))
            adds.append({"op": "add", "prefix": prefix, "metric": metric, "nexthops": [list(t) for t in wire]})
    return adds + dels


def update_global_rib_device(instance: I.Instance, engine, rib_before: List[dict], ifindex: Dict[str, int], device="cuda:0"):
    """compute_spf + update_global_rib with the SPT, the prefix attachment, the comparison with the previous RIB and the
    compaction of what changed ALL on the device; one record stream comes back (hspf_routes_pack).  For instances with
    one (level, topology) table — the shape of every IS-IS step fixture; the L1/L2 merge of a two-level instance is host
    logic (holo_amd.isis.compute_spf).  Returns (messages, number of records copied, number of prefixes compared)."""
    import torch
    cfg = instance.config
    # The device comparison sees metric + first-hop slots.  SR labels (route.sr_label, per-next-hop labels) are compared by
    # the host rule only (holo_amd.isis.update_global_rib, holo-isis/src/route.rs:254-312: a label-only change re-sends the
    # route): an SR-enabled instance, or an old RIB that carries labels, takes the host path — never a silent skip.
    if getattr(cfg, "sr_enabled", False) or any(r.get("sr_label") is not None or r.get("nexthop_labels") for r in rib_before):
        new_rows = I.compute_spf(instance, engine)
        return I.update_global_rib(new_rows, rib_before, ifindex), 0, 0
    tabs = [(lv, mt) for lv in cfg.levels() for mt in (I.MT_STANDARD, I.MT_IPV6_UNICAST) if cfg.is_topology_enabled(mt)]
    if len(tabs) != 1:
        raise ValueError("update_global_rib_device: one (level, topology) table only")
    level, mt_id = tabs[0]
    if level not in instance.lsdb:
        instance.lsdb[level] = I.Lsdb()
    old_rows = {I._net_key(r["prefix"]): r for r in rib_before}
    g = I.LevelGraph(instance, level, mt_id, False)
    root = g.index.get(I.vertex_id((cfg.system_id, 0)))
    table = PrefixTable.build(instance, level, mt_id, g) if root is not None else None
    if root is None or not table.prefixes:
        # no SPT / nothing advertised: the new RIB is empty and every installed route goes (host, nothing to compare)
        return I.update_global_rib([], rib_before, ifindex), 0, 0
    # ONE prefix list for both sides: the table's prefixes plus those only the old RIB knows (no entries: no new route)
    keys = {I._net_key(p): p for p in table.prefixes}
    for k, r in old_rows.items():
        keys.setdefault(k, r["prefix"])
    order = sorted(keys)
    prefixes = [keys[k] for k in order]
    P = len(prefixes)
    cnt = np.zeros(P + 1, np.uint32)
    where = {k: i for i, k in enumerate(order)}
    tpos = [where[I._net_key(p)] for p in table.prefixes]
    for j, i in enumerate(tpos):
        cnt[i + 1] = table.pfx_ptr[j + 1] - table.pfx_ptr[j]
    ptr = np.cumsum(cnt, dtype=np.uint64).astype(np.uint32)           # table.prefixes is sorted the same way: entries keep their order
    roots = np.asarray([root], np.uint32)
    n = g.n
    G = g.device(engine)
    W = G.mask_words(roots)
    dev = torch.device(device)
    dist = torch.empty((1, n), dtype=torch.int32, device=dev); hops = torch.empty((1, n), dtype=torch.int16, device=dev)
    flags = torch.empty((1, n), dtype=torch.int16, device=dev); mask = torch.empty((1, n, W), dtype=torch.int64, device=dev)
    engine.run_device(G, roots, g.run_flags, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                      mask_ptr=mask.data_ptr(), mask_words=W)
    bm = torch.empty((1, P), dtype=torch.int32, device=dev); be = torch.empty((1, P), dtype=torch.int32, device=dev)
    nm = torch.empty((1, P, W), dtype=torch.int64, device=dev)
    engine.routes_device(n, 1, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), ptr, table.pfx_vertex, table.pfx_metric,
                         best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr())
    # first-hop slots -> next hops (needs Interface / Adjacency objects: host, once per slot)
    hd, hh = dist.cpu().numpy().view(np.uint32)[0], hops.cpu().numpy().view(np.uint16)[0]
    hf = flags.cpu().numpy().view(np.uint16)[0]
    in_spt = (hf & E.RF_IN_SPT) != 0
    if (hf & E.RF_EXACT).any():
        pr = engine.run(G, roots, g.run_flags | E.RUN_POP_RANK).pop_rank[0]
        rank_key = lambda v: (int(pr[v]), 0)                          # noqa: E731
    else:
        rank_key = lambda v: (int(hd[v]), v)                          # noqa: E731
    slot_nh = I._slot_nexthops(g, G, root, hd, hh, in_spt, rank_key, True, level, instance)
    # the OLD RIB in the same index space: metric, and the slots whose next hop the old route used.  A next hop of the old
    # route that no slot resolves to any more (its adjacency is gone) cannot be expressed: the metric is poisoned so that
    # the pair compares unequal and the host decides.
    om = np.full((1, P), 0xFFFFFFFF, np.uint32); oe = np.full((1, P), 0xFFFFFFFF, np.uint32); on = np.zeros((1, P, W), np.uint64)
    for k, r in old_rows.items():
        i = where[k]
        v6 = ":" in r["prefix"]
        want = {tuple(nh) for nh in r["nexthops"]}
        seen = set()
        for slot, nh in slot_nh.items():
            addr = nh.ipv6 if v6 else nh.ipv4
            if addr is not None and (addr, nh.iface_name) in want:
                on[0, i, slot // 64] |= np.uint64(1) << np.uint64(slot % 64)
                seen.add((addr, nh.iface_name))
        # (more next hops than max-paths allows NOW: the new route will be truncated on the host: not comparable by mask)
        om[0, i] = r["metric"] if seen == want and len(want) <= cfg.max_paths else 0xFFFFFFFE
        oe[0, i] = 0
        if want and not on[0, i].any():
            on[0, i, 0] = np.uint64(1)            # "it was installed with next hops" (the poisoned metric keeps the pair unequal)
    t_om = torch.from_numpy(om.view(np.int32)).to(dev); t_oe = torch.from_numpy(oe.view(np.int32)).to(dev)
    t_on = torch.from_numpy(on.view(np.int64)).to(dev)
    act = torch.empty((1, P), dtype=torch.uint8, device=dev)
    chg = torch.empty((P,), dtype=torch.int32, device=dev); cptr = torch.empty((2,), dtype=torch.int32, device=dev)
    new = (bm.data_ptr(), be.data_ptr(), nm.data_ptr())
    engine.routes_diff_device(1, P, W, (t_om.data_ptr(), t_oe.data_ptr(), t_on.data_ptr()), new,
                              action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
    rec = engine.routes_pack(1, P, W, new, action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
    return expand_route_records(rec, prefixes, slot_nh, old_rows, ifindex, cfg.max_paths), len(rec), P


# ---- OSPFv2: update_rib_intra_area with the prefix attachment on the GPU ---------------------------------------------

@dataclass
class OspfPrefixTables:
    """CSR-by-prefix tables of one area (root independent): `net` = the prefixes of Network-LSA vertices (metric 0),
    `stub` = the stub links of Router-LSA vertices, in the order intra_area_networks() yields them
    (holo-ospf/src/ospfv2/spf.rs:462-538).  Two tables because the two vertex kinds follow different tie rules
    (include/holo_spf_hip.h, HSPF_PFX_LAST_MIN) and all networks precede all routers in VertexId order."""
    prefixes: List[str]
    net: tuple                        # (pfx_ptr, pfx_vertex, pfx_metric)
    stub: tuple

    @classmethod
    def build(cls, g: "O.AreaGraph") -> "OspfPrefixTables":
        import ipaddress
        rows = {"net": [], "stub": []}
        for v, vid in enumerate(g.vids):
            lsa = g.lsa_of(v)
            if vid[0] == O.NET:
                try:
                    rows["net"].append((str(ipaddress.ip_network((lsa.lsa_id, lsa.mask), strict=False)), v, 0))
                except ValueError:
                    continue
            else:
                for link in lsa.links:
                    if link.link_type != "stub-network-link":
                        continue
                    try:
                        rows["stub"].append((str(ipaddress.ip_network((link.link_id, link.link_data), strict=False)), v, link.metric))
                    except ValueError:
                        continue
        keys = sorted({O._net_key(p): p for kind in rows.values() for p, _, _ in kind}.items())
        pid = {k: i for i, (k, _) in enumerate(keys)}

        def table(rs):
            order = sorted(range(len(rs)), key=lambda i: (pid[O._net_key(rs[i][0])], rs[i][1], i))
            ptr = np.zeros(len(keys) + 1, np.uint64)
            for p, _, _ in rs:
                ptr[pid[O._net_key(p)] + 1] += 1
            return (np.cumsum(ptr).astype(np.uint32), np.asarray([rs[i][1] for i in order], np.uint32),
                    np.asarray([rs[i][2] for i in order], np.uint32))
        return cls([p for _, p in keys], table(rows["net"]), table(rows["stub"]))


def ospf_area_device_routes(engine, g: "O.AreaGraph", root: int, tables: OspfPrefixTables, rib: dict, max_paths: int,
                            device="cuda:0") -> None:
    """run_area + update_rib_intra_area of one area with the SPT and both prefix reductions on the device; folds the
    per-prefix results into `rib` (shared by the areas, holo-ospf/src/route.rs:146-160) with the reference's compare
    rules.  Only the tables needed to turn first-hop slots into next hops come back to the host."""
    import torch
    G = g.device(engine)
    roots = np.asarray([root], np.uint32)
    n, W = g.n if hasattr(g, "n") else len(g.vids), G.mask_words(roots)
    dev = torch.device(device)
    dist = torch.empty((1, n), dtype=torch.int32, device=dev)
    hops = torch.empty((1, n), dtype=torch.int16, device=dev)
    flags = torch.empty((1, n), dtype=torch.int16, device=dev)
    mask = torch.empty((1, n, W), dtype=torch.int64, device=dev)
    stats = engine.run_device(G, roots, E.RUN_NET_NEXTHOPS, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(),
                              flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
    P = len(tables.prefixes)
    out = {}
    for kind, tab, fl in (("net", tables.net, E.PFX_SATURATING | E.PFX_LAST_MIN), ("stub", tables.stub, E.PFX_SATURATING)):
        bm = torch.empty((1, P), dtype=torch.int32, device=dev)
        be = torch.empty((1, P), dtype=torch.int32, device=dev)
        nm = torch.empty((1, P, W), dtype=torch.int64, device=dev)
        if P:
            engine.routes_device(n, 1, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), tab[0], tab[1], tab[2],
                                 best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(),
                                 nexthop_mask_ptr=nm.data_ptr(), flags=fl)
        torch.cuda.synchronize(dev)
        out[kind] = (bm.cpu().numpy().view(np.uint32)[0], be.cpu().numpy().view(np.uint32)[0],
                     nm.cpu().numpy().view(np.uint64)[0])
    res = E.SpfResult(dist.cpu().numpy().view(np.uint32), hops.cpu().numpy().view(np.uint16),
                      flags.cpu().numpy().view(np.uint16), mask.cpu().numpy().view(np.uint64), None, stats)
    slot_nh: dict = {}
    O.spt_from_engine(g, root, engine, O.calc_nexthops, res=res, slots_out=slot_nh)

    def expand(mrow) -> dict:
        nhs = {}
        for w in range(len(mrow)):
            m = int(mrow[w])
            while m:
                b = (m & -m).bit_length() - 1
                m &= m - 1
                nh = slot_nh.get(w * 64 + b)
                if nh:
                    nhs.update(nh)
        return nhs

    for p, prefix in enumerate(tables.prefixes):
        key = O._net_key(prefix)
        for kind, tab in (("net", tables.net), ("stub", tables.stub)):      # networks before routers: VertexId order
            bm, be, nm = out[kind]
            if be[p] == 0xFFFFFFFF:
                continue
            v = int(tab[1][be[p]])
            lsa = g.lsa_of(v)
            metric = int(bm[p])
            origin = O.ip(lsa.lsa_id) if kind == "net" else O.ip(lsa.adv_rtr)
            cur = rib.get(key)
            if cur is not None and metric > cur["metric"]:
                continue
            if kind == "net" and cur is not None:                            # route.rs:388-400
                if metric < cur["metric"] or (metric == cur["metric"] and origin > cur["origin"]):
                    del rib[key]
                else:
                    continue
            new = {"prefix": prefix, "metric": metric, "origin": origin, "connected": int(res.hops[0][v]) == 0,
                   "nexthops": expand(nm[p])}
            cur = rib.get(key)                                               # route_update, route.rs:918-965
            if cur is None or new["metric"] < cur["metric"]:
                cur = rib[key] = new
            elif new["metric"] == cur["metric"]:
                cur["nexthops"].update(new["nexthops"])
            if len(cur["nexthops"]) > max_paths:
                cur["nexthops"] = {k: cur["nexthops"][k] for k in sorted(cur["nexthops"])[:max_paths]}


def ospf_intra_area_device_routes(router_id: str, areas: Sequence["O.Area"], max_paths: int, engine,
                                  device="cuda:0") -> List[dict]:
    """holo_amd.ospf.compute_spf_intra_area with SPT and prefix attachment on the GPU; same rows."""
    rib: dict = {}
    for area in sorted(areas, key=lambda a: O.ip(a.area_id)):
        g = O.AreaGraph(area)
        root = g.index.get((O.RTR, O.ip(router_id)))
        if root is None:
            continue
        ospf_area_device_routes(engine, g, root, OspfPrefixTables.build(g), rib, max_paths, device)
        if g._dev is not None:
            g._dev[1].free()
    return [{"prefix": rib[k]["prefix"], "metric": rib[k]["metric"], "type": "intra-area",
             "nexthops": [[rib[k]["nexthops"][a][1], rib[k]["nexthops"][a][0]] for a in sorted(rib[k]["nexthops"])]}
            for k in sorted(rib)]


# ---- OSPFv2: the wire step from device tables -----------------------------------------------------------

@dataclass
class Ospfv2OrderedTable:
    """ONE CSR-by-prefix table of an OSPFv2 area for HSPF_PFX_ORDERED: the stub networks exactly as
    `Ospfv2::intra_area_networks` yields them (holo-ospf/src/ospfv2/spf.rs:462-538) — the SPT in VertexId order, i.e. all
    Network-LSA vertices (their own prefix, metric 0) before all Router-LSA vertices (their stub links in LSA order) —,
    grouped by prefix with that order kept inside a prefix.  The ordered fold (k_routes_ordered) then IS
    update_rib_intra_area (route.rs:343-448): network-entry rule, saturating add, better replaces, equal merges — one
    result row per prefix, which is what the RIB comparison on the device needs (the two-table form above leaves the fold
    of its two results to the host)."""
    prefixes: List[str]
    pfx_ptr: np.ndarray
    pfx_vertex: np.ndarray            # vertex | PFX_ENTRY_NETWORK
    pfx_metric: np.ndarray
    pfx_origin: np.ndarray            # LS-ID of the vertex's LSA (route.rs:380-381)

    @classmethod
    def build(cls, g: "O.AreaGraph") -> "Ospfv2OrderedTable":
        import ipaddress
        rows = []
        for v, vid in enumerate(g.vids):                       # index order = VertexId order = spt.values()
            lsa = g.lsa_of(v)
            if vid[0] == O.NET:
                try:
                    p = str(ipaddress.ip_network((lsa.lsa_id, lsa.mask), strict=False))
                except ValueError:
                    continue
                rows.append((O._net_key(p), p, v | E.PFX_ENTRY_NETWORK, 0, O.ip(lsa.lsa_id)))
            else:
                for link in lsa.links:
                    if link.link_type != "stub-network-link":
                        continue
                    try:
                        p = str(ipaddress.ip_network((link.link_id, link.link_data), strict=False))
                    except ValueError:
                        continue
                    rows.append((O._net_key(p), p, v, link.metric, O.ip(lsa.adv_rtr)))
        keys = sorted({r[0]: r[1] for r in rows}.items())
        pid = {k: i for i, (k, _) in enumerate(keys)}
        order = sorted(range(len(rows)), key=lambda i: (pid[rows[i][0]], i))
        ptr = np.zeros(len(keys) + 1, np.uint64)
        for r in rows:
            ptr[pid[r[0]] + 1] += 1
        return cls([p for _, p in keys], np.cumsum(ptr).astype(np.uint32),
                   np.asarray([rows[i][2] for i in order], np.uint32), np.asarray([rows[i][3] for i in order], np.uint32),
                   np.asarray([rows[i][4] for i in order], np.uint32))


def ospf_update_global_rib_device(router_id: str, areas: Sequence["O.Area"], max_paths: int, engine, rib_before: List[dict],
                                  ifindex: Dict[str, int], other_rows: Sequence[dict] = (), device="cuda:0", version: int = 2,
                                  af: str = "ipv6"):
    """compute_spf's intra-area part + update_global_rib (holo-ospf/src/route.rs:856-916, ibus/tx.rs:32-77) with the SPTs, the
    ORDERED prefix fold of EVERY area into one RIB, the comparison with the RIB held before and the compaction of what
    changed all on the device: one record stream comes back (hspf_routes_pack) and is expanded into the RouteIpAdd /
    RouteIpDel sequence.  Several areas (round 5): the RIB is shared by the areas (route.rs:146-160, the per-area loop
    spf.rs:540-542) — each area's table is folded into the device-resident state the earlier areas left
    (hspf_rib_fold_device: better replaces, equal merges, the transit-network rule), in ONE instance-wide first-hop slot
    numbering (area a's slots start at a word offset), so that a RIB row can carry next hops of several areas and the
    comparison sees one table.  `version` 2 / 3 (OSPFv3: Intra-Area-Prefix-LSAs in LSDB order, `af`).  `other_rows`: the
    inter-area / external rows of the new RIB (calculations outside this path): compared on the host and merged into the
    sequence in prefix order.  Returns (messages, records copied, prefixes compared)."""
    import ipaddress
    import torch
    if version == 3:
        from . import ospfv3 as V
        mk_graph = lambda area: V.AreaGraph3(area, af)                                   # noqa: E731
        mk_table, calc = Ospfv3PrefixTable.build, V.calc_nexthops
    else:
        V = O
        mk_graph, mk_table, calc = O.AreaGraph, Ospfv2OrderedTable.build, O.calc_nexthops
    dev = torch.device(device)
    old_intra = {O._net_key(r["prefix"]): r for r in rib_before if r.get("type", "intra-area") == "intra-area"}
    old_other = [r for r in rib_before if r.get("type", "intra-area") != "intra-area"]
    # ---- every area that holds the root's Router-LSA: SPT on the device, its table, its first-hop slots
    infos, word_off = [], 0
    for area in sorted(areas, key=lambda a: V.ip(a.area_id)):
        g = mk_graph(area)
        root = g.index.get((V.RTR, V.ip(router_id)))
        if root is None:
            continue
        table = mk_table(g)
        roots = np.asarray([root], np.uint32)
        n = len(g.vids)
        G = g.device(engine)
        W = G.mask_words(roots)
        dist = torch.empty((1, n), dtype=torch.int32, device=dev); hops = torch.empty((1, n), dtype=torch.int16, device=dev)
        flags = torch.empty((1, n), dtype=torch.int16, device=dev); mask = torch.empty((1, n, W), dtype=torch.int64, device=dev)
        stats = engine.run_device(G, roots, E.RUN_NET_NEXTHOPS, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(),
                                  flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
        # first-hop slots -> next hops (needs Interface / Neighbor objects: host, once per slot)
        res = E.SpfResult(dist.cpu().numpy().view(np.uint32), hops.cpu().numpy().view(np.uint16),
                          flags.cpu().numpy().view(np.uint16), mask.cpu().numpy().view(np.uint64), None, stats)
        slot_local: dict = {}
        O.spt_from_engine(g, root, engine, calc, res=res, slots_out=slot_local)
        infos.append(dict(g=g, n=n, table=table, W=W, off=word_off, dist=dist, flags=flags, mask=mask, slot_nh=slot_local))
        word_off += W
    if not infos or not any(i["table"].prefixes for i in infos):
        for i in infos:
            if i["g"]._dev is not None:
                i["g"]._dev[1].free()
        return O.update_global_rib(list(other_rows), rib_before, ifindex), 0, 0
    W = max(word_off, 1)
    # instance-wide first-hop slots: area a's slot s is slot 64 * off_a + s
    slot_nh = {64 * i["off"] + s: nh for i in infos for s, nh in i["slot_nh"].items()}
    table_keys = {O._net_key(p) for i in infos for p in i["table"].prefixes}
    # ONE prefix list for both sides: the tables' prefixes plus those only the old RIB knows (no entries: no new route)
    keys = {O._net_key(p): p for i in infos for p in i["table"].prefixes}
    for k, r in old_intra.items():
        keys.setdefault(k, r["prefix"])
    order = sorted(keys)
    prefixes = [keys[k] for k in order]
    P = len(prefixes)
    where = {k: i for i, k in enumerate(order)}
    # ---- the fold, area after area, on the device
    bm = torch.empty((1, P), dtype=torch.int32, device=dev); be = torch.empty((1, P), dtype=torch.int32, device=dev)
    nm = torch.empty((1, P, W), dtype=torch.int64, device=dev); org = torch.empty((P,), dtype=torch.int32, device=dev)
    ribkw = dict(n_prefixes=P, mask_words=W, best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr(), origin_ptr=org.data_ptr())
    engine.rib_clear_device(P, W, best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr(), origin_ptr=org.data_ptr())
    for ai, i in enumerate(infos):
        t = i["table"]
        if not t.prefixes:
            continue
        pmap = np.asarray([where[O._net_key(p)] for p in t.prefixes], np.uint32)
        engine.rib_fold_device(i["n"], i["W"], i["dist"].data_ptr(), i["flags"].data_ptr(), i["mask"].data_ptr(), t.pfx_ptr, t.pfx_vertex,
                               t.pfx_metric, t.pfx_origin, pmap, ai, i["off"], **ribkw)
    slot_sets = {s: {(addr, name) for (_i, _a), (name, addr) in nh.items()} for s, nh in slot_nh.items() if nh}
    # the OLD RIB in the same index space: metric, and the slots whose next hops the old route used; a next hop no slot
    # resolves to any more, or more next hops than max-paths allows, cannot be expressed: the metric is poisoned so that the
    # pair compares unequal and the host decides on that record
    om = np.full((1, P), 0xFFFFFFFF, np.uint32); oe = np.full((1, P), 0xFFFFFFFF, np.uint32); on = np.zeros((1, P, W), np.uint64)
    for k, r in old_intra.items():
        i = where[k]
        want = {(a, ifn) for a, ifn in r["nexthops"]}
        seen = set()
        for s, nhs in slot_sets.items():
            if nhs <= want:
                on[0, i, s // 64] |= np.uint64(1) << np.uint64(s % 64)
                seen |= nhs
        om[0, i] = r["metric"] if seen == want and len(want) <= max_paths else 0xFFFFFFFE
        oe[0, i] = 0
        if want and not on[0, i].any():
            on[0, i, 0] = np.uint64(1)
    t_om = torch.from_numpy(om.view(np.int32)).to(dev); t_oe = torch.from_numpy(oe.view(np.int32)).to(dev)
    t_on = torch.from_numpy(on.view(np.int64)).to(dev)
    act = torch.empty((1, P), dtype=torch.uint8, device=dev)
    chg = torch.empty((P,), dtype=torch.int32, device=dev); cptr = torch.empty((2,), dtype=torch.int32, device=dev)
    newp = (bm.data_ptr(), be.data_ptr(), nm.data_ptr())
    engine.routes_diff_device(1, P, W, (t_om.data_ptr(), t_oe.data_ptr(), t_on.data_ptr()), newp,
                              action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
    rec = engine.routes_pack(1, P, W, newp, action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
    for i in infos:
        if i["g"]._dev is not None:
            i["g"]._dev[1].free()

    def installed(nhs) -> bool:
        return any(a is not None for a, _ in nhs)
    walk, gone = {}, {}                  # messages of the walk over the new RIB (prefix order) / of the routes that vanished
    # The reference compares old and new route of a PREFIX whatever their types (route.rs:870-885): a prefix that moves between
    # intra-area and inter-area / external with the same metric, tag, label and next hops is "unchanged" — no message (ADVICE
    # r04: the type split of the two paths below used to see an install / a brand-new prefix there).
    old_any = {O._net_key(r["prefix"]): r for r in rib_before}
    new_other_keys = {O._net_key(r["prefix"]) for r in other_rows}
    for r in rec:
        p, action, metric, entry = int(r[1]), int(r[2]), int(r[3]), int(r[4])
        prefix = prefixes[p]
        o = old_intra.get(order[p])
        if o is None:                                                 # held under another type before: still the same route to the reference
            o2 = old_any.get(order[p])
            if o2 is not None and o2.get("tag") is None and o2.get("sr_label") is None:
                o = o2
        if action == E.DIFF_WITHDRAW:
            if entry == 0xFFFFFFFF and o is not None and installed(o["nexthops"]):
                gone[order[p]] = {"op": "del", "prefix": prefix}
            continue
        if action not in (E.DIFF_INSTALL, E.DIFF_SILENT):
            continue
        nhs = {}
        for w in range(W):
            m = int(r[6 + 2 * w]) | (int(r[7 + 2 * w]) << 32)
            while m:
                b = (m & -m).bit_length() - 1
                m &= m - 1
                nhs.update(slot_nh.get(w * 64 + b) or {})
        keep = [[nhs[k][1], nhs[k][0]] for k in sorted(nhs)[:max_paths]]        # rows carry [addr, iface name]
        if o is not None and o["metric"] == metric and sorted(map(tuple, o["nexthops"]), key=str) == sorted(map(tuple, keep), key=str):
            continue                                                             # the reference's "unchanged" (:875-885)
        if installed(keep):
            wire = sorted(((ifindex[ifn], a) for a, ifn in keep if a is not None), key=lambda t: (t[0], int(ipaddress.ip_address(t[1]))))
            walk[order[p]] = {"op": "add", "prefix": prefix, "metric": metric, "nexthops": [list(t) for t in wire]}
        # (a route that turns CONNECTED / loses its next hops: the reference's `else if INSTALLED` branch (:902-905) looks at
        # the NEW route object, which never carries the flag when it changed — nothing goes out, and the old route has left
        # old_rib at :870, so no withdrawal follows either; the host twin, pinned to the recorded messages, does the same)
    # the rows of the other route types: the host rule among themselves (route.rs:856-916 does not look at the type); a
    # prefix that changed its type is one route to the reference: the new row's message stands, no withdrawal
    old_for_other = old_other + [old_intra[k] for k in new_other_keys if k in old_intra]      # (the same rule the other way round)
    for m in O.update_global_rib(list(other_rows), old_for_other, ifindex):
        k = O._net_key(m["prefix"])
        if m["op"] == "add":
            walk[k] = m
            gone.pop(k, None)
        elif k not in walk and k not in table_keys:
            gone[k] = m
    for k in list(gone):
        if k in walk or k in new_other_keys:          # the prefix lives on (under whatever type): one route, never withdrawn
            del gone[k]
    msgs = [walk[k] for k in sorted(walk)] + [gone[k] for k in sorted(gone)]
    return msgs, len(rec), P


# ---- OSPFv3: update_rib_intra_area with the ORDERED prefix fold on the GPU -------------------------------------------

@dataclass
class Ospfv3PrefixTable:
    """CSR-by-prefix table of one OSPFv3 area (root independent) for HSPF_PFX_ORDERED: the stub networks as
    `Ospfv3::intra_area_networks` yields them (holo-ospf/src/ospfv3/spf.rs:421-478) — Intra-Area-Prefix-LSAs in LSDB
    order (adv_rtr, LS-ID), MaxAge skipped, the referenced vertex looked up by (ref type, ref LS-ID, ref adv_rtr),
    NU-bit prefixes dropped — grouped by prefix, the reference's order kept inside a prefix.  Entries whose referenced
    vertex is not in the graph at all can never be in an SPT and are left out."""
    prefixes: List[str]
    pfx_ptr: np.ndarray
    pfx_vertex: np.ndarray            # vertex | PFX_ENTRY_NETWORK
    pfx_metric: np.ndarray
    pfx_origin: np.ndarray            # `stub.vertex.lsa.origin().lsa_id`

    @classmethod
    def build(cls, g) -> "Ospfv3PrefixTable":
        from . import ospfv3 as V3
        rows = []
        for lsa in sorted(g.area.iaps, key=lambda l: (V3.ip(l.adv_rtr), l.lsa_id)):
            if lsa.maxage:
                continue
            if lsa.ref_type == "ospfv3-router-lsa":
                v = g.index.get((V3.RTR, V3.ip(lsa.ref_adv_rtr))) if lsa.ref_lsa_id == 0 else None
            elif lsa.ref_type == "ospfv3-network-lsa":
                v = g.index.get((V3.NET, V3.ip(lsa.ref_adv_rtr), lsa.ref_lsa_id))
            else:
                v = None
            if v is None:
                continue
            is_net = g.vids[v][0] == V3.NET
            vl = g.lsa_of(v)
            origin = vl.lsa_id if is_net else vl[0].lsa_id
            for p in lsa.prefixes:
                if "nu-bit" in p["options"]:
                    continue
                rows.append((V3._net_key(p["prefix"]), p["prefix"], v | (E.PFX_ENTRY_NETWORK if is_net else 0), p["metric"], origin))
        keys = sorted({r[0]: r[1] for r in rows}.items())
        pid = {k: i for i, (k, _) in enumerate(keys)}
        order = sorted(range(len(rows)), key=lambda i: (pid[rows[i][0]], i))          # stable: the reference's order
        ptr = np.zeros(len(keys) + 1, np.uint64)
        for r in rows:
            ptr[pid[r[0]] + 1] += 1
        return cls([p for _, p in keys], np.cumsum(ptr).astype(np.uint32),
                   np.asarray([rows[i][2] for i in order], np.uint32), np.asarray([rows[i][3] for i in order], np.uint32),
                   np.asarray([rows[i][4] for i in order], np.uint32))


def ospfv3_area_device_routes(engine, g, root: int, table: Ospfv3PrefixTable, rib: dict, max_paths: int,
                              device="cuda:0") -> None:
    """run_area + update_rib_intra_area of one OSPFv3 area with the SPT and the ordered prefix fold on the device.  `rib`
    is shared by the areas (holo-ospf/src/route.rs:146-160): what the earlier areas left for a prefix goes in as the
    fold's initial state (init_*), so the result is the reference's sequential fold, not an approximation of it."""
    import torch
    from . import ospfv3 as V3
    G = g.device(engine)
    roots = np.asarray([root], np.uint32)
    n, W = len(g.vids), G.mask_words(roots)
    dev = torch.device(device)
    dist = torch.empty((1, n), dtype=torch.int32, device=dev)
    hops = torch.empty((1, n), dtype=torch.int16, device=dev)
    flags = torch.empty((1, n), dtype=torch.int16, device=dev)
    mask = torch.empty((1, n, W), dtype=torch.int64, device=dev)
    stats = engine.run_device(G, roots, E.RUN_NET_NEXTHOPS, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(),
                              flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
    P = len(table.prefixes)
    if P == 0:
        return
    keys = [V3._net_key(p) for p in table.prefixes]
    iex = np.asarray([1 if k in rib else 0 for k in keys], np.uint8)
    imet = np.asarray([rib[k]["metric"] if k in rib else 0 for k in keys], np.uint32)
    iorg = np.asarray([rib[k]["origin"] if k in rib else 0 for k in keys], np.uint32)
    bm = torch.empty((1, P), dtype=torch.int32, device=dev)
    be = torch.empty((1, P), dtype=torch.int32, device=dev)
    nm = torch.empty((1, P, W), dtype=torch.int64, device=dev)
    engine.routes_device(n, 1, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), table.pfx_ptr, table.pfx_vertex,
                         table.pfx_metric, best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(),
                         nexthop_mask_ptr=nm.data_ptr(), flags=E.PFX_SATURATING | E.PFX_ORDERED, pfx_origin=table.pfx_origin,
                         init_exists=iex, init_metric=imet, init_origin=iorg)
    torch.cuda.synchronize(dev)
    bm = bm.cpu().numpy().view(np.uint32)[0]; be = be.cpu().numpy().view(np.uint32)[0]; nm = nm.cpu().numpy().view(np.uint64)[0]
    res = E.SpfResult(dist.cpu().numpy().view(np.uint32), hops.cpu().numpy().view(np.uint16),
                      flags.cpu().numpy().view(np.uint16), mask.cpu().numpy().view(np.uint64), None, stats)
    slot_nh: dict = {}
    O.spt_from_engine(g, root, engine, V3.calc_nexthops, res=res, slots_out=slot_nh)

    def expand(mrow) -> dict:
        nhs = {}
        for w in range(len(mrow)):
            m = int(mrow[w])
            while m:
                b = (m & -m).bit_length() - 1
                m &= m - 1
                nh = slot_nh.get(w * 64 + b)
                if nh:
                    nhs.update(nh)
        return nhs

    for p, prefix in enumerate(table.prefixes):
        key = keys[p]
        if be[p] == 0xFFFFFFFF:
            continue                                        # no entry in the SPT, nothing in the RIB
        if be[p] == E.PFX_KEPT_INIT:
            cur = rib[key]
            cur["nexthops"].update(expand(nm[p]))           # equal-cost entries of this area merged into the earlier route
        else:
            cur = rib[key] = {"prefix": prefix, "metric": int(bm[p]), "origin": int(table.pfx_origin[be[p]]),
                              "nexthops": expand(nm[p])}
        if len(cur["nexthops"]) > max_paths:
            cur["nexthops"] = {k: cur["nexthops"][k] for k in sorted(cur["nexthops"])[:max_paths]}


def ospfv3_intra_area_device_routes(router_id: str, areas, max_paths: int, engine, af: str = "ipv6",
                                    device="cuda:0") -> List[dict]:
    """holo_amd.ospfv3.compute_spf_intra_area with SPT and prefix attachment on the GPU; same rows."""
    from . import ospfv3 as V3
    rib: dict = {}
    for area in sorted(areas, key=lambda a: V3.ip(a.area_id)):
        g = V3.AreaGraph3(area, af)
        root = g.index.get((V3.RTR, V3.ip(router_id)))
        if root is None:
            continue
        ospfv3_area_device_routes(engine, g, root, Ospfv3PrefixTable.build(g), rib, max_paths, device)
        if g._dev is not None:
            g._dev[1].free()
    return [{"prefix": rib[k]["prefix"], "metric": rib[k]["metric"], "type": "intra-area",
             "nexthops": [[rib[k]["nexthops"][a][1], rib[k]["nexthops"][a][0]] for a in sorted(rib[k]["nexthops"])]}
            for k in sorted(rib)]
