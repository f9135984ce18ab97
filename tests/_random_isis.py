"""Random IS-IS instances in the schema of tests/golden/isis/*.json (tools/make_golden.py), for differential tests of the
host twin (holo_amd/isis.py: LSDB -> CSR, slot replay through resolve_nexthop, route build) against the literal
restatement (oracle/isis_ref.py).  TEST INFRASTRUCTURE ONLY.

What is varied: router count, p2p links (also parallel ones between the same pair, with equal or different metrics),
LANs with a pseudonode LSP, metric type (standard / wide / both), overload and attached bits, missing or partial
protocols-supported TLVs, expired LSPs, one-way adjacencies in the LSDB, fragments, shared prefixes (ECMP across
advertisers), the local router's interfaces in name order (p2p metric must equal the link cost, holo-isis/src/spf.rs:988-990)."""
import numpy as np


def sid(i):
    return f"0000.0000.{i:04x}"


def make(seed: int, zero: bool = False) -> dict:
    """zero=True: about a third of the link metrics are 0 (holo-isis/src/spf.rs:629-704 treats 0 like any metric; YANG allows it): the
    pop order of many roots is then dynamic — the engine's k_repair / pop ranks and the twins' slot replay in that order."""
    rng = np.random.default_rng(seed)
    _rand_metric = rng.integers

    def draw(lo, hi):
        m = int(_rand_metric(lo, hi))
        return 0 if (zero and rng.random() < 0.35) else m
    n = int(rng.integers(3, 13))
    mtype = str(rng.choice(["standard", "wide", "both"]))
    maxm = 60 if mtype != "wide" else 5000
    local = int(rng.integers(1, n + 1))
    # topology
    p2p = []                                     # (a, b, metric a->b, metric b->a)
    for a in range(1, n + 1):
        for b in range(a + 1, n + 1):
            if rng.random() < min(1.0, 2.2 / n):
                reps = 2 if rng.random() < 0.25 else 1
                m = draw(1, maxm)
                for _ in range(reps):
                    same = rng.random() < 0.6
                    p2p.append((a, b, m, m if rng.random() < 0.7 else draw(1, maxm)))
                    if not same:
                        m = draw(1, maxm)
    for a in range(1, n):                        # keep it mostly connected
        if not any(x[:2] in ((a, a + 1),) for x in p2p) and rng.random() < 0.8:
            m = draw(1, maxm); p2p.append((a, a + 1, m, m))
    lans = []                                    # (dis, pn id, members, metric per member)
    for k in range(int(rng.integers(0, 3))):
        size = int(rng.integers(2, min(n, 6) + 1))
        members = sorted(rng.choice(np.arange(1, n + 1), size=size, replace=False).tolist())
        dis = int(rng.choice(members))
        lans.append((dis, k + 1, members, {m: draw(1, maxm) for m in members}))
    # LSPs
    lsps = []
    std_on, wide_on = mtype in ("standard", "both"), mtype in ("wide", "both")
    for r in range(1, n + 1):
        nbrs = []
        for a, b, mab, mba in p2p:
            if a == r and rng.random() > 0.04: nbrs.append((f"{sid(b)}.00", mab))
            if b == r and rng.random() > 0.04: nbrs.append((f"{sid(a)}.00", mba))     # 4 %: one-way in the LSDB
        for dis, pn, members, met in lans:
            if r in members: nbrs.append((f"{sid(dis)}.{pn:02x}", met[r]))
        order = rng.permutation(len(nbrs)).tolist()
        nbrs = [nbrs[i] for i in order]
        flags = (["ol"] if rng.random() < 0.12 else []) + (["att"] if rng.random() < 0.1 else [])
        protos = [204, 142] if rng.random() > 0.1 else ([204] if rng.random() < 0.5 else None)
        pf4 = [[f"{r}.{r}.{r}.{r}/32", int(rng.integers(0, 20))]]
        if rng.random() < 0.6: pf4.append([f"10.{int(rng.integers(0, 4))}.0.0/24", int(rng.integers(0, 20))])     # shared
        pf6 = [[f"2001:db8::{r:x}/128", int(rng.integers(0, 20)), False]]
        if rng.random() < 0.4: pf6.append([f"fc00:{int(rng.integers(0, 3))}::/64", int(rng.integers(0, 20)), bool(rng.random() < 0.2)])
        split = len(nbrs) // 2 if (len(nbrs) > 2 and rng.random() < 0.3) else len(nbrs)      # second fragment
        for frag, part in ((0, nbrs[:split]), (1, nbrs[split:])):
            if frag == 1 and not part:
                continue
            l = {"id": f"{sid(r)}.00-{frag:02x}", "flags": flags if frag == 0 else [], "protocols": protos if frag == 0 else None,
                 "is_reach": [[x, min(m, 63)] for x, m in part] if std_on else [],
                 "ext_is_reach": [[x, m] for x, m in part] if wide_on else [],
                 "mt": [], "mt_is_reach": [], "mt_ipv6": [],
                 "ipv4_int": pf4 if (frag == 0 and std_on) else [], "ipv4_ext": [],
                 "ext_ipv4": [[p, m, False] for p, m in pf4] if (frag == 0 and wide_on) else [],
                 "ipv6": pf6 if frag == 0 else []}
            if rng.random() < 0.04:
                l["lifetime"] = 0                                                               # expired fragment
            lsps.append(l)
    for dis, pn, members, met in lans:
        if rng.random() < 0.93:                                                                 # 7 %: DIS LSP missing
            ms = [m for m in members if rng.random() > 0.05]
            lsps.append({"id": f"{sid(dis)}.{pn:02x}-00", "flags": [], "protocols": None,
                         "is_reach": [[f"{sid(m)}.00", 0] for m in ms] if std_on else [],
                         "ext_is_reach": [[f"{sid(m)}.00", 0] for m in ms] if wide_on else [],
                         "mt": [], "mt_is_reach": [], "mt_ipv6": [], "ipv4_int": [], "ipv4_ext": [], "ext_ipv4": [], "ipv6": []})
    # the local router's interfaces
    ifaces, k = [], 0
    for a, b, mab, mba in p2p:
        if local in (a, b):
            other, m = (b, mab) if a == local else (a, mba)
            k += 1
            ifaces.append({"name": f"eth{int(rng.integers(0, 50)):02d}-{k}", "type": "point-to-point",
                           "metric": {"1": min(m, 63) if mtype == "standard" else m, "2": min(m, 63) if mtype == "standard" else m},
                           "adjacencies": [{"system_id": sid(other), "usage": "level-2", "state": "up" if rng.random() > 0.05 else "down",
                                            "ipv4": [f"10.{local}.{k}.{other}"], "ipv6": [f"fe80::{local:x}:{k:x}:{other:x}"],
                                            "topologies": [0], "area_addrs": ["49.0000"]}]})
    for dis, pn, members, met in lans:
        if local in members:
            k += 1
            ifaces.append({"name": f"lan{int(rng.integers(0, 50)):02d}-{k}", "type": "broadcast", "metric": {"1": met[local], "2": met[local]},
                           "adjacencies": [{"system_id": sid(m), "usage": "level-2", "state": "up",
                                            "ipv4": [f"172.16.{pn}.{m}"], "ipv6": [f"fe80::aa:{pn:x}:{m:x}"],
                                            "topologies": [0], "area_addrs": ["49.0000"]} for m in members if m != local]})
    ifaces.append({"name": "lo", "type": "broadcast", "metric": {"1": 10, "2": 10}, "adjacencies": []})
    return {"proto": "isis", "source": f"random instance {seed}",
            "config": {"system_id": sid(local), "level_type": "level-2", "metric_type": {"1": mtype, "2": mtype},
                       "afs": {"ipv4": bool(rng.random() > 0.1), "ipv6": bool(rng.random() > 0.2)}, "mt_ipv6_unicast": False,
                       "max_paths": int(rng.choice([1, 2, 16])), "att_ignore": bool(rng.random() < 0.2), "area_addrs": ["49.0000"]},
            "interfaces": ifaces, "lsdb": {"2": lsps}, "rib": []}


def add_sr(vec: dict, seed: int) -> dict:
    """Segment-routing data on top of make(seed) (its own generator: make() is unchanged): sr_enabled, per router an
    SR-Capabilities entry (I / V flags, one or two SRGB ranges; sometimes missing), the SR-Algorithm list (sometimes
    without SPF), Prefix-SIDs of algorithm SPF on wide IPv4 and on IPv6 entries (index — sometimes beyond the SRGB — or
    absolute label; P / E flags).  Shared prefixes get different SIDs from different advertisers."""
    import copy
    rng = np.random.default_rng(900_000 + seed)
    v = copy.deepcopy(vec)
    v["config"]["sr_enabled"] = True
    for l in v["lsdb"]["2"]:
        lan, frag = l["id"].rsplit("-", 1)
        if not lan.endswith(".00"):
            continue                                           # pseudonode LSP
        r = int(lan.split(".")[2], 16)
        if frag == "00":
            if rng.random() < 0.88:
                flags = ["I", "V"] if rng.random() < 0.75 else ([["I"], ["V"], []][int(rng.integers(0, 3))])
                srgb = [[16000 + 1000 * r, int(rng.integers(20, 120))]]
                if rng.random() < 0.3:
                    srgb.append([40000 + 1000 * r, int(rng.integers(20, 120))])
                l["sr_cap"] = {"flags": flags, "srgb": srgb}
            l["sr_algos"] = [0] if rng.random() < 0.9 else ([1] if rng.random() < 0.5 else [])
        sids = {}
        for kind in ("ext_ipv4", "ipv6"):
            for i in range(len(l[kind])):
                if rng.random() < 0.75:
                    fl = [f for f in ("P", "E") if rng.random() < 0.35]
                    if rng.random() < 0.15:
                        sids.setdefault(kind, {})[str(i)] = {"flags": fl + ["V", "L"], "label": int(rng.integers(5000, 6000))}
                    else:
                        sids.setdefault(kind, {})[str(i)] = {"flags": fl, "index": int(rng.integers(0, 150))}
        if sids:
            l["prefix_sids"] = sids
    return v


def make_two_level(seed: int) -> dict:
    """A level-all instance: the level-2 LSDB of make(seed) and a level-1 LSDB derived from it (a sub-area: some routers left
    out, LSP-level changes, other prefix metrics, ATT bits on some zeroth LSPs), adjacencies usable on one or both levels,
    foreign area addresses on some (is_l2_attached_to_backbone) — for the L1 / L2 merge of holo-isis/src/route.rs:185-249 and
    the default route of attached L2 routers (spf.rs:1175-1190)."""
    import copy
    from test_host_isis_random import mutate
    rng = np.random.default_rng(seed + 777)
    v = make(seed)
    l1 = copy.deepcopy(v["lsdb"]["2"])
    drop = {l["id"][:14] for l in l1 if rng.random() < 0.15 and l["id"][:14] != v["config"]["system_id"]}
    tmp = dict(v)
    tmp["lsdb"] = {"2": [l for l in l1 if l["id"][:14] not in drop]}
    for _ in range(3):
        tmp = mutate(tmp, rng)
    l1 = tmp["lsdb"]["2"]
    for l in l1:
        if l["id"].endswith(".00-00"):
            if rng.random() < 0.3 and "att" not in l["flags"]:
                l["flags"] = l["flags"] + ["att"]
            for key in ("ipv4_int", "ext_ipv4", "ipv6"):
                for e in l[key]:
                    if rng.random() < 0.4:
                        e[1] = int(rng.integers(0, 25))
    v["lsdb"] = {"1": l1, "2": v["lsdb"]["2"]}
    v["config"]["level_type"] = "level-all"
    for f in v["interfaces"]:
        for a in f["adjacencies"]:
            a["usage"] = str(rng.choice(["level-all", "level-all", "level-1", "level-2"]))
            if rng.random() < 0.25:
                a["area_addrs"] = ["49.0002"]
    v["source"] = f"random two-level instance {seed}"
    return v


def make_mt(seed: int) -> dict:
    """MT IPv6-unicast on: a second topology (MT id 2) with its own adjacency entries (a subset of the links, other metrics), its
    own prefixes (TLV 237) and per-topology overload / attached bits, an entry of a topology that is not enabled (ignored), the
    local router's adjacencies in one or both topologies (spf.rs:1013-1128, 1149-1296 with mt_id = Ipv6Unicast)."""
    rng = np.random.default_rng(seed + 4242)
    v = make(seed)
    v["config"]["mt_ipv6_unicast"] = True
    v["config"]["afs"]["ipv6"] = True
    for l in v["lsdb"]["2"]:
        src = l["ext_is_reach"] or l["is_reach"]
        l["mt_is_reach"] = [[2, nbr, int(rng.integers(1, 60)) if rng.random() < 0.5 else m] for nbr, m in src if rng.random() > 0.15]
        if l["id"].endswith("-00"):
            fl = (["ol"] if rng.random() < 0.1 else []) + (["att"] if rng.random() < 0.1 else [])
            l["mt"] = [{"id": 0, "flags": []}, {"id": 2, "flags": fl}] if rng.random() > 0.1 else [{"id": 0, "flags": []}]
            l["mt_ipv6"] = [[2, p, int(rng.integers(0, 20)), bool(x)] for p, m, x in l["ipv6"]]
            if rng.random() < 0.3:
                l["mt_ipv6"].append([2, f"fc07:{int(rng.integers(0, 3))}::/64", int(rng.integers(0, 20)), False])
            if rng.random() < 0.2:
                l["mt_ipv6"].append([3, "fc08::/64", 1, False])
    for f in v["interfaces"]:
        for a in f["adjacencies"]:
            a["topologies"] = [0, 2] if rng.random() > 0.2 else ([0] if rng.random() < 0.7 else [2])
    v["source"] = f"random MT instance {seed}"
    return v


def make_long(seed: int) -> dict:
    """A long chain of routers with narrow metrics near their maximum (63): path metrics cross MAX_PATH_METRIC_STANDARD = 1023
    (spf.rs:637-641) a dozen and a half hops out — vertices beyond it must stay off the SPT; a few chords, some with a cheap way back."""
    rng = np.random.default_rng(seed + 99)
    n = int(rng.integers(22, 31))
    mtype = "standard"      # ("both" lists every adjacency twice: the reference's next-hop list then doubles per hop, 2^29 entries at the far end)
    local = int(rng.integers(1, 4))
    links = []
    for a in range(1, n):
        m = int(rng.integers(45, 64)); links.append((a, a + 1, m, m if rng.random() < 0.8 else int(rng.integers(45, 64))))
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(1, n - 3)); b = int(rng.integers(a + 2, min(n, a + 6) + 1))
        m = int(rng.integers(30, 64)); links.append((a, b, m, m))
    std_on, wide_on = True, mtype == "both"
    lsps = []
    for r in range(1, n + 1):
        nbrs = []
        for a, b, mab, mba in links:
            if a == r: nbrs.append((f"{sid(b)}.00", mab))
            if b == r: nbrs.append((f"{sid(a)}.00", mba))
        pf4 = [[f"{r}.{r}.{r}.{r}/32", int(rng.integers(0, 20))]]
        if rng.random() < 0.5: pf4.append([f"10.{int(rng.integers(0, 3))}.0.0/24", int(rng.integers(0, 40))])
        pf6 = [[f"2001:db8::{r:x}/128", int(rng.integers(0, 20)), False]]
        lsps.append({"id": f"{sid(r)}.00-00", "flags": [], "protocols": [204, 142],
                     "is_reach": [[x, m] for x, m in nbrs] if std_on else [], "ext_is_reach": [[x, m] for x, m in nbrs] if wide_on else [],
                     "mt": [], "mt_is_reach": [], "mt_ipv6": [], "ipv4_int": pf4, "ipv4_ext": [],
                     "ext_ipv4": [[p, m, False] for p, m in pf4] if wide_on else [], "ipv6": pf6})
    ifaces, k = [], 0
    for a, b, mab, mba in links:
        if local in (a, b):
            other, m = (b, mab) if a == local else (a, mba)
            k += 1
            ifaces.append({"name": f"eth{k:02d}", "type": "point-to-point", "metric": {"1": m, "2": m},
                           "adjacencies": [{"system_id": sid(other), "usage": "level-2", "state": "up", "ipv4": [f"10.{local}.{k}.{other}"],
                                            "ipv6": [f"fe80::{local:x}:{k:x}:{other:x}"], "topologies": [0], "area_addrs": ["49.0000"]}]})
    ifaces.append({"name": "lo", "type": "broadcast", "metric": {"1": 10, "2": 10}, "adjacencies": []})
    return {"proto": "isis", "source": f"random long chain {seed}",
            "config": {"system_id": sid(local), "level_type": "level-2", "metric_type": {"1": mtype, "2": mtype},
                       "afs": {"ipv4": True, "ipv6": True}, "mt_ipv6_unicast": False, "max_paths": int(rng.choice([1, 2, 16])),
                       "att_ignore": False, "area_addrs": ["49.0000"]},
            "interfaces": ifaces, "lsdb": {"2": lsps}, "rib": []}
