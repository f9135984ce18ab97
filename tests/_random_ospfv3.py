"""Random single-area OSPFv3 LSDBs in the schema of tests/golden/ospfv3/*.json, for differential tests of the host twin
(holo_amd/ospfv3.py) against the literal restatement (oracle/ospfv3_ref.py).  TEST INFRASTRUCTURE ONLY.

Varied: routers (some split over two Router-LSA fragments, some without R-bit / V6-bit), point-to-point links (also
parallel), transit networks (Network-LSA keyed (DR router id, DR interface id)), one-way links, Intra-Area-Prefix LSAs
referencing routers and networks (NU-bit prefixes, non-zero ref LS-ID), Link-LSAs with the neighbours' link-local
addresses on the local router's interfaces (missing ones included), costs, max-paths."""
import numpy as np


def rid(i):
    return f"{i}.{i}.{i}.{i}"


def make(seed: int) -> dict:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 11))
    local = int(rng.integers(1, n + 1))
    hi = int(rng.choice([3, 10, 60]))
    next_if = {r: 1 for r in range(1, n + 1)}

    def new_if(r):
        next_if[r] += 1
        return next_if[r]
    links = {r: [] for r in range(1, n + 1)}
    ifaces, nets, iaps = [], [], []
    arena = 0
    for a in range(1, n + 1):
        for b in range(a + 1, n + 1):
            if rng.random() < min(1.0, 2.0 / n) or (b == a + 1 and rng.random() < 0.7):
                for _ in range(2 if rng.random() < 0.2 else 1):
                    ia, ib = new_if(a), new_if(b)
                    ma = int(rng.integers(1, hi + 1)); mb = ma if rng.random() < 0.7 else int(rng.integers(1, hi + 1))
                    links[a].append({"type": "point-to-point-link", "iface_id": ia, "nbr_iface_id": ib, "nbr_router_id": rid(b), "metric": ma})
                    if rng.random() > 0.05:
                        links[b].append({"type": "point-to-point-link", "iface_id": ib, "nbr_iface_id": ia, "nbr_router_id": rid(a), "metric": mb})
                    for me, other, i_me, i_ot in ((a, b, ia, ib), (b, a, ib, ia)):
                        if me == local:
                            ll = [{"adv_rtr": rid(me), "lsa_id": i_me, "lladdr": f"fe80::{me:x}:{i_me:x}"}]
                            if rng.random() > 0.05:
                                ll.append({"adv_rtr": rid(other), "lsa_id": i_ot, "lladdr": f"fe80::{other:x}:{i_ot:x}"})
                            arena += 1
                            ifaces.append({"name": f"eth{int(rng.integers(0, 90)):02d}-{arena}", "type": "point-to-point", "iface_id": i_me,
                                           "index": arena, "link_lsas": ll, "neighbors": [{"router_id": rid(other), "src": f"fe80::{other:x}:{i_ot:x}"}]})
    for k in range(int(rng.integers(0, 3))):
        size = int(rng.integers(2, min(n, 5) + 1))
        members = sorted(rng.choice(np.arange(1, n + 1), size=size, replace=False).tolist())
        dr = int(rng.choice(members))
        ifid = {m: new_if(m) for m in members}
        nets.append({"adv_rtr": rid(dr), "lsa_id": ifid[dr], "attached": [rid(m) for m in members if rng.random() > 0.06]})
        for m in members:
            links[m].append({"type": "transit-network-link", "iface_id": ifid[m], "nbr_iface_id": ifid[dr], "nbr_router_id": rid(dr),
                             "metric": int(rng.integers(1, hi + 1))})
            if m == local:
                arena += 1
                ll = [{"adv_rtr": rid(x), "lsa_id": ifid[x], "lladdr": f"fe80::aa:{x:x}:{ifid[x]:x}"} for x in members if x == m or rng.random() > 0.05]
                ifaces.append({"name": f"lan{int(rng.integers(0, 90)):02d}-{arena}", "type": "broadcast", "iface_id": ifid[m], "index": arena,
                               "link_lsas": ll, "neighbors": [{"router_id": rid(x), "src": f"fe80::aa:{x:x}:{ifid[x]:x}"} for x in members if x != m]})
        iaps.append({"adv_rtr": rid(dr), "lsa_id": 100 + k, "ref_type": "ospfv3-network-lsa", "ref_adv_rtr": rid(dr), "ref_lsa_id": ifid[dr],
                     "prefixes": [{"prefix": f"fc00:{k}::/64", "metric": 0, "options": []}]})
    routers = []
    for r in range(1, n + 1):
        ls = links[r]
        order = rng.permutation(len(ls)).tolist()
        ls = [ls[i] for i in order]
        opts = ["v6-bit", "e-bit", "r-bit", "af-bit"]
        if r != local and rng.random() < 0.06: opts.remove("r-bit")
        if r != local and rng.random() < 0.04: opts.remove("v6-bit")
        split = len(ls) // 2 if (len(ls) > 2 and rng.random() < 0.3) else len(ls)
        routers.append({"adv_rtr": rid(r), "lsa_id": 0, "bits": [], "options": opts, "links": ls[:split]})
        if split < len(ls):
            routers.append({"adv_rtr": rid(r), "lsa_id": 1, "bits": [], "options": opts, "links": ls[split:]})
        px = [{"prefix": f"2001:db8::{r:x}/128", "metric": 0, "options": ["la-bit"]}]
        if rng.random() < 0.6: px.append({"prefix": f"fc00:ff:{int(rng.integers(0, 3))}::/64", "metric": int(rng.integers(1, hi + 1)), "options": []})
        if rng.random() < 0.2: px.append({"prefix": f"fc00:dead:{r:x}::/64", "metric": 1, "options": ["nu-bit"]})
        iaps.append({"adv_rtr": rid(r), "lsa_id": 0, "ref_type": "ospfv3-router-lsa", "ref_adv_rtr": rid(r),
                     "ref_lsa_id": 0 if rng.random() > 0.05 else 7, "prefixes": px})
    idx = rng.permutation(len(ifaces)).tolist()
    for j, i in enumerate(ifaces):
        i["index"] = int(idx[j])
    return {"proto": "ospfv3", "af": "ipv6", "source": f"random area {seed}", "router_id": rid(local),
            "max_paths": int(rng.choice([1, 2, 16])), "has_vlinks": False, "rib": [],
            "areas": [{"area_id": "0.0.0.0", "routers": routers, "networks": nets, "iaps": iaps, "interfaces": ifaces}]}
