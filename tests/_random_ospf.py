"""Random single-area OSPFv2 LSDBs in the schema of tests/golden/ospfv2/*.json (tools/make_golden_ospf.py), for
differential tests of the host twin (holo_amd/ospf.py) against the literal restatement (oracle/ospf_ref.py).
TEST INFRASTRUCTURE ONLY.

Varied: router count, point-to-point links (numbered: one stub link per end; also parallel ones), transit networks with
a DR and a Network-LSA, costs (tie-heavy), one-way links and Network-LSAs that miss a member (two-way check), MaxAge
LSAs, stub-only routers, shared stub prefixes, max-paths.  The local router's interfaces follow its Router-LSA: the
k-th non-stub link belongs to the k-th interface in NAME order that has a neighbour (holo-ospf/src/ospfv2/spf.rs:194-200)."""
import numpy as np


def rid(i):
    return f"{i}.{i}.{i}.{i}"


def make(seed: int, zero: bool = False) -> dict:
    """zero=True: about a third of the p2p link metrics are 0 (a u16 like any other to holo-ospf/src/spf.rs:666-719)."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 12))
    local = int(rng.integers(1, n + 1))
    hi = int(rng.choice([3, 10, 60]))
    links = {r: [] for r in range(1, n + 1)}          # per router: list of dicts (LSA order decided later)
    local_ifaces = []                                 # (sort key name, type, neighbors, addrs, link dict)
    net_lsas = []
    subnet = 0
    for a in range(1, n + 1):
        for b in range(a + 1, n + 1):
            if rng.random() < min(1.0, 2.0 / n) or b == a + 1 and rng.random() < 0.7:
                for _ in range(2 if rng.random() < 0.2 else 1):
                    subnet += 1
                    aa, ab = f"10.{subnet // 250}.{subnet % 250}.1", f"10.{subnet // 250}.{subnet % 250}.2"
                    ma, mb = int(rng.integers(1, hi + 1)), int(rng.integers(1, hi + 1))
                    if zero:
                        ma, mb = (0 if rng.random() < 0.35 else ma), (0 if rng.random() < 0.35 else mb)
                    if rng.random() < 0.7: mb = ma
                    la = {"type": "point-to-point-link", "id": rid(b), "data": aa, "metric": ma}
                    lb = {"type": "point-to-point-link", "id": rid(a), "data": ab, "metric": mb}
                    stub = f"10.{subnet // 250}.{subnet % 250}.0"
                    links[a] += [la, {"type": "stub-network-link", "id": stub, "data": "255.255.255.252", "metric": ma}]
                    if rng.random() > 0.05:                                  # 5 %: the other end does not list the link
                        links[b] += [lb, {"type": "stub-network-link", "id": stub, "data": "255.255.255.252", "metric": mb}]
                    if local == a: local_ifaces.append((la, "point-to-point", [{"router_id": rid(b), "src": ab}], [aa + "/30"]))
                    if local == b: local_ifaces.append((lb, "point-to-point", [{"router_id": rid(a), "src": aa}], [ab + "/30"]))
    for k in range(int(rng.integers(0, 3))):
        size = int(rng.integers(2, min(n, 5) + 1))
        members = sorted(rng.choice(np.arange(1, n + 1), size=size, replace=False).tolist())
        dr = int(rng.choice(members))
        base = f"172.16.{k}"
        addr = {m: f"{base}.{m}" for m in members}
        attached = [rid(m) for m in members if rng.random() > 0.06]          # 6 %: a member missing from the Network-LSA
        net_lsas.append({"lsa_id": addr[dr], "adv_rtr": rid(dr), "mask": "255.255.255.0", "attached": attached,
                         "maxage": bool(rng.random() < 0.05)})
        for m in members:
            l = {"type": "transit-network-link", "id": addr[dr], "data": addr[m], "metric": int(rng.integers(1, hi + 1))}
            links[m].append(l)
            if m == local:
                local_ifaces.append((l, "broadcast", [{"router_id": rid(x), "src": addr[x]} for x in members if x != m], [addr[m] + "/24"]))
    routers = []
    ifaces = []
    for r in range(1, n + 1):
        ls = links[r]
        if r == local:
            # name order of the interfaces == order of the non-stub links in the Router-LSA
            non_stub = [l for l in ls if l["type"] != "stub-network-link"]
            stubs = [l for l in ls if l["type"] == "stub-network-link"]
            rng.shuffle(non_stub)
            ls = []
            for pos, l in enumerate(non_stub):
                ls.append(l)
                if stubs and rng.random() < 0.7: ls.append(stubs.pop())     # stub links interleaved: they take no position
                typ, nbrs, addrs = next((t, nb, ad) for (ll, t, nb, ad) in local_ifaces if ll is l)
                ifaces.append({"name": f"eth{pos:02d}", "type": typ, "state": typ, "index": int(rng.integers(0, 1000)) * 100 + pos,
                               "neighbors": nbrs, "addrs": addrs})
            ls += stubs
            ifaces.append({"name": "aaa-no-neighbors", "type": "broadcast", "state": "dr", "index": 99999, "neighbors": [], "addrs": ["192.168.0.1/24"]})
        else:
            order = rng.permutation(len(ls)).tolist()
            ls = [ls[i] for i in order]
        ls = ls + [{"type": "stub-network-link", "id": rid(r), "data": "255.255.255.255", "metric": 0}]
        if rng.random() < 0.5:
            ls.append({"type": "stub-network-link", "id": f"192.168.{int(rng.integers(0, 3))}.0", "data": "255.255.255.0",
                       "metric": int(rng.integers(1, hi + 1))})                                   # shared stub prefix
        routers.append({"adv_rtr": rid(r), "lsa_id": rid(r), "bits": [], "links": ls, "maxage": bool(r != local and rng.random() < 0.04)})
    idx = rng.permutation(len(ifaces)).tolist()                        # arena slots unrelated to name order
    for j, i in enumerate(ifaces):
        i["index"] = int(idx[j])
    return {"proto": "ospfv2", "source": f"random area {seed}", "router_id": rid(local), "max_paths": int(rng.choice([1, 2, 16])),
            "has_vlinks": False, "rib": [],
            "areas": [{"area_id": "0.0.0.0", "routers": routers, "networks": net_lsas, "interfaces": ifaces}]}
