"""Route derivation on device (hspf_routes_device) — bit-exact against
(1) the reference's recorded local RIBs end to end (SPT and prefix attachment both on the GPU) and
(2) a numpy restatement of compute_routes' per-prefix reduction for many roots on random graphs."""
import glob
import json
import os

import numpy as np
import pytest

from holo_amd import isis as H
from holo_amd import routes as RT
from holo_amd import synth
from oracle import graph_oracle as go
from oracle import isis_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ISIS = sorted(glob.glob(os.path.join(GOLD, "isis", "*.json"))) + sorted(glob.glob(os.path.join(GOLD, "isis_steps", "*.json")))


@pytest.mark.parametrize("path", ISIS, ids=[os.path.basename(p)[:-5] for p in ISIS])
def test_device_routes_reproduce_reference_local_rib(spf_ctx, path):
    vec = json.load(open(path))
    want = sorted(vec["rib"], key=lambda r: R._net_key(r["prefix"]))
    assert RT.compute_spf_device_routes(H.Instance.from_vector(vec), spf_ctx) == want


@pytest.mark.parametrize("seed", range(4))
def test_device_routes_many_roots_vs_numpy(spf_ctx, seed):
    """Oracle: plain restatement of holo-isis/src/spf.rs:891-918 on the oracle's SPT tables —
    first-smallest `distance + metric` in vertex order, OR of the masks of every entry attaining it."""
    import torch
    rng = np.random.default_rng(seed)
    g = synth.random_lsdb(150, 15, 3.0, 500 + seed, metric_hi=6)
    n = g.n
    roots = np.arange(15, 15 + 100, dtype=np.uint32)
    P = 400
    n_e = 900
    pfx = np.sort(rng.integers(0, P, n_e))
    vtx = rng.integers(0, n, n_e)
    order = np.lexsort((vtx, pfx))
    pfx, vtx = pfx[order], vtx[order].astype(np.uint32)
    met = rng.integers(0, 5, n_e).astype(np.uint32)
    ptr = np.zeros(P + 1, np.uint32)
    np.add.at(ptr, pfx + 1, 1)
    ptr = np.cumsum(ptr, dtype=np.uint64).astype(np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP)
    W = ref.mask.shape[2]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    dev = torch.device("cuda:0")
    Rn = len(roots)
    dist = torch.empty((Rn, n), dtype=torch.int32, device=dev); hops = torch.empty((Rn, n), dtype=torch.int16, device=dev)
    flags = torch.empty((Rn, n), dtype=torch.int16, device=dev); mask = torch.empty((Rn, n, W), dtype=torch.int64, device=dev)
    spf_ctx.run_device(G, roots, 0, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                       mask_ptr=mask.data_ptr(), mask_words=W)
    bm = torch.empty((Rn, P), dtype=torch.int32, device=dev); be = torch.empty((Rn, P), dtype=torch.int32, device=dev)
    nm = torch.empty((Rn, P, W), dtype=torch.int64, device=dev)
    spf_ctx.routes_device(n, Rn, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), ptr, vtx, met,
                          best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr())
    torch.cuda.synchronize()
    G.free()
    bm = bm.cpu().numpy().view(np.uint32); be = be.cpu().numpy().view(np.uint32); nm = nm.cpu().numpy().view(np.uint64)
    for r in range(Rn):
        for p in range(P):
            best, ent, acc = 0xFFFFFFFF, 0xFFFFFFFF, np.zeros(W, np.uint64)
            for e in range(ptr[p], ptr[p + 1]):
                v = vtx[e]
                if not ref.flags[r, v]:
                    continue
                m = int(ref.dist[r, v]) + int(met[e])
                if m < best:
                    best, ent, acc = m, e, ref.mask[r, v].copy()
                elif m == best:
                    acc |= ref.mask[r, v]
            assert bm[r, p] == best and be[r, p] == ent and np.array_equal(nm[r, p], acc), (r, p)


# ---- OSPFv2: update_rib_intra_area on device ------------------------------------------------------------------------
from holo_amd import engine as E             # noqa: E402
from holo_amd import ospf as HO              # noqa: E402
from oracle import ospf_ref as RO            # noqa: E402

OSPF = sorted(glob.glob(os.path.join(GOLD, "ospfv2", "*.json"))) + sorted(glob.glob(os.path.join(GOLD, "ospfv2_steps", "*.json")))


@pytest.mark.parametrize("path", OSPF, ids=[os.path.basename(p)[:-5] for p in OSPF])
def test_ospf_device_routes_reproduce_reference_intra_area_rib(spf_ctx, path):
    vec = json.load(open(path))
    areas = [HO.Area.from_vector(a) for a in vec["areas"]]
    got = RT.ospf_intra_area_device_routes(vec["router_id"], areas, vec["max_paths"], spf_ctx)
    assert got == RO.intra_area_rib(vec)                                   # literal restatement, every vector
    if not vec["has_vlinks"]:                                              # the reference's own recorded answer
        want = sorted([r for r in vec["rib"] if r["type"] == "intra-area"], key=lambda r: RO._net_key(r["prefix"]))
        assert got == want


OSPF_WIRE = [p for p in sorted(glob.glob(os.path.join(GOLD, "ospfv2_steps", "*.json"))) if "ibus_routes" in json.load(open(p))]


@pytest.mark.parametrize("path", OSPF_WIRE, ids=[os.path.basename(p)[:-5] for p in OSPF_WIRE])
def test_ospf_wire_step_from_device_routes_reproduces_recorded_ibus_messages(spf_ctx, path):
    """SURVEY.md 8f-4 for OSPFv2: SPT and both prefix reductions of every area on the device, the fold into the RIB and
    the wire step (holo_amd.ospf.update_global_rib) on the host: the RouteIpAdd / RouteIpDel sequence the reference
    recorded for the step (inter-area rows, where a step has any, come from the recording)."""
    vec = json.load(open(path))
    want = [{k: m[k] for k in m if k != "distance"} for m in vec["ibus_routes"]]
    areas = [HO.Area.from_vector(a) for a in vec["areas"]]
    rows = RT.ospf_intra_area_device_routes(vec["router_id"], areas, vec["max_paths"], spf_ctx)
    rows = rows + [r for r in vec["rib"] if r["type"] != "intra-area"]
    assert HO.update_global_rib(rows, vec["rib_before"], vec["ifindex"]) == want


@pytest.mark.parametrize("path", OSPF_WIRE, ids=[os.path.basename(p)[:-5] for p in OSPF_WIRE])
def test_ospf_hand_off_from_device_tables_reproduces_recorded_ibus_messages(spf_ctx, path):
    """SURVEY.md 8f-4 for OSPFv2, END TO END on the device like the IS-IS one: SPT, the ORDERED prefix fold (ONE table in
    Ospfv2::intra_area_networks order, so that the device result IS the RIB row), the comparison with the RIB the
    reference held before the step, the compaction and the packing; one record stream comes back and is expanded into
    the exact RouteIpAdd / RouteIpDel sequence the reference recorded (inter-area rows, where a step has any, come from
    the recording and go through the host rule).  The two two-area steps: both areas folded into one RIB ON THE DEVICE (round 5, hspf_rib_fold_device)."""
    vec = json.load(open(path))
    want = [{k: m[k] for k in m if k != "distance"} for m in vec["ibus_routes"]]
    areas = [HO.Area.from_vector(a) for a in vec["areas"]]
    other = [r for r in vec["rib"] if r["type"] != "intra-area"]
    msgs, n_rec, n_pfx = RT.ospf_update_global_rib_device(vec["router_id"], areas, vec["max_paths"], spf_ctx, vec["rib_before"],
                                                          vec["ifindex"], other)
    assert msgs == want
    if len(areas) == 1:
        assert n_pfx >= len([r for r in vec["rib"] if r["type"] == "intra-area"]) and n_rec <= n_pfx
        # the device fold alone: the rows of the ordered table equal the twin's intra-area RIB
        assert RT.ospf_intra_area_device_routes(vec["router_id"], areas, vec["max_paths"], spf_ctx) == \
            HO.compute_spf_intra_area(vec["router_id"], areas, vec["max_paths"], spf_ctx)


def test_ospf_hand_off_random_areas_before_and_after_a_change_equal_the_host_rule(spf_ctx):
    """Random OSPFv2 areas (tests/_random_ospf.py: parallel links, transit networks, one-way links, shared stub prefixes,
    MaxAge LSAs, max-paths 1 / 2 / 16): the RIB of the area as it is, then remote routers change link costs, lose a
    Router-LSA or a Network-LSA; the device hand-off (SPT, ordered fold, comparison with the old RIB, record stream)
    must give the message sequence of the host rule on the twin's two RIBs — which is the restatement pinned to the
    reference's recorded ibus output (tests/test_oracle_golden.py)."""
    import copy
    import random
    from _random_ospf import make
    checked = nonempty = 0
    for seed in range(2000, 2060):
        vec = make(seed)
        rng = random.Random(seed)
        a0 = vec["areas"][0]
        ifindex = {i["name"]: 10 + k for k, i in enumerate(sorted(a0["interfaces"], key=lambda i: i["name"]))}
        before = HO.compute_spf_intra_area(vec["router_id"], [HO.Area.from_vector(a0)], vec["max_paths"], spf_ctx)
        a1 = copy.deepcopy(a0)
        for r in a1["routers"]:
            if r["adv_rtr"] == vec["router_id"]:
                continue                                       # the root's own row keeps its first-hop slots comparable by construction
            what = rng.random()
            if what < 0.25:
                for l in r["links"]:
                    if rng.random() < 0.5:
                        l["metric"] = rng.randint(1, 12)
            elif what < 0.32:
                r["maxage"] = True
        for nl in a1["networks"]:
            if rng.random() < 0.15:
                nl["maxage"] = not nl["maxage"]
        area1 = HO.Area.from_vector(a1)
        after = HO.compute_spf_intra_area(vec["router_id"], [area1], vec["max_paths"], spf_ctx)
        want = HO.update_global_rib(after, before, ifindex)
        got, n_rec, n_pfx = RT.ospf_update_global_rib_device(vec["router_id"], [area1], vec["max_paths"], spf_ctx, before, ifindex)
        assert got == want, seed
        checked += 1
        nonempty += bool(want)
    assert checked == 60 and nonempty >= 15


def _multi_area_instance(make, first_seed, rng, n_areas):
    """A multi-area instance out of single-area random vectors that share the local router: areas 0.0.0.0, 0.0.0.1, ... (the
    random vectors reuse one subnet numbering, so the areas advertise overlapping prefixes: ties, take-overs and merges across
    areas); interface names made unique per area."""
    import copy
    base = make(first_seed)
    areas, seed = [], first_seed
    while len(areas) < n_areas:
        v = make(seed) if seed != first_seed else base
        seed += 1000
        if v["router_id"] != base["router_id"]:
            continue
        a = copy.deepcopy(v["areas"][0])
        a["area_id"] = f"0.0.0.{len(areas)}"
        for i in a["interfaces"]:
            i["name"] = f"a{len(areas)}-{i['name']}"
        areas.append(a)
    return base["router_id"], base["max_paths"], areas


def _mutate_remote_routers(area, router_id, rng):
    for r in area["routers"]:
        if r["adv_rtr"] == router_id:
            continue                                       # the root's own rows keep their first-hop slots comparable by construction
        what = rng.random()
        if what < 0.25:
            for l in r["links"]:
                if rng.random() < 0.5:
                    l["metric"] = rng.randint(1, 12)
        elif what < 0.32:
            r["maxage"] = True
    for nl in area["networks"]:
        if rng.random() < 0.15:
            nl["maxage"] = not nl["maxage"]


def test_ospf_multi_area_fold_on_device_random_instances_equal_the_host_rule(spf_ctx):
    """Round 5: SEVERAL areas folded into ONE RIB on the device (hspf_rib_fold_device:
    update_rib_intra_area of each area on what the earlier areas left, one instance-wide first-hop slot numbering), compared
    there with the RIB held before, one record stream back.  80 random 2- and 3-area OSPFv2 instances before / after remote
    routers change: the messages must be those of the host rule on the twin's two multi-area RIBs."""
    import copy
    import random
    from _random_ospf import make
    checked = nonempty = shared = 0
    for seed in range(5000, 5080):
        rng = random.Random(seed)
        router_id, max_paths, areas0 = _multi_area_instance(make, seed, rng, 2 + seed % 2)
        ifindex = {i["name"]: 10 + k for k, i in enumerate(sorted((i for a in areas0 for i in a["interfaces"]), key=lambda i: i["name"]))}
        before = HO.compute_spf_intra_area(router_id, [HO.Area.from_vector(a) for a in areas0], max_paths, spf_ctx)
        areas1 = copy.deepcopy(areas0)
        for a in areas1:
            _mutate_remote_routers(a, router_id, rng)
        live1 = [HO.Area.from_vector(a) for a in areas1]
        after = HO.compute_spf_intra_area(router_id, live1, max_paths, spf_ctx)
        want = HO.update_global_rib(after, before, ifindex)
        got, n_rec, n_pfx = RT.ospf_update_global_rib_device(router_id, live1, max_paths, spf_ctx, before, ifindex)
        assert got == want, seed
        # the fold alone: from an empty RIB every installable route is announced, with the merged next hops
        all_new, _, _ = RT.ospf_update_global_rib_device(router_id, live1, max_paths, spf_ctx, [], ifindex)
        assert all_new == HO.update_global_rib(after, [], ifindex), seed
        checked += 1
        nonempty += bool(want)
        pa = [{r["prefix"] for r in HO.compute_spf_intra_area(router_id, [x], max_paths, spf_ctx)} for x in live1]
        shared += bool(set.intersection(*pa)) if len(pa) > 1 else 0
    assert checked == 80 and nonempty >= 20 and shared >= 20, (checked, nonempty, shared)


def test_ospfv3_wire_step_and_multi_area_fold_on_device_equal_the_host_rule(spf_ctx):
    """The same for OSPFv3 (Intra-Area-Prefix-LSAs in LSDB order, per-entry origins, link-local next hops): one- and two-area
    random instances before / after remote routers change their Intra-Area-Prefix / Router-LSAs; messages = the host rule on
    the twin's RIBs.  (The reference holds no OSPFv3 STEP tests — the conformance module is commented out upstream — but it
    does hold the 44 topology recordings `output/ibus.jsonl`: those pin the OSPFv3 wire step in
    test_ospf_cold_start_from_device_tables_reproduces_recorded_ibus_state below.)"""
    import copy
    import random
    from _random_ospfv3 import make
    from holo_amd import ospfv3 as H3
    checked = nonempty = 0
    for seed in range(6000, 6060):
        rng = random.Random(seed)
        base = make(seed)
        af = base["af"]
        areas0, s2 = [], seed
        while len(areas0) < 1 + seed % 2:
            v = make(s2)
            s2 += 1000
            if v["router_id"] != base["router_id"] or v["af"] != af:
                continue
            a = copy.deepcopy(v["areas"][0])
            a["area_id"] = f"0.0.0.{len(areas0)}"
            for i in a["interfaces"]:
                i["name"] = f"a{len(areas0)}-{i['name']}"
            areas0.append(a)
        router_id, max_paths = base["router_id"], base["max_paths"]
        ifindex = {i["name"]: 10 + k for k, i in enumerate(sorted((i for a in areas0 for i in a["interfaces"]), key=lambda i: i["name"]))}
        before = H3.compute_spf_intra_area(router_id, [H3.Area3.from_vector(a) for a in areas0], max_paths, spf_ctx, af)
        areas1 = copy.deepcopy(areas0)
        for a in areas1:
            for l in a["iaps"]:
                if l["adv_rtr"] != router_id and rng.random() < 0.3:
                    for pfx in l["prefixes"]:
                        pfx["metric"] = rng.randint(0, 9)
            for r in a["routers"]:
                if r["adv_rtr"] != router_id and rng.random() < 0.25:
                    for k in r["links"]:
                        k["metric"] = rng.randint(1, 12)
        live1 = [H3.Area3.from_vector(a) for a in areas1]
        after = H3.compute_spf_intra_area(router_id, live1, max_paths, spf_ctx, af)
        want = HO.update_global_rib(after, before, ifindex)
        got, n_rec, n_pfx = RT.ospf_update_global_rib_device(router_id, live1, max_paths, spf_ctx, before, ifindex, version=3, af=af)
        assert got == want, seed
        all_new, _, _ = RT.ospf_update_global_rib_device(router_id, live1, max_paths, spf_ctx, [], ifindex, version=3, af=af)
        assert all_new == HO.update_global_rib(after, [], ifindex), seed
        checked += 1
        nonempty += bool(want)
    assert checked == 60 and nonempty >= 10, (checked, nonempty)


@pytest.mark.parametrize("seed", range(3))
@pytest.mark.parametrize("mode", [E.PFX_SATURATING, E.PFX_SATURATING | E.PFX_LAST_MIN, E.PFX_LAST_MIN])
def test_device_routes_ospf_rules_many_roots_vs_restatement(spf_ctx, seed, mode):
    """holo-ospf/src/route.rs:343-448 restated per prefix on the oracle's SPT tables: saturating add; with LAST_MIN an
    equal metric from a later entry replaces the route (transit-network rule, larger LS-ID wins), otherwise it merges.
    A few advertised metrics sit just below 2^32 so that the saturation is exercised."""
    import torch
    rng = np.random.default_rng(50 + seed)
    g = synth.random_lsdb(120, 12, 3.0, 900 + seed, metric_hi=4, max_path=0xFFFFFFFF)
    n = g.n
    roots = np.arange(12, 12 + 70, dtype=np.uint32)
    P, n_e = 300, 800
    pfx = np.sort(rng.integers(0, P, n_e))
    vtx = rng.integers(0, n, n_e)
    order = np.lexsort((vtx, pfx))
    pfx, vtx = pfx[order], vtx[order].astype(np.uint32)
    met = rng.integers(0, 4, n_e).astype(np.uint32)
    big = rng.random(n_e) < 0.05
    met[big] = (0xFFFFFFFF - rng.integers(0, 6, int(big.sum()))).astype(np.uint32)
    ptr = np.zeros(P + 1, np.uint32)
    np.add.at(ptr, pfx + 1, 1)
    ptr = np.cumsum(ptr, dtype=np.uint64).astype(np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, E.RUN_NET_NEXTHOPS, go.HEAP)
    W = ref.mask.shape[2]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    dev = torch.device("cuda:0")
    Rn = len(roots)
    dist = torch.empty((Rn, n), dtype=torch.int32, device=dev); hops = torch.empty((Rn, n), dtype=torch.int16, device=dev)
    flags = torch.empty((Rn, n), dtype=torch.int16, device=dev); mask = torch.empty((Rn, n, W), dtype=torch.int64, device=dev)
    spf_ctx.run_device(G, roots, E.RUN_NET_NEXTHOPS, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(),
                       flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
    bm = torch.empty((Rn, P), dtype=torch.int32, device=dev); be = torch.empty((Rn, P), dtype=torch.int32, device=dev)
    nm = torch.empty((Rn, P, W), dtype=torch.int64, device=dev)
    spf_ctx.routes_device(n, Rn, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), ptr, vtx, met,
                          best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr(),
                          flags=mode)
    torch.cuda.synchronize()
    G.free()
    bm = bm.cpu().numpy().view(np.uint32); be = be.cpu().numpy().view(np.uint32); nm = nm.cpu().numpy().view(np.uint64)
    sat, last = bool(mode & E.PFX_SATURATING), bool(mode & E.PFX_LAST_MIN)
    for r in range(Rn):
        for p in range(P):
            best, ent, acc = 0xFFFFFFFF, 0xFFFFFFFF, np.zeros(W, np.uint64)
            for e in range(ptr[p], ptr[p + 1]):
                v = vtx[e]
                if not ref.flags[r, v]:
                    continue
                m = int(ref.dist[r, v]) + int(met[e])
                m = min(m, 0xFFFFFFFF) if sat else m & 0xFFFFFFFF
                if ent == 0xFFFFFFFF or m < best or (last and m == best):
                    best, ent, acc = m, e, ref.mask[r, v].copy()
                elif m == best:
                    acc |= ref.mask[r, v]
            assert bm[r, p] == best and be[r, p] == ent and np.array_equal(nm[r, p], acc), (r, p)


# ---- OSPFv3: the ordered fold of update_rib_intra_area on device (HSPF_PFX_ORDERED) -----------------------------------
from holo_amd import ospfv3 as H3            # noqa: E402
from oracle import ospfv3_ref as R3          # noqa: E402
from _random_ospfv3 import make as make_v3   # noqa: E402

OSPF3 = sorted(glob.glob(os.path.join(GOLD, "ospfv3", "*.json")))


@pytest.mark.parametrize("path", OSPF3, ids=[os.path.basename(p)[:-5] for p in OSPF3])
def test_ospfv3_device_routes_reproduce_reference_intra_area_rib(spf_ctx, path):
    vec = json.load(open(path))
    areas = [H3.Area3.from_vector(a) for a in vec["areas"]]
    got = RT.ospfv3_intra_area_device_routes(vec["router_id"], areas, vec["max_paths"], spf_ctx, vec["af"])
    assert got == R3.intra_area_rib(vec)                                   # literal restatement, every vector
    if not vec["has_vlinks"]:                                              # the reference's own recorded answer
        want = sorted([r for r in vec["rib"] if r["type"] == "intra-area"], key=lambda r: R3._net_key(r["prefix"]))
        assert got == want


@pytest.mark.parametrize("block", range(4))
def test_ospfv3_device_routes_random_areas(spf_ctx, block):
    for seed in range(block * 20, block * 20 + 20):
        vec = make_v3(seed)
        areas = [H3.Area3.from_vector(a) for a in vec["areas"]]
        got = RT.ospfv3_intra_area_device_routes(vec["router_id"], areas, vec["max_paths"], spf_ctx, vec["af"])
        assert got == R3.intra_area_rib(vec), seed


@pytest.mark.parametrize("seed", range(3))
def test_device_routes_ordered_fold_many_roots_vs_restatement(spf_ctx, seed):
    """holo-ospf/src/route.rs:343-448 restated entry by entry on the oracle's SPT tables: entries in arbitrary (not
    vertex) order, network entries with the greater-origin rule, an initial route per prefix for a third of the
    prefixes, metrics that saturate."""
    import torch
    rng = np.random.default_rng(70 + seed)
    g = synth.random_lsdb(120, 12, 3.0, 950 + seed, metric_hi=4, max_path=0xFFFFFFFF)
    n = g.n
    roots = np.arange(12, 12 + 70, dtype=np.uint32)
    P, n_e = 300, 900
    pfx = np.sort(rng.integers(0, P, n_e))
    vtx = rng.integers(0, n, n_e).astype(np.uint32)
    isnet = rng.random(n_e) < 0.4
    met = rng.integers(0, 3, n_e).astype(np.uint32)
    big = rng.random(n_e) < 0.05
    met[big] = (0xFFFFFFFF - rng.integers(0, 6, int(big.sum()))).astype(np.uint32)
    org = rng.integers(0, 6, n_e).astype(np.uint32)
    ptr = np.zeros(P + 1, np.uint32)
    np.add.at(ptr, pfx + 1, 1)
    ptr = np.cumsum(ptr, dtype=np.uint64).astype(np.uint32)
    iex = (rng.random(P) < 0.33).astype(np.uint8)
    imet = rng.integers(0, 12, P).astype(np.uint32)
    iorg = rng.integers(0, 6, P).astype(np.uint32)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, E.RUN_NET_NEXTHOPS, go.HEAP)
    W = ref.mask.shape[2]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    dev = torch.device("cuda:0")
    Rn = len(roots)
    dist = torch.empty((Rn, n), dtype=torch.int32, device=dev); hops = torch.empty((Rn, n), dtype=torch.int16, device=dev)
    flags = torch.empty((Rn, n), dtype=torch.int16, device=dev); mask = torch.empty((Rn, n, W), dtype=torch.int64, device=dev)
    spf_ctx.run_device(G, roots, E.RUN_NET_NEXTHOPS, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(),
                       flags_ptr=flags.data_ptr(), mask_ptr=mask.data_ptr(), mask_words=W)
    bm = torch.empty((Rn, P), dtype=torch.int32, device=dev); be = torch.empty((Rn, P), dtype=torch.int32, device=dev)
    nm = torch.empty((Rn, P, W), dtype=torch.int64, device=dev)
    pv = vtx | np.where(isnet, E.PFX_ENTRY_NETWORK, 0).astype(np.uint32)
    spf_ctx.routes_device(n, Rn, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), ptr, pv, met,
                          best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr(),
                          flags=E.PFX_SATURATING | E.PFX_ORDERED, pfx_origin=org, init_exists=iex, init_metric=imet, init_origin=iorg)
    torch.cuda.synchronize()
    G.free()
    bm = bm.cpu().numpy().view(np.uint32); be = be.cpu().numpy().view(np.uint32); nm = nm.cpu().numpy().view(np.uint64)
    for r in range(Rn):
        for p in range(P):
            exists = bool(iex[p])
            best, bo, ent, acc = (int(imet[p]), int(iorg[p]), E.PFX_KEPT_INIT, np.zeros(W, np.uint64)) if exists else (0xFFFFFFFF, 0, 0xFFFFFFFF, np.zeros(W, np.uint64))
            for e in range(ptr[p], ptr[p + 1]):
                v = vtx[e]
                if not ref.flags[r, v]:
                    continue
                m = min(int(ref.dist[r, v]) + int(met[e]), 0xFFFFFFFF)
                if exists and m > best:
                    continue
                if isnet[e] and exists:
                    if m < best or (m == best and int(org[e]) > bo):
                        exists = False
                    else:
                        continue
                if not exists or m < best:
                    exists, best, bo, ent, acc = True, m, int(org[e]), e, ref.mask[r, v].copy()
                else:
                    acc = acc | ref.mask[r, v]
            assert bm[r, p] == (best if exists else 0xFFFFFFFF) and be[r, p] == ent and np.array_equal(nm[r, p], acc), (r, p)


# ---- the wire step after the path: RIB diff on device (hspf_routes_diff_device) ---------------------

@pytest.mark.parametrize("seed", range(3))
def test_routes_diff_device_vs_restatement(spf_ctx, seed):
    """update_global_rib's comparison (holo-isis/src/route.rs:254-312) per (root, prefix) on two result sets of
    hspf_routes_device: the SPTs of 70 roots before and after a handful of link costs changed (same prefix table, the
    roots' own rows untouched), against a per-pair restatement; the compacted index lists are ascending per root."""
    import torch
    rng = np.random.default_rng(90 + seed)
    g = synth.random_lsdb(150, 10, 3.0, 990 + seed, metric_hi=5)
    n = g.n
    roots = np.arange(10, 10 + 70, dtype=np.uint32)
    P, n_e = 400, 900
    pfx = np.sort(rng.integers(0, P, n_e))
    vtx = rng.integers(0, n, n_e)
    order = np.lexsort((vtx, pfx))
    pfx, vtx = pfx[order], vtx[order].astype(np.uint32)
    met = rng.integers(0, 4, n_e).astype(np.uint32)
    ptr = np.zeros(P + 1, np.uint32)
    np.add.at(ptr, pfx + 1, 1)
    ptr = np.cumsum(ptr, dtype=np.uint64).astype(np.uint32)
    m2 = g.metric.copy()
    rp = g.row_ptr.astype(np.int64)
    far = np.nonzero(np.repeat(np.arange(n), np.diff(rp)) >= 90)[0]           # rows of non-root vertices only
    pick = rng.choice(far, size=12, replace=False)
    m2[pick] = m2[pick] + rng.integers(1, 4, 12).astype(np.uint32)
    dev = torch.device("cuda:0")
    Rn = len(roots)
    sets, refs = [], []
    for metric in (g.metric, m2):
        G = spf_ctx.upload(g.row_ptr, g.col, metric, g.vflags, g.max_path_metric)
        W = G.mask_words(roots)
        dist = torch.empty((Rn, n), dtype=torch.int32, device=dev); hops = torch.empty((Rn, n), dtype=torch.int16, device=dev)
        flags = torch.empty((Rn, n), dtype=torch.int16, device=dev); mask = torch.empty((Rn, n, W), dtype=torch.int64, device=dev)
        spf_ctx.run_device(G, roots, 0, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                           mask_ptr=mask.data_ptr(), mask_words=W)
        bm = torch.empty((Rn, P), dtype=torch.int32, device=dev); be = torch.empty((Rn, P), dtype=torch.int32, device=dev)
        nm = torch.empty((Rn, P, W), dtype=torch.int64, device=dev)
        spf_ctx.routes_device(n, Rn, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), ptr, vtx, met,
                              best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr())
        G.free()
        sets.append((bm, be, nm))
        refs.append((bm.cpu().numpy().view(np.uint32), be.cpu().numpy().view(np.uint32), nm.cpu().numpy().view(np.uint64)))
    act = torch.empty((Rn, P), dtype=torch.uint8, device=dev)
    chg = torch.empty((Rn * P,), dtype=torch.int32, device=dev)
    cptr = torch.empty((Rn + 1,), dtype=torch.int32, device=dev)
    spf_ctx.routes_diff_device(Rn, P, W, tuple(t.data_ptr() for t in sets[0]), tuple(t.data_ptr() for t in sets[1]),
                               action_ptr=act.data_ptr(), changed_ptr=chg.data_ptr(), changed_ptr_ptr=cptr.data_ptr())
    torch.cuda.synchronize()
    act = act.cpu().numpy(); chg = chg.cpu().numpy().view(np.uint32); cptr = cptr.cpu().numpy().view(np.uint32)
    (om, oe, on), (nm_, ne, nn) = refs
    n_changed = 0
    for r in range(Rn):
        want_idx = []
        for p in range(P):
            had, has = oe[r, p] != 0xFFFFFFFF, ne[r, p] != 0xFFFFFFFF
            if has:
                same = had and om[r, p] == nm_[r, p] and np.array_equal(on[r, p], nn[r, p])
                a = E.DIFF_SAME if same else (E.DIFF_INSTALL if nn[r, p].any() else E.DIFF_SILENT)
            else:
                a = (E.DIFF_WITHDRAW if on[r, p].any() else E.DIFF_SILENT) if had else E.DIFF_SAME
            assert act[r, p] == a, (r, p)
            if a in (E.DIFF_INSTALL, E.DIFF_WITHDRAW):
                want_idx.append(p)
        assert chg[cptr[r]:cptr[r + 1]].tolist() == want_idx, r
        n_changed += len(want_idx)
    assert cptr[0] == 0 and cptr[Rn] == n_changed and 0 < n_changed < Rn * P


ISIS_WIRE = sorted(p for p in glob.glob(os.path.join(GOLD, "isis_steps", "*.json")) if "summary" not in os.path.basename(p))


@pytest.mark.parametrize("path", ISIS_WIRE, ids=[os.path.basename(p)[:-5] for p in ISIS_WIRE])
def test_hand_off_from_device_tables_reproduces_recorded_ibus_messages(spf_ctx, path):
    """SURVEY.md 8f-4 end to end: SPT, prefix attachment, the comparison with the RIB the reference held BEFORE the step
    and the compaction of what changed on the device, ONE packed record stream to the host (hspf_routes_pack), expanded
    into messages — equal to the RouteIpAdd / RouteIpDel sequence the reference recorded on the ibus for the step, in
    order; and the stream is the changed routes, not the RIB."""
    vec = json.load(open(path))
    inst = H.Instance.from_vector(vec)
    want = [{k: m[k] for k in m if k != "distance"} for m in vec["ibus_routes"]]
    got, n_rec, n_pfx = RT.update_global_rib_device(inst, spf_ctx, vec["rib_before"], vec["ifindex"])
    assert got == want
    assert n_rec <= n_pfx and (n_pfx == 0 or n_rec <= len(want) + 4)


def test_resident_prefix_table_skips_the_upload_and_gives_the_same_routes(spf_ctx):
    """HSPF_PFX_RESIDENT: the second call with the caller's unchanged arrays skips checks and copies; results equal the
    first call's; a table in OTHER arrays with the flag set is uploaded as usual (nothing recorded matches), and so is
    the old table again afterwards."""
    import torch
    rng = np.random.default_rng(3)
    g = synth.random_lsdb(120, 8, 3.0, 4711, metric_hi=5)
    n = g.n
    roots = np.arange(8, 8 + 20, dtype=np.uint32)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    dev = torch.device("cuda:0")
    W, Rn = G.mask_words(roots), len(roots)
    dist = torch.empty((Rn, n), dtype=torch.int32, device=dev); hops = torch.empty((Rn, n), dtype=torch.int16, device=dev)
    flags = torch.empty((Rn, n), dtype=torch.int16, device=dev); mask = torch.empty((Rn, n, W), dtype=torch.int64, device=dev)
    spf_ctx.run_device(G, roots, 0, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=flags.data_ptr(),
                       mask_ptr=mask.data_ptr(), mask_words=W)
    G.free()

    def table(P, ne):
        pfx = np.sort(rng.integers(0, P, ne)); vtx = rng.integers(0, n, ne)
        order = np.lexsort((vtx, pfx))
        ptr = np.zeros(P + 1, np.uint32); np.add.at(ptr, pfx + 1, 1)
        return np.cumsum(ptr, dtype=np.uint64).astype(np.uint32), vtx[order].astype(np.uint32), rng.integers(0, 4, ne).astype(np.uint32)

    def routes(t, fl):
        P = len(t[0]) - 1
        bm = torch.empty((Rn, P), dtype=torch.int32, device=dev); be = torch.empty((Rn, P), dtype=torch.int32, device=dev)
        nm = torch.empty((Rn, P, W), dtype=torch.int64, device=dev)
        spf_ctx.routes_device(n, Rn, W, dist.data_ptr(), flags.data_ptr(), mask.data_ptr(), t[0], t[1], t[2],
                              best_metric_ptr=bm.data_ptr(), best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr(), flags=fl)
        return bm.cpu().numpy().copy(), be.cpu().numpy().copy(), nm.cpu().numpy().copy()
    t1, t2 = table(300, 700), table(300, 700)
    a = routes(t1, 0)
    b = routes(t1, E.PFX_RESIDENT)
    c = routes(t2, E.PFX_RESIDENT)                           # other arrays: not what was recorded -> full path
    d = routes(t2, 0)
    e = routes(t1, E.PFX_RESIDENT)                           # t2 is resident now: t1 is uploaded again
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for x, y in zip(c, d):
        assert np.array_equal(x, y)
    for x, y in zip(a, e):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[0], c[0])


# ---- the reference's recorded COLD-START wire output, every topology router ----------------------------
# tests/golden/wire/ (tools/make_golden_wire.py): the final per-prefix state of `output/ibus.jsonl` — IS-IS 38, OSPFv2 50,
# OSPFv3 44 (fe80:: link-local next hops).  From DEVICE tables: SPT, prefix attachment / ordered fold, the comparison with an
# empty RIB, compaction and ONE packed record stream (hspf_routes_diff_device, hspf_routes_pack), expanded on the host.
import _wire as W        # noqa: E402

_WI = W.wire_paths("isis")
_WO = W.wire_paths("ospfv2") + W.wire_paths("ospfv3")


def test_cold_start_wire_vectors_present():
    assert (len(_WI), len(W.wire_paths("ospfv2")), len(W.wire_paths("ospfv3"))) == (38, 50, 44)


@pytest.mark.parametrize("path", _WI, ids=[os.path.basename(p)[:-5] for p in _WI])
def test_isis_cold_start_from_device_tables_reproduces_recorded_ibus_state(spf_ctx, path):
    W.check_isis_cold_start(path, spf_ctx, device=True)


@pytest.mark.parametrize("path", _WO, ids=[("v3-" if "ospfv3" in p else "v2-") + os.path.basename(p)[:-5] for p in _WO])
def test_ospf_cold_start_from_device_tables_reproduces_recorded_ibus_state(spf_ctx, path):
    W.check_ospf_cold_start(path, spf_ctx, device=True)
