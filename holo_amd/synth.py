"""Seeded synthetic LSDB graphs, produced directly in the CSR form of
include/holo_spf_hip.h.

All randomness comes from a counter-mode splitmix64 stream (vectorisable, reproducible on any
box, no dependency on numpy's generator versions).  Vertex index == rank in the reference's
VertexId order (networks / pseudonodes first, then routers ascending), so integer compare is the
candidate-list tie-break (holo-isis/src/spf.rs:96-100, holo-ospf/src/ospfv2/spf.rs:41-45).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

SEED = 0x9E3779B97F4A7C15
MAX_PATH_METRIC_STANDARD = 1023          # holo-isis/src/spf.rs:45
MAX_PATH_METRIC_WIDE = 0xFE000000        # holo-isis/src/spf.rs:47
MAX_PATH_METRIC_OSPF = 0xFFFFFFFF        # saturating add, holo-ospf/src/spf.rs:672

VF_NETWORK, VF_NO_TRANSIT, VF_NO_EXPAND = 1, 2, 4


@dataclass
class CsrGraph:
    row_ptr: np.ndarray            # u32 [N+1]
    col: np.ndarray                # u32 [E]
    metric: np.ndarray             # u32 [E]
    vflags: np.ndarray             # u8  [N]
    max_path_metric: int = MAX_PATH_METRIC_WIDE
    name: str = ""
    meta: dict = field(default_factory=dict)

    @property
    def n(self) -> int:
        return len(self.row_ptr) - 1

    @property
    def e(self) -> int:
        return len(self.col)


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n 64-bit values of splitmix64 in counter mode (value i = mix(seed + stream*2^40 + i + 1))."""
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
             + (np.uint64(stream) << np.uint64(40))) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _csr_from_links(n: int, src: np.ndarray, dst: np.ndarray, met: np.ndarray):
    """Directed entries (src->dst, met) -> CSR; row order = order of appearance (stable)."""
    order = np.argsort(src, kind="stable")
    src, dst, met = src[order], dst[order], met[order]
    row_ptr = np.zeros(n + 1, np.uint32)
    np.add.at(row_ptr, src.astype(np.int64) + 1, 1)
    row_ptr = np.cumsum(row_ptr, dtype=np.uint64).astype(np.uint32)
    return row_ptr, dst.astype(np.uint32), met.astype(np.uint32)


def _grid8_links(rows: int, cols: int) -> np.ndarray:
    """Undirected links of an 8-neighbour rows x cols grid, vertex id = r*cols + c."""
    idx = np.arange(rows * cols, dtype=np.int64).reshape(rows, cols)
    parts = [
        (idx[:, :-1], idx[:, 1:]),        # E
        (idx[:-1, :], idx[1:, :]),        # S
        (idx[:-1, :-1], idx[1:, 1:]),     # SE
        (idx[:-1, 1:], idx[1:, :-1]),     # SW
    ]
    a = np.concatenate([p[0].ravel() for p in parts])
    b = np.concatenate([p[1].ravel() for p in parts])
    return np.stack([a, b], axis=1)


def _grid4_links(rows: int, cols: int) -> np.ndarray:
    idx = np.arange(rows * cols, dtype=np.int64).reshape(rows, cols)
    a = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    b = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    return np.stack([a, b], axis=1)


def _add_chords(n: int, links: np.ndarray, n_links_total: int, seed: int) -> np.ndarray:
    """Pad `links` with distinct random chords (no self loops, no duplicates) up to n_links_total."""
    need = n_links_total - len(links)
    if need <= 0:
        return links[:n_links_total]
    have = set((links[:, 0] * n + links[:, 1]).tolist())
    out = []
    stream = 1
    while need > 0:
        r = splitmix64(seed, 2 * need + 64, stream)
        stream += 1
        a = (r[0::2] % np.uint64(n)).astype(np.int64)
        b = (r[1::2] % np.uint64(n)).astype(np.int64)
        for x, y in zip(a.tolist(), b.tolist()):
            if x == y:
                continue
            lo, hi = (x, y) if x < y else (y, x)
            key = lo * n + hi
            if key in have:
                continue
            have.add(key)
            out.append((lo, hi))
            need -= 1
            if need == 0:
                break
    return np.concatenate([links, np.array(out, dtype=np.int64).reshape(-1, 2)])


def _routers_only(n: int, links: np.ndarray, seed: int, lo: int, hi: int, max_path: int,
                  name: str, meta: dict) -> CsrGraph:
    m = len(links)
    r = splitmix64(seed, 2 * m, 7)
    span = np.uint64(hi - lo + 1)
    w_fwd = (r[:m] % span).astype(np.int64) + lo
    w_rev = (r[m:] % span).astype(np.int64) + lo
    src = np.concatenate([links[:, 0], links[:, 1]])
    dst = np.concatenate([links[:, 1], links[:, 0]])
    met = np.concatenate([w_fwd, w_rev])
    row_ptr, col, metric = _csr_from_links(n, src, dst, met)
    return CsrGraph(row_ptr, col, metric, np.zeros(n, np.uint8), max_path, name,
                    dict(meta, n_links=m, n_entries=2 * m))


# ---- BASELINE.json configs -------------------------------------------------------------------

def ospf_500(seed: int = SEED) -> CsrGraph:
    """configs[0]: OSPFv2 single area, 500 routers, 20x25 4-neighbour grid of p2p links
    (955 links -> 1910 directed entries), metrics U[1,100]; root = router 0."""
    links = _grid4_links(20, 25)
    return _routers_only(500, links, seed, 1, 100, MAX_PATH_METRIC_OSPF, "ospf-500",
                         {"roots": [0], "proto": "ospfv2"})


def ospf_10k(seed: int = SEED) -> CsrGraph:
    """configs[1]: OSPFv2 single area, 10 000 routers, exactly 40 000 p2p links
    (100x100 8-neighbour grid = 39 402 links + 598 random chords) -> 80 000 entries; 1 root."""
    links = _add_chords(10000, _grid8_links(100, 100), 40000, seed)
    return _routers_only(10000, links, seed, 1, 100, MAX_PATH_METRIC_OSPF, "ospf-10k",
                         {"roots": [0], "proto": "ospfv2"})


def isis_100k(seed: int = SEED) -> CsrGraph:
    """configs[2] (HEADLINE): IS-IS L2 wide metrics, N = 100 000 routers (no pseudonodes),
    250x400 8-neighbour grid (398 052 links) + 101 948 random chords = exactly 500 000 links =
    1 000 000 directed IS-reachability entries, per-direction metrics U[1,100].
    64 roots = vertices floor(i*N/64)."""
    n = 100000
    links = _add_chords(n, _grid8_links(250, 400), 500000, seed)
    g = _routers_only(n, links, seed, 1, 100, MAX_PATH_METRIC_WIDE, "isis-100k",
                      {"proto": "isis-l2"})
    g.meta["roots"] = [(i * n) // 64 for i in range(64)]
    return g


def isis_fattree(k: int = 100, seed: int = SEED) -> CsrGraph:
    """configs[4]: k-ary three-tier fat-tree with single-homed hosts as IS nodes, unit metrics
    (maximal ECMP).  k=100: 2500 core + 5000 agg + 5000 edge + 250 000 hosts = 262 500 vertices,
    750 000 links -> 1 500 000 entries.  Vertex order: core, agg, edge, hosts.
    roots = one edge switch + its 100 neighbours (101 roots)."""
    h = k // 2
    n_core, n_agg, n_edge = h * h, k * h, k * h
    n_host = n_edge * h
    core0, agg0, edge0, host0 = 0, n_core, n_core + n_agg, n_core + n_agg + n_edge
    n = host0 + n_host
    pods = np.arange(k)
    a, b = [], []
    # core (i,j) i,j<h  <->  agg i of every pod
    ci, cj = np.meshgrid(np.arange(h), np.arange(h), indexing="ij")
    for p in pods.tolist():
        a.append((core0 + ci * h + cj).ravel())
        b.append((agg0 + p * h + ci).ravel())
    # agg <-> edge inside a pod (full bipartite)
    ai, ei = np.meshgrid(np.arange(h), np.arange(h), indexing="ij")
    for p in pods.tolist():
        a.append((agg0 + p * h + ai).ravel())
        b.append((edge0 + p * h + ei).ravel())
    # edge <-> hosts
    e_idx = np.repeat(np.arange(n_edge), h)
    a.append(edge0 + e_idx)
    b.append(host0 + np.arange(n_host))
    links = np.stack([np.concatenate(a), np.concatenate(b)], axis=1).astype(np.int64)
    g = _routers_only(n, links, seed, 1, 1, MAX_PATH_METRIC_WIDE, f"isis-fattree-k{k}",
                      {"proto": "isis-l2"})
    me = edge0
    nbrs = g.col[g.row_ptr[me]:g.row_ptr[me + 1]].tolist()
    g.meta["roots"] = [me] + nbrs
    return g


def ospf_multi_area(n_areas: int = 10, per_area: int = 5000, seed: int = SEED) -> list:
    """configs[3]: multi-area OSPF, `n_areas` areas of `per_area` routers, each a 50x100 8-neighbour
    grid (19 502 links) padded with random chords to exactly 20 000 links -> 40 000 entries, metrics
    U[1,100]; every area is its own graph (run_area is per area), roots = every 5th router of the
    area (1 000 per area, ~10 k in total).  The 50-router backbone of the survey's description only
    joins the areas at the route level and carries no per-prefix SPT work; it is left out."""
    out = []
    for a in range(n_areas):
        links = _add_chords(per_area, _grid8_links(50, per_area // 50), 20000, seed + 1000 * (a + 1))
        g = _routers_only(per_area, links, seed + 77 * (a + 1), 1, 100, MAX_PATH_METRIC_OSPF, f"ospf-ma-area{a}",
                          {"proto": "ospf", "area": a})
        g.meta["roots"] = list(range(0, per_area, 5))
        out.append(g)
    return out


# ---- adversarial random graphs for parity tests -------------------------------------------------

def random_lsdb(n_routers: int, n_networks: int, avg_deg: float, seed: int, *,
                metric_lo: int = 1, metric_hi: int = 20, max_path: int = MAX_PATH_METRIC_WIDE,
                p_oneway: float = 0.03, p_parallel: float = 0.05, p_overload: float = 0.03,
                p_noexpand: float = 0.02, zero_cost_router_links: bool = False,
                lan_size: int = 5, hopcount: bool = False) -> CsrGraph:
    """A small adversarial LSDB: routers + LAN pseudonodes/network vertices, parallel links,
    one-way links (fail the two-way check), overloaded and non-expandable vertices, tie-heavy
    metrics.  Networks take indices [0, n_networks) (they sort first in the reference),
    routers [n_networks, n).  network->router entries cost 0 (holo-ospf/src/ospfv2/spf.rs:412).
    With hopcount=True costs follow MetricMode::HopCount (holo-isis/src/spf.rs:1138-1145)."""
    n = n_routers + n_networks
    rng = np.random.default_rng(seed)      # test-only graphs: numpy's generator is fine here
    src, dst, met = [], [], []
    R0 = n_networks

    def add(u, v, w):
        src.append(u); dst.append(v); met.append(w)

    # router-router p2p links
    n_p2p = int(n_routers * avg_deg / 2)
    for _ in range(n_p2p):
        u, v = rng.integers(0, n_routers, 2)
        if u == v:
            continue
        lo = 0 if zero_cost_router_links else metric_lo
        w1, w2 = rng.integers(lo, metric_hi + 1, 2)
        reps = 2 if rng.random() < p_parallel else 1
        for _r in range(reps):
            add(R0 + u, R0 + v, int(w1))
            if rng.random() >= p_oneway:
                add(R0 + v, R0 + u, int(w2))
            if rng.random() < 0.5:
                w1 = int(rng.integers(metric_lo, metric_hi + 1))
    # LANs
    for net in range(n_networks):
        members = rng.choice(n_routers, size=min(lan_size, n_routers), replace=False)
        for m in members.tolist():
            add(R0 + m, net, int(rng.integers(metric_lo, metric_hi + 1)))
            if rng.random() >= p_oneway:
                add(net, R0 + m, 0)
    src = np.array(src, np.int64); dst = np.array(dst, np.int64); met = np.array(met, np.int64)
    if hopcount:
        met = np.where(dst < n_networks, 0, 1)
    # shuffle link order inside rows (LSP link order is arbitrary)
    perm = rng.permutation(len(src))
    row_ptr, col, metric = _csr_from_links(n, src[perm], dst[perm], met[perm])
    vflags = np.zeros(n, np.uint8)
    vflags[:n_networks] |= VF_NETWORK
    r = rng.random(n)
    vflags[(r < p_overload) & (np.arange(n) >= n_networks)] |= VF_NO_TRANSIT
    r = rng.random(n)
    vflags[r < p_noexpand] |= VF_NO_EXPAND
    return CsrGraph(row_ptr, col, metric, vflags, max_path, f"random-{seed}",
                    {"n_networks": n_networks})
