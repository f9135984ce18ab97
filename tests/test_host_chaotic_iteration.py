"""CPU model of the argument in DESIGN.md section 4 for the dense multi-pass launches of the lean sweep: row evaluations of
arbitrary (stale) snapshots, applied in arbitrary order — so that a word can get WORSE for a while —, followed by what the
engine does behind a dense stretch (one sweep over EVERY row that wakes the dependents of what it changes, then stamped
sweeps until one changes nothing), end in the oracle's result.  The row function is the fixed point the kernels iterate:
distance = min over in-links, hops from the FIRST tight link in row order (cost descending, source ascending), first-hop
mask = OR over the tight links of (the link's own slot when the parent is the root, else the parent's mask).
TEST INFRASTRUCTURE: plain Python on tiny graphs, no engine involved."""
import numpy as np
import pytest

from holo_amd import synth
from oracle import graph_oracle as go

INF = 0xFFFFFFFF


def rows_of(g):
    """in-rows in the engine's order: (cost descending, source ascending, position ascending); entries (source, cost, position)."""
    rp = g.row_ptr.astype(np.int64)
    rows = [[] for _ in range(g.n)]
    for u in range(g.n):
        for k in range(rp[u], rp[u + 1]):
            rows[int(g.col[k])].append((u, int(g.metric[k]), int(k - rp[u])))
    for r in rows:
        r.sort(key=lambda e: (-e[1], e[0], e[2]))
    outs = [[int(t) for t in g.col[rp[u]:rp[u + 1]]] for u in range(g.n)]
    return rows, outs


def evaluate(v, root, row, snap):
    """F(v) from a snapshot of the neighbours' words."""
    if v == root:
        return (0, 0, 0)
    best = INF
    for u, w, _ in row:
        d = snap[u][0]
        if d != INF:
            best = min(best, d + w)
    if best == INF:
        return (INF, 0, 0)
    hops, mask = None, 0
    for u, w, pos in row:
        d, h, m = snap[u]
        if d != INF and d + w == best:
            if hops is None:
                hops = h + 1
            mask |= (1 << pos) if u == root else m
    return (best, hops, mask)


def graph(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(8, 40))
    links = set()
    for v in range(1, n):
        links.add((int(rng.integers(0, v)), v))                # connected
    for _ in range(int(rng.integers(0, 2 * n))):
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        if a != b:
            links.add((min(a, b), max(a, b)))
    links = sorted(links)
    s = np.array([a for a, b in links] + [b for a, b in links]); d = np.array([b for a, b in links] + [a for a, b in links])
    m = rng.integers(1, 5, len(s))                               # tie-heavy, positive
    row_ptr, col, met = synth._csr_from_links(n, s, d, m)
    return synth.CsrGraph(row_ptr, col, met, np.zeros(n, np.uint8), synth.MAX_PATH_METRIC_WIDE), rng


@pytest.mark.parametrize("seed", range(40))
def test_any_interleaving_of_stale_row_evaluations_then_the_stamped_protocol_ends_in_the_oracles_result(seed):
    g, rng = graph(seed)
    rows, outs = rows_of(g)
    n = g.n
    root = int(rng.integers(0, n))
    state = [(INF, 0, 0)] * n
    state[root] = (0, 0, 0)
    history = [list(state)]
    # ---- the dense stretch: evaluations of stale snapshots, their stores delayed and reordered
    pending = []                                                 # (due step, vertex, word)
    steps = int(rng.integers(2 * n, 12 * n))
    for t in range(steps):
        v = int(rng.integers(0, n))
        snap = history[int(rng.integers(max(0, len(history) - 6), len(history)))]    # up to five versions old
        pending.append((t + int(rng.integers(0, 8)), v, evaluate(v, root, rows[v], snap)))
        for item in [p for p in pending if p[0] <= t]:
            pending.remove(item)
            state[item[1]] = item[2]                             # may overwrite a better word with a staler one
        history.append(list(state))
    for _, v, word in sorted(pending, key=lambda p: p[0]):
        state[v] = word
    truth = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, np.array([root], np.uint32), 0, go.MAP)
    for v in range(n):                                           # (a) no word is ever better than final
        assert state[v][0] >= int(truth.dist[0, v])
    # ---- behind the stretch: every row once, dependents of changes woken; then stamped sweeps
    due = set()
    for v in range(n):
        w = evaluate(v, root, rows[v], state)
        if w != state[v]:
            state[v] = w
            due.update(outs[v])
    sweeps = 0
    while due:
        nxt = set()
        for v in sorted(due):
            w = evaluate(v, root, rows[v], state)
            if w != state[v]:
                state[v] = w
                nxt.update(outs[v])
        due = nxt
        sweeps += 1
        assert sweeps < 10 * n
    for v in range(n):
        assert state[v][0] == int(truth.dist[0, v]), v
        if state[v][0] != INF:
            assert state[v][1] == int(truth.hops[0, v]) and state[v][2] == int(truth.mask[0, v, 0]), v
