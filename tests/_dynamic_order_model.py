"""CPU model of holo_amd/csrc/spf_repair.hip.h (k_repair), statement by statement in plain Python: the pop order of a root
with zero-cost links in closed form — R (the largest index on the best release path), the groups' walks, the keys
(dist, R, pos) — and hops / first-hop masks evaluated in that order.  Test infrastructure: compared with the oracle's
literal loop (pop_rank, hops, masks) in tests/test_host_dynamic_order.py; the GPU kernel is compared with the same oracle."""
import numpy as np

INF = 0xFFFFFFFF
VF_NETWORK, VF_NO_TRANSIT, VF_NO_EXPAND = 1, 2, 4


def kept_links(row_ptr, col, metric, vflags):
    """Links that survive the two-way check and whose source can be expanded (the engine's kept links), with their
    position inside the source row: in-rows and out-rows."""
    n = len(row_ptr) - 1
    ins = [[] for _ in range(n)]
    outs = [[] for _ in range(n)]
    for u in range(n):
        if vflags[u] & VF_NO_EXPAND:
            continue
        for k in range(row_ptr[u], row_ptr[u + 1]):
            v = int(col[k])
            if not (col[row_ptr[v]:row_ptr[v + 1]] == u).any():
                continue
            ins[v].append((u, int(metric[k]), k - int(row_ptr[u])))
            outs[u].append((v, int(metric[k])))
    return ins, outs


def slot_bases(row_ptr, col, vflags, root):
    """The slot table of include/holo_spf_hip.h: H = [root] ++ networks reached through networks, two-way links only."""
    hv, hb = [root], {root: 0}
    total = int(row_ptr[root + 1] - row_ptr[root])
    qi = 0
    while qi < len(hv):
        p = hv[qi]; qi += 1
        for k in range(int(row_ptr[p]), int(row_ptr[p + 1])):
            t = int(col[k])
            if t in hb or not (vflags[t] & VF_NETWORK) or not (col[row_ptr[t]:row_ptr[t + 1]] == p).any():
                continue
            hb[t] = total
            hv.append(t)
            total += int(row_ptr[t + 1] - row_ptr[t])
    return hb


def dynamic_order(row_ptr, col, metric, vflags, root, dist, ignore_ovl=False, net_nexthops=False, words=1, heap_cap=None):
    """dist: final distances of the root (u32, INF = not in the SPT).  Returns (R, pos, hops, mask[n][words], rank) or None
    when a group outgrows heap_cap (the kernel then hands the root to k_exact)."""
    n = len(row_ptr) - 1
    ins, outs = kept_links(row_ptr, col, metric, vflags)
    D = [int(x) for x in dist]

    def src_ok(u):
        return ignore_ovl or u == root or (vflags[u] & VF_NETWORK) or not (vflags[u] & VF_NO_TRANSIT)

    def tight(u, w, v):
        return D[u] != INF and D[u] + w == D[v]

    zflag = [any(w == 0 for _, w, _ in ins[v]) for v in range(n)]
    zl = [v for v in range(n) if zflag[v]]
    R = list(range(n))
    pos = [0] * n
    UN = INF
    for v in zl:                                             # 1. seeds
        seed = D[v] == INF or v == root or any(w != 0 and src_ok(u) and tight(u, w, v) for u, w, _ in ins[v])
        R[v] = v if seed else UN
    while True:                                              # 2. min-max relaxation
        ch = False
        for v in zl:
            if R[v] == v:
                continue
            cand = R[v]
            for u, w, _ in ins[v]:
                if w != 0 or not src_ok(u) or D[u] != D[v] or u == v:
                    continue
                if R[u] != UN:
                    cand = min(cand, max(R[u], v))
            if cand < R[v]:
                R[v] = cand; ch = True
        if not ch:
            break
    assert all(R[v] != UN for v in zl if D[v] != INF)
    for v in zl:                                             # 3. the groups' walks
        y = R[v]
        if y == v or D[v] == INF or not src_ok(y):
            continue
        direct = [x for x, w in outs[y] if w == 0 and x != y and D[x] == D[v] and zflag[x] and R[x] == y]
        if v not in direct or min(direct) != v:
            continue
        heap, p = [], 0
        for x in direct:
            if pos[x] == 0:
                pos[x] = -1; heap.append(x)
        while heap:
            if heap_cap is not None and len(heap) > heap_cap:
                return None
            heap.sort()
            m = heap.pop(0)
            p += 1
            pos[m] = p
            if not src_ok(m):
                continue
            for x, w in outs[m]:
                if w == 0 and x != y and x != m and D[x] == D[v] and zflag[x] and R[x] == y and pos[x] == 0:
                    pos[x] = -1; heap.append(x)
    assert all(pos[v] > 0 for v in zl if D[v] != INF and R[v] != v)
    key = lambda v: (D[v], R[v], pos[v])                     # noqa: E731
    order = sorted((v for v in range(n) if D[v] != INF), key=key)
    rank = np.full(n, INF, np.uint32)
    for i, v in enumerate(order):
        rank[v] = i
    # hops / masks in that order (a vertex is evaluated after everything that precedes it)
    hb = slot_bases(row_ptr, col, vflags, root)
    hops = np.zeros(n, np.uint16)
    mask = np.zeros((n, words), np.uint64)
    for v in order:
        if v == root:
            continue
        best = None
        router = not (vflags[v] & VF_NETWORK)
        for u, w, fpos in ins[v]:
            if not src_ok(u) or not tight(u, w, v):
                continue
            if w == 0 and not (R[u], pos[u]) < (R[v], pos[v]):
                continue
            k = (D[u], R[u], pos[u])
            if best is None or k < best[0]:
                best = (k, int(hops[u]))
            if hops[u] == 0:
                if (router or net_nexthops) and u in hb:
                    s = hb[u] + fpos
                    if s // 64 < words:
                        mask[v, s // 64] |= np.uint64(1) << np.uint64(s % 64)
            else:
                mask[v] |= mask[u]
        assert best is not None, v
        hops[v] = min(best[1] + (1 if router else 0), 0xFFFF)
    return R, pos, hops, mask, rank
