"""An `engine` stand-in for CPU-only tests of the HOST logic (holo_amd/isis.py, holo_amd/ospf.py):
same upload()/run()/slot_table() surface as holo_amd.engine.SpfContext, answered by the CPU
oracle.  Lives under tests/ on purpose: the product never sees it (no CPU fallback there)."""
from __future__ import annotations

import numpy as np

from holo_amd import engine as E
from oracle import graph_oracle as go


class OracleGraph:
    def __init__(self, row_ptr, col, metric, vflags, max_path_metric):
        self.row_ptr = np.ascontiguousarray(row_ptr, np.uint32)
        self.col = np.ascontiguousarray(col, np.uint32)
        self.metric = np.ascontiguousarray(metric, np.uint32)
        self.vflags = np.ascontiguousarray(vflags, np.uint8)
        self.max_path_metric = max_path_metric
        self.n = len(self.row_ptr) - 1

    def _twoway(self, u, k):
        t = int(self.col[k])
        return bool((self.col[self.row_ptr[t]:self.row_ptr[t + 1]] == u).any())

    def slot_table(self, root: int):
        """Independent restatement of the slot numbering of include/holo_spf_hip.h."""
        hv, hb = [root], [0]
        total = int(self.row_ptr[root + 1] - self.row_ptr[root])
        seen = {root}
        qi = 0
        while qi < len(hv):
            p = hv[qi]; qi += 1
            for k in range(int(self.row_ptr[p]), int(self.row_ptr[p + 1])):
                t = int(self.col[k])
                if t in seen or not (self.vflags[t] & 1) or not self._twoway(p, k):
                    continue
                seen.add(t)
                hv.append(t); hb.append(total)
                total += int(self.row_ptr[t + 1] - self.row_ptr[t])
        return np.asarray(hv, np.uint32), np.asarray(hb, np.uint32), total

    def patch(self, vertices, rows, vflags):
        """Row replacement with the semantics of hspf_graph_patch (holo_amd.engine.SpfGraph.patch)."""
        order = np.argsort(np.asarray(vertices, dtype=np.int64), kind="stable")
        vs = np.asarray(vertices, np.uint32)[order]
        assert len(set(vs.tolist())) == len(vs) and (vs < self.n).all()
        cols = [np.asarray(rows[i][0], np.uint32) for i in order]
        mets = [np.asarray(rows[i][1], np.uint32) for i in order]
        self.row_ptr, self.col, self.metric, self.vflags = E.splice_rows(
            self.row_ptr, self.col, self.metric, self.vflags, vs, cols, mets, np.asarray(vflags, np.uint8)[order])

    def free(self):
        pass


class OracleEngine:
    variant = go.MAP

    def upload(self, row_ptr, col, metric, vflags, max_path_metric):
        return OracleGraph(row_ptr, col, metric, vflags, max_path_metric)

    def run(self, G: OracleGraph, roots, run_flags: int = 0, **_kw):
        roots = np.ascontiguousarray(roots, np.uint32)
        oflags = run_flags & (E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD)
        r = go.run(G.row_ptr, G.col, G.metric, G.vflags, G.max_path_metric, roots, oflags, self.variant)
        rank = r.pop_rank if (run_flags & E.RUN_POP_RANK) else None
        # like the real engine, tell the caller which roots did NOT pop in the static (distance,
        # index) order (zero-cost plateaus): RF_EXACT makes the host layer ask for pop ranks
        flags = r.flags.copy()
        for j in range(len(roots)):
            members = np.nonzero(r.flags[j])[0]
            static = members[np.lexsort((members, r.dist[j][members]))]
            popped = members[np.argsort(r.pop_rank[j][members], kind="stable")]
            if not np.array_equal(static, popped):
                flags[j][members] |= E.RF_EXACT
        return E.SpfResult(r.dist, r.hops, flags, r.mask, rank, {})
