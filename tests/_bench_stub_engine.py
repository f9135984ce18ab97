"""Stand-in for holo_amd.engine behind bench.py's rank code (HSPF_BENCH_STUB; tests/test_bench_ranks_gloo.py).

TEST INFRASTRUCTURE.  The interface bench.py's main() uses of MultiEngine — upload, mask_words, run / run_async / run_wait
with tables addressed by raw pointers, the in-place gather of the distance table, allgather_rows, stats — with the CPU
oracle computing the rows and torch.distributed (gloo) moving them, so that the launcher, the id exchange, the slicing
(the REAL hspf_shard_bounds of the C ABI), the in-flight loop, the verification against the gathered table and the JSON
assembly of bench.py run at world size 2 without a GPU.  Nothing here measures anything.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from holo_amd import synth
from holo_amd.engine import GATHER_ASYNC, GATHER_DIST, RUN_COUNT_ROWS, shard_bounds   # noqa: F401  (the C ABI's own arithmetic)
from oracle import graph_oracle as go


def bench_graph():
    g = synth.random_lsdb(700, 0, 3.0, 4242, metric_hi=30, p_oneway=0.0, p_overload=0.0, p_noexpand=0.0)
    return g


def multi_unique_id() -> bytes:
    return bytes(range(128))


class SpfContext:
    @staticmethod
    def async_lanes_of(_handle) -> int:
        return 0


def _view(ptr: int, count: int, ctype, dtype):
    return np.ctypeslib.as_array((ctype * count).from_address(ptr)).view(dtype)


class MultiEngine:
    def __init__(self, devices, world=None, first_rank=0, unique_id=None):
        self.world = world or len(devices)
        self.rank = first_rank
        self.sharded = unique_id is not None and self.world > 1
        if self.sharded:
            assert unique_id == multi_unique_id(), "the communicator id did not survive the broadcast"
        self.tickets, self.next = {}, 1
        self.last = {}

    def ctx_handle(self, i):
        return None

    def upload(self, row_ptr, col, metric, vflags, max_path_metric):
        return dict(row_ptr=row_ptr, col=col, metric=metric, vflags=vflags, mpm=max_path_metric, n=len(row_ptr) - 1)

    def free_graph(self, g):
        pass

    def mask_words(self, g, roots) -> int:
        return int(go.mask_words(g["row_ptr"], g["col"], g["metric"], g["vflags"], np.asarray(roots, np.uint32)))

    def _compute(self, g, roots, results):
        roots = np.asarray(roots, np.uint32)
        lo, hi = shard_bounds(len(roots), self.world, self.rank) if self.sharded else (0, len(roots))
        res, n, W = results[0], g["n"], results[0]["mask_words"]
        ref = go.run(g["row_ptr"], g["col"], g["metric"], g["vflags"], g["mpm"], roots[lo:hi], 0, go.HEAP, mask_words_=W)
        k = hi - lo
        _view(res["dist"] + lo * n * 4, k * n, ctypes.c_uint32, np.uint32)[:] = ref.dist.reshape(-1)
        _view(res["hops"] + lo * n * 2, k * n, ctypes.c_uint16, np.uint16)[:] = ref.hops.reshape(-1)
        _view(res["flags"] + lo * n * 2, k * n, ctypes.c_uint16, np.uint16)[:] = ref.flags.reshape(-1).astype(np.uint16)
        _view(res["mask"] + lo * n * 8 * W, k * n * W, ctypes.c_uint64, np.uint64)[:] = ref.mask.reshape(-1)
        self.last = {"n_roots": k, "rows": len(roots), "n": n}

    def run_async(self, g, roots, run_flags, results):
        t = self.next
        self.next += 1
        self.tickets[t] = (g, np.array(roots, np.uint32), results)
        return t

    def _gather(self, ptr: int, row_bytes: int, n_rows: int):
        words = row_bytes // 4
        full = torch.from_numpy(_view(ptr, n_rows * words, ctypes.c_uint32, np.int32))
        lo, hi = shard_bounds(n_rows, self.world, self.rank)
        assert (hi - lo) * self.world == n_rows, "equal slices expected by the stub"
        dist.all_gather_into_tensor(full, full[lo * words:hi * words].clone())

    def run_wait(self, ticket, results, gather):
        g, roots, res = self.tickets.pop(ticket)
        self._compute(g, roots, res)
        if self.sharded and gather & GATHER_DIST:
            self._gather(results[0]["dist"], g["n"] * 4, len(roots))

    def run(self, g, roots, run_flags, results, gather):
        self.run_wait(self.run_async(g, roots, run_flags, results), results, gather)

    def wait(self):
        pass

    def allgather_rows(self, table_ptrs, row_bytes, n_roots):
        self._gather(table_ptrs[0], row_bytes, n_roots)

    def stats(self, i=0):
        return {"ms_relax": 0.01, "ms_dag": 0.0, "ms_finish": 0.001, "ms_total": 0.012, "n_relax_launches": 1, "n_dag_launches": 0,
                "n_exact_roots": 0, "state_bytes": 4, "narrow_overflow": 0, "dbg": [0, 0, 0, 0], "rows_recomputed": self.last.get("n", 0)}

    def close(self):
        pass
