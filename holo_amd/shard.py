"""Root sharding across the GPUs of one node.

SPF roots are independent units over a read-only graph: the CSR is replicated on every GPU, rank g
gets a contiguous slice of the root list (sizes differ by at most one 64-root wavefront batch), and
ONE collective — an all-gather of the per-root result slabs over RCCL/xGMI — gives every rank the
tables of all roots.  The reference has no analogue (its only multi-root caller is the sequential
loop of holo-isis/src/flooding/manet.rs:47-69).  torch.distributed is plumbing here: `nccl` (= RCCL)
on GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

BATCH = 64          # roots per wavefront batch (lane = root)


def shard_bounds(n_roots: int, world: int) -> List[Tuple[int, int]]:
    """[begin, end) of every rank's slice — the C ABI's hspf_shard_bounds (include/holo_spf_hip.h), the ONE place the
    slicing rule lives: whole 64-root batches dealt out as evenly as possible, the ragged tail batch to the last rank
    that has work, empty slices beyond the number of batches.  (Pure host arithmetic: the library loads without a GPU.)"""
    from . import engine as E
    return [E.shard_bounds(n_roots, world, r) for r in range(world)]


def plan_areas(roots_per_area: Sequence[int], world: int):
    """Areas first, then roots (BASELINE configs[3]) — the C ABI's hspf_plan_areas: list of (rank, area, begin, end)."""
    from . import engine as E
    return E.plan_areas(roots_per_area, world)


def shard_roots(roots: Sequence[int], rank: int, world: int) -> np.ndarray:
    lo, hi = shard_bounds(len(roots), world)[rank]
    return np.ascontiguousarray(np.asarray(roots, np.uint32)[lo:hi])


def gather_rows(local, n_roots: int, group=None, async_op: bool = False):
    """All-gather of row-major per-root tables.  `local` is this rank's [r_local, ...] tensor (any
    dtype, CPU for gloo / device for nccl); returns ([n_roots, ...] tensor, work-or-None).  Slices
    are padded to the largest one so that a single all_gather_into_tensor moves everything."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_roots, world)
    width = max(hi - lo for lo, hi in bounds)
    tail = tuple(local.shape[1:])
    dtype = local.dtype
    # exchanged as raw bytes: one code path for every table type (and gloo has no 16-bit integers)
    row_bytes = int(np.prod(tail, dtype=np.int64)) * local.element_size() if tail else local.element_size()
    send = torch.zeros((width, row_bytes), dtype=torch.uint8, device=local.device)
    if local.shape[0]:
        send[:local.shape[0]] = local.contiguous().view(torch.uint8).reshape(local.shape[0], row_bytes)
    recv = torch.empty((world * width, row_bytes), dtype=torch.uint8, device=local.device)
    work = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)

    def finish():
        if all(hi - lo == width for lo, hi in bounds):
            rows = recv[:n_roots]
        else:
            rows = torch.cat([recv[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(bounds)], 0)
        return rows.contiguous().view(dtype).reshape((rows.shape[0],) + tail)
    if async_op:
        return finish, work
    return finish(), None


def run_sharded(engine, graph, roots: Sequence[int], run_flags: int = 0, *, device=None,
                gather: Sequence[str] = ("dist",)):
    """Every rank runs its slice of `roots` on its own engine/graph replica and the requested tables
    ("dist", "hops", "flags", "first_hop_mask") are all-gathered.  Returns dict name -> [R, N(, W)]
    tensor holding ALL roots on every rank."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_roots(roots, rank, world)
    n = graph.n
    W = 1
    tables = {}
    if len(mine):
        res = engine.run(graph, mine, run_flags)
        W = res.first_hop_mask.shape[2]
        tables = {"dist": res.dist.view(np.int32), "hops": res.hops.view(np.int16),
                  "flags": res.flags.view(np.int16), "first_hop_mask": res.first_hop_mask.view(np.int64)}
    # all ranks must agree on the mask width before exchanging masks
    w = torch.tensor([W], dtype=torch.int64, device=device)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    W = int(w.item())
    out = {}
    for name in gather:
        if name in tables:
            t = torch.from_numpy(np.ascontiguousarray(tables[name]))
            if name == "first_hop_mask" and t.shape[2] != W:
                t = torch.nn.functional.pad(t, (0, W - t.shape[2]))
        else:
            shape = (0, n, W) if name == "first_hop_mask" else (0, n)
            dt = {"dist": torch.int32, "hops": torch.int16, "flags": torch.int16, "first_hop_mask": torch.int64}[name]
            t = torch.zeros(shape, dtype=dt)
        if device is not None:
            t = t.to(device)
        out[name], _ = gather_rows(t, len(roots))
    return out
