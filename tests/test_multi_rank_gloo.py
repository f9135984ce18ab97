"""N > 1 path on CPU: two gloo ranks shard the roots, run their slices (engine = the CPU oracle behind
the engine interface) and all-gather the tables; every rank must end up with exactly what a single
process computes for all roots — including ragged root counts and ranks with no work."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from holo_amd import shard, synth
from oracle import graph_oracle as go

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_cover_and_align():
    for n_roots in (0, 1, 63, 64, 65, 128, 129, 1000, 10000):
        for world in (1, 2, 3, 4, 8):
            b = shard.shard_bounds(n_roots, world)
            assert b[0][0] == 0 and b[-1][1] == n_roots
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(lo % 64 == 0 for lo, hi in b if hi > lo)
            sizes = [(hi - lo + 63) // 64 for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_roots, q):
    import sys
    sys.path.insert(0, HERE)
    from _oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth.random_lsdb(120, 12, 3.0, 7, metric_hi=6)
        roots = (np.arange(n_roots, dtype=np.uint32) * 5) % g.n
        eng = OracleEngine()
        G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        out = shard.run_sharded(eng, G, roots, 0, gather=("dist", "hops", "first_hop_mask"))
        q.put((rank, {k: v.numpy() for k, v in out.items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_roots", [130, 64, 3])
def test_two_rank_sharded_run_equals_single_process(n_roots):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_roots, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = synth.random_lsdb(120, 12, 3.0, 7, metric_hi=6)
    roots = (np.arange(n_roots, dtype=np.uint32) * 5) % g.n
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.MAP)
    for rank in range(world):
        assert np.array_equal(got[rank]["dist"].view(np.uint32), ref.dist)
        assert np.array_equal(got[rank]["hops"].view(np.uint16), ref.hops)
        W = got[rank]["first_hop_mask"].shape[2]
        assert np.array_equal(got[rank]["first_hop_mask"].view(np.uint64)[:, :, :ref.mask.shape[2]], ref.mask[:, :, :W])


def _area_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, HERE)
    from _oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        areas = [synth.random_lsdb(60 + 10 * a, 6, 3.0, 100 + a, metric_hi=6) for a in range(3)]
        roots = [np.arange(0, g.n, 1 + a, dtype=np.uint32)[: (70, 130, 20)[a]] for a, g in enumerate(areas)]
        # the ONLY source of slicing: the C ABI's area plan (areas first, then roots; whole 64-root batches)
        plan = [s for s in shard.plan_areas([len(r) for r in roots], world) if s[0] == rank]
        eng = OracleEngine()
        mine = []
        for _, a, lo, hi in plan:
            g = areas[a]
            G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
            res = eng.run(G, roots[a][lo:hi], 0)
            mine.append((a, lo, hi, res.dist.copy(), res.hops.copy()))
        everything = [None] * world
        dist.all_gather_object(everything, mine)              # the exchange step (RCCL all-gather of per-root tables on GPUs)
        q.put((rank, everything))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_area_plan_from_the_c_abi_covers_every_root_once():
    """configs[3] shape on CPU: areas sharded over 2 ranks by hspf_plan_areas, each rank runs ITS (area, root range)
    slices, one exchange, and every rank holds every root's table exactly once, equal to the single-process run."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_area_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    areas = [synth.random_lsdb(60 + 10 * a, 6, 3.0, 100 + a, metric_hi=6) for a in range(3)]
    roots = [np.arange(0, g.n, 1 + a, dtype=np.uint32)[: (70, 130, 20)[a]] for a, g in enumerate(areas)]
    refs = [go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, r, 0, go.MAP) for g, r in zip(areas, roots)]
    for rank in range(world):
        seen = [np.zeros(len(r), np.int32) for r in roots]
        for part in got[rank]:
            for a, lo, hi, d, h in part:
                seen[a][lo:hi] += 1
                assert np.array_equal(d, refs[a].dist[lo:hi]) and np.array_equal(h, refs[a].hops[lo:hi])
        assert all((s == 1).all() for s in seen)
