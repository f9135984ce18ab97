"""hspf_multi_*: the sharded run through the C ABI.  A one-GPU box cannot hold several devices, so the device list
repeats ordinal 0: N contexts and N streams on one GPU, the gather done by device-to-device copies — the same code as on
N GPUs of one process.  The RCCL back end is exercised with a one-rank communicator."""
import numpy as np
import pytest

from holo_amd import synth
from holo_amd import engine as E

pytestmark = pytest.mark.gpu


def _tables(torch, dev, R, n, W):
    return dict(dist=torch.zeros((R, n), dtype=torch.int32, device=dev), hops=torch.zeros((R, n), dtype=torch.int16, device=dev),
                flags=torch.zeros((R, n), dtype=torch.int16, device=dev), mask=torch.zeros((R, n, W), dtype=torch.int64, device=dev))


def _ptrs(t, W):
    return dict(dist=t["dist"].data_ptr(), hops=t["hops"].data_ptr(), flags=t["flags"].data_ptr(), mask=t["mask"].data_ptr(), mask_words=W)


@pytest.mark.parametrize("n_ctx,n_roots", [(1, 100), (2, 128), (3, 300), (4, 70)])
def test_sharded_run_equals_unsharded(spf_ctx, n_ctx, n_roots):
    import torch
    dev = torch.device("cuda:0")
    g = synth.ospf_10k()
    n = g.n
    roots = ((np.arange(n_roots, dtype=np.uint64) * n) // n_roots).astype(np.uint32)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    ref = spf_ctx.run(G, roots, E.RUN_NET_NEXTHOPS)
    G.free()
    W = ref.first_hop_mask.shape[2]
    m = E.MultiEngine([0] * n_ctx)
    try:
        mg = m.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        assert m.mask_words(mg, roots) == W
        tabs = [_tables(torch, dev, n_roots, n, W) for _ in range(n_ctx)]
        for mode in (E.GATHER_DIST | E.GATHER_HOPS | E.GATHER_FLAGS | E.GATHER_MASK,
                     E.GATHER_DIST | E.GATHER_HOPS | E.GATHER_FLAGS | E.GATHER_MASK | E.GATHER_ASYNC):
            for t in tabs:
                for x in t.values():
                    x.zero_()
            torch.cuda.synchronize()
            m.run(mg, roots, E.RUN_NET_NEXTHOPS, [_ptrs(t, W) for t in tabs], mode)
            m.wait()
            for t in tabs:       # every "device" holds every root's rows
                assert np.array_equal(t["dist"].cpu().numpy().view(np.uint32), ref.dist)
                assert np.array_equal(t["hops"].cpu().numpy().view(np.uint16), ref.hops)
                assert np.array_equal(t["flags"].cpu().numpy().view(np.uint16) & 1, ref.flags & 1)
                assert np.array_equal(t["mask"].cpu().numpy().view(np.uint64), ref.first_hop_mask)
        # no gather: each context holds its own slice only
        for t in tabs:
            t["dist"].zero_()
        torch.cuda.synchronize()
        m.run(mg, roots, E.RUN_NET_NEXTHOPS, [_ptrs(t, W) for t in tabs], 0)
        for i, t in enumerate(tabs):
            lo, hi = E.shard_bounds(n_roots, n_ctx, i)
            d = t["dist"].cpu().numpy().view(np.uint32)
            assert np.array_equal(d[lo:hi], ref.dist[lo:hi])
            if n_ctx > 1:
                assert (np.delete(d, np.s_[lo:hi], axis=0) == 0).all()
        m.free_graph(mg)
    finally:
        m.close()


def test_allgather_of_route_tables_generic_rows():
    """hspf_multi_allgather_rows on an arbitrary per-root table (what a caller does with hspf_routes_device output)."""
    import torch
    dev = torch.device("cuda:0")
    n_ctx, n_roots, row = 3, 200, 37
    m = E.MultiEngine([0] * n_ctx)
    try:
        full = torch.arange(n_roots * row, dtype=torch.int32, device=dev).reshape(n_roots, row)
        tabs = []
        for i in range(n_ctx):
            lo, hi = E.shard_bounds(n_roots, n_ctx, i)
            t = torch.full((n_roots, row), -1, dtype=torch.int32, device=dev)
            t[lo:hi] = full[lo:hi]
            tabs.append(t)
        torch.cuda.synchronize()
        m.allgather_rows([t.data_ptr() for t in tabs], row * 4, n_roots)
        for t in tabs:
            assert torch.equal(t, full)
    finally:
        m.close()


def test_rccl_backend_loads_and_gathers_with_one_rank():
    """The RCCL path of the library (dlopen of librccl.so, communicator from a unique id, in-place ncclAllGather) with a
    one-rank job: what every process of `bench.py --gpus N` does, minus the other ranks."""
    import torch
    dev = torch.device("cuda:0")
    uid = E.multi_unique_id()
    assert len(uid) == 128
    m = E.MultiEngine([0], world=1, first_rank=0, unique_id=uid)
    try:
        t = torch.arange(64 * 10, dtype=torch.int32, device=dev).reshape(64, 10)
        want = t.clone()
        torch.cuda.synchronize()
        m.allgather_rows([t.data_ptr()], 40, 64)
        assert torch.equal(t, want)
    finally:
        m.close()
