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


def test_rccl_ragged_branch_grouped_broadcasts_with_one_rank(monkeypatch):
    """The ragged-slices branch of the RCCL back end (one grouped ncclBroadcast per rank instead of the in-place
    ncclAllGather), forced with HSPF_GATHER_FORCE_BCAST on a one-rank communicator and a root count that is not a
    multiple of 64: the only way to run it on one GPU (RCCL refuses two ranks on one device, see
    profiles/r03_rccl_two_ranks_one_gpu.txt)."""
    import torch
    dev = torch.device("cuda:0")
    monkeypatch.setenv("HSPF_GATHER_FORCE_BCAST", "1")
    m = E.MultiEngine([0], world=1, first_rank=0, unique_id=E.multi_unique_id())
    try:
        t = torch.arange(70 * 13, dtype=torch.int32, device=dev).reshape(70, 13)
        want = t.clone()
        torch.cuda.synchronize()
        m.allgather_rows([t.data_ptr()], 13 * 4, 70)
        assert torch.equal(t, want)
        g = synth.ospf_10k()
        roots = ((np.arange(70, dtype=np.uint64) * g.n) // 70).astype(np.uint32)
        mg = m.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        W = m.mask_words(mg, roots)
        tab = _tables(torch, dev, 70, g.n, W)
        m.run(mg, roots, E.RUN_NET_NEXTHOPS, [_ptrs(tab, W)], E.GATHER_DIST | E.GATHER_HOPS | E.GATHER_ASYNC)
        m.wait()
        ctx = E.SpfContext(0)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        ref = ctx.run(G, roots, E.RUN_NET_NEXTHOPS)
        assert np.array_equal(tab["dist"].cpu().numpy().view(np.uint32), ref.dist)
        assert np.array_equal(tab["hops"].cpu().numpy().view(np.uint16), ref.hops)
        G.free(); ctx.close(); m.free_graph(mg)
    finally:
        m.close()


def test_two_async_gathers_into_the_same_tables_back_to_back():
    """a second asynchronous run into tables whose gather is still in flight must wait for exactly that
    gather (events are kept per local device), and a third run reuses the recycled events."""
    import torch
    dev = torch.device("cuda:0")
    g = synth.ospf_10k()
    n, n_ctx, n_roots = g.n, 3, 192
    m = E.MultiEngine([0] * n_ctx)
    try:
        mg = m.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        sets = [((np.arange(n_roots, dtype=np.uint64) * n) // n_roots + k * 7).astype(np.uint32) % n for k in range(3)]
        W = max(m.mask_words(mg, r) for r in sets)
        tabs = [_tables(torch, dev, n_roots, n, W) for _ in range(n_ctx)]
        mode = E.GATHER_DIST | E.GATHER_HOPS | E.GATHER_ASYNC
        for roots in sets:                      # no wait in between: run k + 1 must not overtake gather k
            m.run(mg, roots, E.RUN_NET_NEXTHOPS, [_ptrs(t, W) for t in tabs], mode)
        m.wait()
        ctx = E.SpfContext(0)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        ref = ctx.run(G, sets[-1], E.RUN_NET_NEXTHOPS)
        for t in tabs:
            assert np.array_equal(t["dist"].cpu().numpy().view(np.uint32), ref.dist)
            assert np.array_equal(t["hops"].cpu().numpy().view(np.uint16), ref.hops)
        G.free(); ctx.close(); m.free_graph(mg)
    finally:
        m.close()


def _rccl_rank(rank, uid, q):
    try:
        from holo_amd import engine as E2
        m = E2.MultiEngine([0], world=2, first_rank=rank, unique_id=uid)
        q.put((rank, "ok", ""))
        m.close()
    except Exception as ex:     # noqa: BLE001
        q.put((rank, "error", str(ex)))


def test_two_rccl_ranks_on_one_gpu_are_refused_or_work(tmp_path):
    """Two PROCESSES, one GPU, hspf_multi_init(world = 2) from a shared communicator id.  RCCL refuses two ranks on one
    device ("Duplicate GPU detected"): then the library must fail cleanly (an HspfError, no hang, no crash) and the
    text goes to gpurun_out/ for the record; should a future RCCL allow it, both ranks must come up."""
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = E.multi_unique_id()
    procs = [ctx.Process(target=_rccl_rank, args=(r, uid, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(2):
            got.append(q.get(timeout=90))
    except Exception:           # noqa: BLE001   (a hang is a failure of the library, not of the test harness)
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.fail("hspf_multi_init with two ranks on one GPU did not return within 90 s")
    for p in procs:
        p.join(timeout=30)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/rccl_two_ranks_one_gpu.txt", "w") as f:
        for r in sorted(got):
            f.write(f"rank {r[0]}: {r[1]} {r[2]}\n")
    states = {r[1] for r in got}
    assert states in ({"ok"}, {"error"}), got
