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
