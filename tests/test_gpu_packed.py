"""GPU parity of the packed hand-off (ABI 7: hspf_run_packed / _device / _async): ONE word per (root, vertex), decoded with the
layout the run reports, against the CPU oracle bit for bit — on every kernel path that can produce it (lean sweep, k_fused
4- and 8-byte, one-workgroup kernel, lane = vertex kernel, sequential-kernel rows), into page-locked and pageable host
memory and into device memory; and the refusals (HSPF_E_NO_PACKED) where a word cannot hold the result.
"""
import ctypes
import os

import numpy as np
import pytest

from holo_amd import synth
from holo_amd import engine as E
from holo_amd import _lib as L
from oracle import graph_oracle as go

from _engines import both_engines  # noqa: E402

pytestmark = pytest.mark.gpu
ORACLE_THREADS = min(64, os.cpu_count() or 1)


def decode_c(pr: E.PackedResult):
    """The header's own inline rule (hspf_packed_*), spelled out per word in numpy — independent of PackedResult's accessors."""
    w = pr.words.astype(np.uint64)
    inn = w < np.uint64(pr.not_reached)
    dist = np.where(inn, (w >> np.uint64(pr.dist_shift)) & np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hops = np.where(inn, (w >> np.uint64(pr.hops_shift)) & np.uint64(pr.hops_mask), 0).astype(np.uint16)
    mask = np.where(inn, w & np.uint64((1 << pr.mask_bits) - 1), 0).astype(np.uint64)
    return inn, dist, hops, mask


def check_packed(ctx, g, roots, run_flags=0, buffer=None, expect_words=None, oracle_variant=go.MAP, threads=1):
    roots = np.asarray(roots, np.uint32)
    G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    try:
        pr = ctx.run_packed(G, roots, run_flags, buffer=buffer)
        full = ctx.run(G, roots, run_flags)
    finally:
        G.free()
    oflags = run_flags & (E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, oflags, oracle_variant, mask_words_=1, threads=threads)
    inn, dist, hops, mask = decode_c(pr)
    assert np.array_equal(inn, ref.flags.astype(bool)), "in-SPT"
    assert np.array_equal(dist, ref.dist), "dist"
    assert np.array_equal(hops, ref.hops), "hops"
    assert np.array_equal(mask, ref.mask[..., 0]), "first-hop mask"
    # the wrapper's accessors, and the same values as hspf_run
    assert np.array_equal(pr.dist, full.dist) and np.array_equal(pr.hops, full.hops)
    assert np.array_equal(pr.first_hop_mask[..., 0], full.first_hop_mask[..., 0])
    assert np.array_equal(pr.in_spt, (full.flags & 1).astype(bool))
    exact_rows = ((full.flags & E.RF_EXACT) != 0).any(axis=1)
    assert np.array_equal((pr.root_status & E.ROOT_EXACT) != 0, exact_rows), "root status"
    if expect_words is not None:
        assert pr.word_bytes == expect_words, (pr.word_bytes, pr.stats)
    return pr


@both_engines
@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("run_flags", [0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD])
def test_random_lsdb(spf_ctx, seed, run_flags):
    g = synth.random_lsdb(70, 6, 3.0, 900 + seed, metric_hi=6)
    roots = np.arange(6, 6 + 45, dtype=np.uint32)
    try:
        check_packed(spf_ctx, g, roots, run_flags)
    except E.HspfError as e:                   # a LAN of the random graph may give a root more than 24 slots: refused, not wrong
        assert e.code == E.E_NO_PACKED
        G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        assert max(G.slot_table(int(r))[2] for r in roots) > 16
        G.free()


@both_engines
@pytest.mark.parametrize("seed", range(4))
def test_routers_only_ragged_roots_with_padding(spf_ctx, seed):
    g = synth.random_lsdb(120, 0, 3.0, 950 + seed, metric_hi=9)
    roots = np.arange(100, dtype=np.uint32)
    roots[7] = E.NO_ROOT
    check_packed(spf_ctx, g, roots)


@both_engines
@pytest.mark.parametrize("seed", range(4))
def test_zero_cost_links_rows_of_the_sequential_kernel(spf_ctx, seed):
    g = synth.random_lsdb(50, 0, 3.0, 200 + seed, metric_hi=3, zero_cost_router_links=True)
    roots = np.arange(6, 36, dtype=np.uint32)
    pr = check_packed(spf_ctx, g, roots)
    assert pr.stats["n_exact_roots"] + pr.stats["n_repaired_roots"] == int(((pr.root_status & 1) != 0).sum())
    assert pr.stats["n_exact_roots"] == 0          # round 6: k_repair, not the sequential kernel


@both_engines
def test_forced_exact(spf_ctx):
    g = synth.random_lsdb(40, 0, 3.0, 401, metric_hi=5)
    pr = check_packed(spf_ctx, g, np.arange(5, 15, dtype=np.uint32), E.RUN_FORCE_EXACT)
    assert (pr.root_status & 1).all()


def test_pop_rank_is_refused(spf_ctx):
    g = synth.random_lsdb(40, 0, 3.0, 402, metric_hi=5)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    with pytest.raises(E.HspfError) as ei:
        spf_ctx.run_packed(G, [1, 2], E.RUN_POP_RANK)
    assert ei.value.code == -1
    G.free()


def test_more_than_24_slots_is_no_packed(spf_ctx):
    """A root on a LAN of 40 routers has 40 first-hop slots: no word holds its masks."""
    n_r, lan = 60, 40
    pn = 0                                                       # vertex 0 = the pseudonode (networks sort first)
    s, d, m = [], [], []
    for r in range(1, lan + 1):
        s += [r, pn]; d += [pn, r]; m += [10, 0]
    for r in range(lan + 1, n_r):
        s += [r, r - 1]; d += [r - 1, r]; m += [3, 3]
    row_ptr, col, metric = synth._csr_from_links(n_r, np.array(s), np.array(d), np.array(m))
    vf = np.zeros(n_r, np.uint8); vf[0] = 1
    g = synth.CsrGraph(row_ptr, col, metric, vf, synth.MAX_PATH_METRIC_WIDE)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    with pytest.raises(E.HspfError) as ei:
        spf_ctx.run_packed(G, [1, 2, 50], E.RUN_NET_NEXTHOPS)
    assert ei.value.code == E.E_NO_PACKED
    pr = spf_ctx.run_packed(G, [50, 55], E.RUN_NET_NEXTHOPS)     # roots off the LAN: one or two slots
    full = spf_ctx.run(G, [50, 55], E.RUN_NET_NEXTHOPS)
    assert np.array_equal(pr.dist, full.dist) and np.array_equal(pr.first_hop_mask[..., 0], full.first_hop_mask[..., 0])
    G.free()


def test_headline_graph_four_byte_words_pinned_and_pageable(spf_ctx):
    """isis-100k x 64 roots (BASELINE configs[2]): 4-byte words (25.6 MB instead of 102 MB), every root against the oracle,
    through a page-locked buffer (one copy) and a pageable one (staged in 8 MB blocks)."""
    g = synth.isis_100k()
    n = g.n
    roots = ((np.arange(64, dtype=np.int64) * n) // 64).astype(np.uint32)
    pinned = spf_ctx.host_alloc(8 * 64 * n)
    pr = check_packed(spf_ctx, g, roots, 0, buffer=pinned, expect_words=4, oracle_variant=go.HEAP, threads=ORACLE_THREADS)
    assert pr.stats["dbg"][0] == 1, "the lean sweep took the run"
    a = pr.words.copy()
    d0, h0, m0 = pr.dist[0].copy(), pr.hops[0].copy(), pr.first_hop_mask[0].copy()     # (pr.words is a view of `pinned`)
    pageable = np.zeros(8 * 64 * n + 64, np.uint8)[64:]
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    pr2 = spf_ctx.run_packed(G, roots, 0, buffer=pageable)
    assert pr2.word_bytes == 4 and np.array_equal(pr2.words, a)
    # one root (lane = vertex kernel on the default engine): 8-byte words through k_pack_full
    pr1 = spf_ctx.run_packed(G, roots[:1], 0, buffer=pinned)
    assert np.array_equal(pr1.dist[0], d0) and np.array_equal(pr1.hops[0], h0) and np.array_equal(pr1.first_hop_mask[0], m0)
    G.free()
    pinned.free()


def test_async_tickets_overlap_copy_and_compute(spf_ctx):
    g = synth.random_lsdb(3000, 0, 3.0, 77, metric_hi=20)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    sets = [np.arange(k, k + 64, dtype=np.uint32) for k in (0, 100, 500, 900, 1500)]
    bufs = [spf_ctx.host_alloc(8 * 64 * g.n) for _ in sets]
    hs = [spf_ctx.run_packed_async(G, r, 0, b) for r, b in zip(sets, bufs)]
    for h, r in zip(hs, sets):
        pr = spf_ctx.wait_packed(h)
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, r, 0, go.HEAP, mask_words_=1)
        assert np.array_equal(pr.dist, ref.dist) and np.array_equal(pr.hops, ref.hops)
        assert np.array_equal(pr.first_hop_mask[..., 0], ref.mask[..., 0]) and np.array_equal(pr.in_spt, ref.flags.astype(bool))
    G.free()
    for b in bufs:
        b.free()


def test_device_destination(spf_ctx):
    import torch
    g = synth.random_lsdb(2000, 0, 3.0, 78, metric_hi=20)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.arange(10, 10 + 130, dtype=np.uint32)
    t = torch.empty(8 * len(roots) * g.n, dtype=torch.uint8, device="cuda:0")
    ly, status, st = spf_ctx.run_packed_device(G, roots, 0, words_ptr=t.data_ptr(), cap_bytes=t.numel())
    host = t.cpu().numpy()[: ly.word_bytes * len(roots) * g.n].view(np.uint32 if ly.word_bytes == 4 else np.uint64).reshape(len(roots), g.n)
    pr = E.PackedResult(host, ly.word_bytes, ly.dist_shift, ly.hops_shift, ly.hops_mask, ly.mask_bits, ly.not_reached, status, st)
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, 0, go.HEAP, mask_words_=1)
    assert np.array_equal(pr.dist, ref.dist) and np.array_equal(pr.hops, ref.hops) and np.array_equal(pr.first_hop_mask[..., 0], ref.mask[..., 0])
    G.free()


def test_buffer_too_small_is_invalid(spf_ctx):
    g = synth.random_lsdb(200, 0, 3.0, 79, metric_hi=20)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    with pytest.raises(E.HspfError) as ei:
        spf_ctx.run_packed(G, np.arange(64, dtype=np.uint32), 0, buffer=np.zeros(64 * 200 * 2, np.uint8))
    assert ei.value.code == -1
    G.free()


@both_engines
def test_device_buffer_too_small_is_refused_before_anything_is_written(spf_ctx):
    """a short DEVICE destination used to be overrun by the emit and refused afterwards.  A canary behind the short
    buffer must survive the refused call."""
    import torch
    g = synth.random_lsdb(2000, 0, 3.0, 78, metric_hi=20)
    G = spf_ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
    roots = np.arange(10, 10 + (8 if spf_ctx.mode == "xcd" else 70), dtype=np.uint32)
    short = 2 * len(roots) * g.n                                   # half of what 4-byte words need
    t = torch.full((8 * len(roots) * g.n,), 0xA5, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(E.HspfError) as ei:
        spf_ctx.run_packed_device(G, roots, 0, words_ptr=t.data_ptr(), cap_bytes=short)
    assert ei.value.code == -1
    assert bool((t[short:] == 0xA5).all()), "bytes behind the caller's buffer were written"
    G.free()


def test_large_costs_take_eight_byte_words(spf_ctx):
    """Costs up to 2^20: the 4-byte fields cannot hold the distances — the run is redone wide and says so in the layout."""
    g = synth.random_lsdb(300, 0, 3.0, 80, metric_hi=1 << 20)
    check_packed(spf_ctx, g, np.arange(0, 70, dtype=np.uint32), expect_words=8)
