"""Randomised differential run on the GPU box: many small adversarial LSDBs (LANs of every size, one-way / parallel /
zero-cost links, overloaded and non-expandable vertices, hop-count mode, tight max-path metrics, ragged root lists with
padding entries, network vertices as roots), every run flag combination, optional row patches in between — the HIP
engine against the CPU oracle, bit for bit.  Not part of the pytest suite (minutes, not seconds):

    python tools/gpu_fuzz.py [first_seed] [n_graphs]
    FUZZ_WIDE=n python tools/gpu_fuzz.py [first_seed]      (n graphs with LANs of 150-900 routers: up to 15 mask words)
    FUZZ_MID=n python tools/gpu_fuzz.py [first_seed]       (n graphs of 2 000-19 500 vertices, 1-8 roots per run: k_xcd's range)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holo_amd import _lib                      # noqa: E402
if os.environ.get("HSPF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HSPF_LIB"])
from holo_amd import synth                     # noqa: E402
from holo_amd import engine as E               # noqa: E402
from oracle import graph_oracle as go          # noqa: E402


PATHS = {"k_xcd": 0, "other": 0, "repaired": 0, "exact": 0}   # which kernel took the runs (fuzz_mid reports it: a k_xcd run that gave up shows here);
                                                                # roots put right by k_repair / re-run by the sequential kernel


def compare(ctx, G, g, roots, flags, tag):
    try:
        res = ctx.run(G, roots, flags)
        PATHS["k_xcd" if res.stats.get("single_wg") == 2 else "other"] += 1
        PATHS["repaired"] += res.stats.get("n_repaired_roots", 0); PATHS["exact"] += res.stats.get("n_exact_roots", 0)
    except E.HspfError as e:
        if e.code == -5:                       # documented limit: more than 1024 first-hop slots (16 mask words)
            return True
        raise
    ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, flags & 3, go.MAP,
                 mask_words_=res.first_hop_mask.shape[2])
    bad = []
    if not np.array_equal(res.dist, ref.dist): bad.append("dist")
    if not np.array_equal(res.hops, ref.hops): bad.append("hops")
    if not np.array_equal(res.flags & 1, ref.flags): bad.append("flags")
    if not np.array_equal(res.first_hop_mask, ref.mask): bad.append("mask")
    if res.pop_rank is not None and not np.array_equal(res.pop_rank, ref.pop_rank): bad.append("pop_rank")
    # round 5: the same run through the packed hand-off (hspf_run_packed); a run whose results do not fit packed words must
    # say so (HSPF_E_NO_PACKED), never deliver something else
    if not (flags & E.RUN_POP_RANK) and os.environ.get("FUZZ_PACKED", "1") != "0":
        try:
            pr = ctx.run_packed(G, roots, flags)
            if not np.array_equal(pr.dist, ref.dist): bad.append("packed dist")
            if not np.array_equal(pr.hops, ref.hops): bad.append("packed hops")
            if not np.array_equal(pr.in_spt, ref.flags.astype(bool)): bad.append("packed in-SPT")
            if res.first_hop_mask.shape[2] != 1 or not np.array_equal(pr.first_hop_mask[..., 0], ref.mask[..., 0]): bad.append("packed mask")
        except E.HspfError as e:
            if e.code != E.E_NO_PACKED:
                raise
            slots = max((G.slot_table(int(r))[2] for r in np.unique(roots) if r != E.NO_ROOT), default=0)
            fused_off = (int(os.environ.get("HSPF_VARIANT", "0") or "0", 0) & 1) or getattr(ctx, "mode", "") == "widemask"   # the A/B switch that turns
            # the fused path off (tests/conftest.py "widemask"): every packed request is refused then, as the header says (include/holo_spf_hip.h)
            if slots <= 16 and not fused_off:  # (17-24 slots: refused once the graph has seen a hop-field overflow)
                bad.append("packed refused a run whose roots have at most 16 first-hop slots")
    if bad:
        print("MISMATCH", tag, bad, "stats", res.stats, flush=True)
        if os.environ.get("FUZZ_DUMP"):
            dump_mismatch(ctx, G, g, roots, flags, res, ref)
    return not bad


def dump_mismatch(ctx, G, g, roots, flags, res, ref):
    """Where the tables differ, and whether the same run differs again (FUZZ_DUMP=1: an intermittent mismatch of round 6)."""
    for name, got, want in (("dist", res.dist, ref.dist), ("hops", res.hops, ref.hops), ("mask", res.first_hop_mask[..., 0], ref.mask[..., 0])):
        d = np.argwhere(got != want)
        if not len(d):
            continue
        rr, vv = int(d[0][0]), int(d[0][1])
        print("  ", name, "differs at", len(d), "places; roots", sorted(set(d[:, 0].tolist()))[:10], "first: slot", rr, "root", int(roots[rr]), "vertex", vv,
              "got", int(got[rr, vv]), "want", int(want[rr, vv]), "dist", int(ref.dist[rr, vv]), "exact flag", int(res.flags[rr, vv] & 2), flush=True)
        print("   vertices of that root:", d[d[:, 0] == rr][:12, 1].tolist(), "vflags", int(g.vflags[vv]))
        ins = [(int(u), int(g.metric[kx])) for u in range(g.n) for kx in range(int(g.row_ptr[u]), int(g.row_ptr[u + 1])) if g.col[kx] == vv]
        print("   links into it (source, cost, ref dist, ref hops, got hops, ref rank):",
              [(u, c, int(ref.dist[rr, u]), int(ref.hops[rr, u]), int(res.hops[rr, u]), int(ref.pop_rank[rr, u]) if ref.pop_rank is not None else None) for u, c in ins][:20], flush=True)
        break
    for again in range(3):
        r2 = ctx.run(G, roots, flags)
        print("   again:", {"dist": int((r2.dist != ref.dist).sum()), "hops": int((r2.hops != ref.hops).sum()), "mask": int((r2.first_hop_mask != ref.mask).sum())},
              "repaired", r2.stats.get("n_repaired_roots"), "sweeps", r2.stats.get("repair_sweeps"), flush=True)


def fuzz(ctx, first, count, verbose=True):
    ok = runs = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(10_000 + seed)
        nr = int(rng.integers(5, 260)) if rng.random() > 0.04 else int(rng.integers(800, 3000))
        nn = int(rng.integers(0, 14))
        hop = rng.random() < 0.2
        zero = os.environ.get("FUZZ_ZERO") is not None              # round 6: every graph with zero-cost router links and tie-heavy costs (dynamic pop orders)
        g = synth.random_lsdb(nr, nn, float(rng.uniform(1.2, 4.5)), 50_000 + seed,
                              metric_lo=1, metric_hi=int(rng.integers(1, 5 if zero else 40)),
                              max_path=(1023 if rng.random() < 0.15 else (0xFFFFFFFF if rng.random() < 0.3 else synth.MAX_PATH_METRIC_WIDE)),
                              p_oneway=float(rng.choice([0.0, 0.03, 0.3])), p_parallel=float(rng.choice([0.0, 0.05, 0.4])),
                              p_overload=float(rng.choice([0.0, 0.03, 0.3])), p_noexpand=float(rng.choice([0.0, 0.02, 0.2])),
                              zero_cost_router_links=bool(rng.random() < 0.15) or zero, lan_size=int(rng.choice([2, 3, 5, 8, 14, 20, 30, 45, 70, 140])), hopcount=hop)
        if rng.random() < 0.15 and not hop:                       # large costs: 8-byte state, max-path pruning, u32 saturation
            g.metric = (g.metric.astype(np.uint64) << int(rng.integers(8, 25))).clip(0, 0xFFFFFFFE).astype(np.uint32)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for rep in range(3):
            k = int(rng.integers(1, min(g.n, 200 if g.n < 800 else 700) + 1))
            if os.environ.get("FUZZ_MAX_ROOTS"):                     # (the one-XCD-per-root kernel takes runs of at most eight roots)
                k = min(k, int(rng.integers(1, int(os.environ["FUZZ_MAX_ROOTS"]) + 1)))
            roots = rng.choice(g.n, size=k, replace=rng.random() < 0.2).astype(np.uint32)
            if k > 3 and rng.random() < 0.3:
                roots[int(rng.integers(0, k))] = E.NO_ROOT
            flags = int(rng.choice([0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD, E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD]))
            if hop:
                flags |= E.RUN_IGNORE_OVERLOAD
            if rng.random() < (0.3 if zero else 0.08):
                flags |= E.RUN_POP_RANK
            runs += 1
            t_run = time.time()
            ok += compare(ctx, G, g, roots, flags, (seed, rep, flags, k))
            if rng.random() < 0.12:                                  # an instance repeating its run: the lean sweep learns its mode schedule
                for again in range(4):                                # (plain, learning, scheduled, scheduled)
                    runs += 1
                    ok += compare(ctx, G, g, roots, flags, (seed, rep, flags, k, "repeat", again))
            if os.environ.get("FUZZ_TIMING") and time.time() - t_run > float(os.environ["FUZZ_TIMING"]):
                print(f"SLOW seed {seed} rep {rep} n {g.n} roots {k} flags {flags} hop {hop} maxpath {g.max_path_metric:#x}: {time.time() - t_run:.2f} s, stats {ctx.stats()}", flush=True)
            if rep < 2 and rng.random() < 0.5:                      # re-originate a few rows in between
                vs = np.sort(rng.choice(g.n, size=int(rng.integers(1, 6)), replace=False))
                rows, fl = [], []
                costs_only = rng.random() < 0.5                     # a metric change: applied in place (build mode 2)
                for v in vs.tolist():
                    c = g.col[g.row_ptr[v]:g.row_ptr[v + 1]]; m = g.metric[g.row_ptr[v]:g.row_ptr[v + 1]]
                    keep = rng.random(len(c)) > (-1.0 if costs_only else 0.25)
                    c, m = c[keep], m[keep].copy()
                    if len(m) and (v >= nn or costs_only):
                        m[rng.random(len(m)) < 0.5] = int(rng.integers(0 if costs_only else 1, 9)) if not hop else int(rng.integers(0, 2))
                    rows.append((c, m)); fl.append(int(g.vflags[v]) ^ (synth.VF_NO_TRANSIT if (v >= nn and not costs_only and rng.random() < 0.3) else 0))
                G.patch(vs, rows, fl)
                if costs_only and int(G.export("build_mode")[0]) != 2 and int(np.diff(G.export("in_ptr")).max(initial=0)) <= 256:
                    print("COST PATCH NOT IN PLACE", seed, rep, flush=True); ok -= 1
                g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
        G.free()
    if verbose:
        print(f"fuzz: {ok}/{runs} runs bit-exact over {count} graphs in {time.time() - t0:.1f} s (roots put right by k_repair: {PATHS['repaired']}, by the sequential kernel: {PATHS['exact']})", flush=True)
    return ok, runs


def fuzz_mid(ctx, first, count, verbose=True):
    """Mid-size LSDBs (2 000 - 19 500 vertices: beyond the one-workgroup kernel, inside k_xcd's 20 000), one to eight roots per
    run — what k_xcd and the choice between it and the sweep engine see in production —, the adversarial ingredients of
    fuzz() and row patches in between."""
    ok = runs = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(170_000 + seed)
        nr = int(rng.integers(2000, 19000))
        nn = int(rng.integers(0, 500))
        hop = rng.random() < 0.2
        g = synth.random_lsdb(nr, nn, float(rng.uniform(1.5, 4.0)), 180_000 + seed, metric_lo=1, metric_hi=int(rng.integers(1, 60)),
                              max_path=(4095 if rng.random() < 0.1 else (0xFFFFFFFF if rng.random() < 0.3 else synth.MAX_PATH_METRIC_WIDE)),
                              p_oneway=float(rng.choice([0.0, 0.03])), p_parallel=float(rng.choice([0.0, 0.05])),
                              p_overload=float(rng.choice([0.0, 0.03, 0.2])), p_noexpand=float(rng.choice([0.0, 0.02])),
                              zero_cost_router_links=bool(rng.random() < 0.1), lan_size=int(rng.choice([2, 4, 8, 20, 40])), hopcount=hop)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for rep in range(4):
            k = int(rng.integers(1, 9))
            roots = rng.choice(g.n, size=k, replace=False).astype(np.uint32)
            if k > 2 and rng.random() < 0.3:
                roots[int(rng.integers(0, k))] = E.NO_ROOT
            flags = int(rng.choice([0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD, E.RUN_NET_NEXTHOPS | E.RUN_IGNORE_OVERLOAD]))
            if hop:
                flags |= E.RUN_IGNORE_OVERLOAD
            for again in range(1 if rep else 7):                       # the first root set seven times: the product context tries both kernels, then chooses
                runs += 1
                ok += compare(ctx, G, g, roots, flags, ("mid", seed, rep, flags, k, again))
            if rep < 3:
                vs = np.sort(rng.choice(g.n, size=int(rng.integers(1, 5)), replace=False))
                rows, fl = [], []
                costs_only = rng.random() < 0.5
                for v in vs.tolist():
                    c = g.col[g.row_ptr[v]:g.row_ptr[v + 1]]; m = g.metric[g.row_ptr[v]:g.row_ptr[v + 1]]
                    keep = rng.random(len(c)) > (-1.0 if costs_only else 0.25)
                    c, m = c[keep], m[keep].copy()
                    if len(m) and (v >= nn or costs_only):
                        m[rng.random(len(m)) < 0.5] = int(rng.integers(0 if costs_only else 1, 9)) if not hop else int(rng.integers(0, 2))
                    rows.append((c, m)); fl.append(int(g.vflags[v]))
                G.patch(vs, rows, fl)
                g = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric, g.name, g.meta)
        G.free()
    if verbose:
        print(f"fuzz_mid: {ok}/{runs} runs bit-exact over {count} graphs in {time.time() - t0:.1f} s (k_xcd took {PATHS['k_xcd']} of them)", flush=True)
    return ok, runs


def fuzz_wide(ctx, first, count, verbose=True):
    """Many first-hop slots: one to three LANs of 150-900 routers (up to 15 mask words, k_fw<W> for every W, rows far beyond
    32 in-links = work units; 1 000+ members = hub-mode graph build), roots on and off the LANs, LAN vertices as roots."""
    ok = runs = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(130_000 + seed)
        lan = int(rng.choice([150, 260, 400, 640, 900]))
        nr = lan + int(rng.integers(0, 300)) if rng.random() < 0.5 else lan * int(rng.integers(2, 5))   # most routers off the LAN:
                                                                  # one mask word, the LAN's row is a giant row of k_fused
        nn = int(rng.integers(1, 4))
        hop = rng.random() < 0.15
        g = synth.random_lsdb(nr, nn, float(rng.uniform(1.0, 3.0)), 140_000 + seed, metric_lo=1, metric_hi=int(rng.integers(1, 12)),
                              p_oneway=float(rng.choice([0.0, 0.05])), p_parallel=float(rng.choice([0.0, 0.1])),
                              p_overload=float(rng.choice([0.0, 0.05])), p_noexpand=float(rng.choice([0.0, 0.02])),
                              zero_cost_router_links=bool(rng.random() < 0.1), lan_size=lan, hopcount=hop)
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for rep in range(2):
            k = int(rng.integers(1, 90))
            roots = rng.choice(g.n, size=k, replace=False).astype(np.uint32)
            if rep == 0:
                roots[0] = int(rng.integers(0, nn))                       # a LAN vertex itself
            flags = int(rng.choice([0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD, 3]))
            if hop:
                flags |= E.RUN_IGNORE_OVERLOAD
            runs += 1
            ok += compare(ctx, G, g, roots, flags, ("wide", seed, rep, flags, k))
        G.free()
    if verbose:
        print(f"fuzz_wide: {ok}/{runs} runs bit-exact over {count} graphs in {time.time() - t0:.1f} s", flush=True)
    return ok, runs


def fuzz_layout(ctx, first, count, verbose=True, spf=False):
    """Device graph construction and row patches: every exported array against the numpy restatement
    (tests/_layout_ref.py), and a patched graph against a fresh upload of the patched CSR."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from _layout_ref import layout
    built = ("twoway", "in_ptr", "in_src", "in_cost", "in_pos", "out_ptr", "out_dst", "out_cost", "out_pos", "rowflags", "leaf")
    derived = ("ell_src", "ell_cost", "ell_out", "summary")
    ok = runs = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(70_000 + seed)
        g = synth.random_lsdb(int(rng.integers(3, 150)), int(rng.integers(0, 10)), float(rng.uniform(1.0, 5.0)), 90_000 + seed,
                              metric_hi=int(rng.integers(1, 10)), p_oneway=float(rng.choice([0.0, 0.1, 0.5])),
                              p_parallel=float(rng.choice([0.0, 0.2, 0.6])), p_overload=float(rng.choice([0.0, 0.3])),
                              p_noexpand=float(rng.choice([0.0, 0.3])), zero_cost_router_links=bool(rng.random() < 0.3),
                              lan_size=int(rng.choice([2, 6, 20, 50])), hopcount=bool(rng.random() < 0.2))
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        for rep in range(3):
            want = layout(G.row_ptr, G.col, G.metric, G.vflags)
            good = all(np.array_equal(G.export(k), want[k]) for k in built) and G.n_edges_kept == len(want["in_src"])
            good = good and all(np.array_equal(G.export(k), getattr(G, k)) for k in ("row_ptr", "col", "metric", "vflags"))
            good = good and np.array_equal(G.export("host_row_ptr"), G.row_ptr) and np.array_equal(G.export("host_col"), G.col)   # the library's host mirrors
            runs += 1
            ok += good
            if not good:
                print("LAYOUT MISMATCH", seed, rep, flush=True)
            n = G.n
            vs = np.sort(rng.choice(n, size=int(rng.integers(1, min(n, 12) + 1)), replace=False))
            rows, fl = [], []
            for v in vs.tolist():
                deg = int(rng.integers(0, 9))
                rows.append((rng.integers(0, n, deg).astype(np.uint32), rng.integers(0, 6, deg).astype(np.uint32)))
                fl.append(int(rng.integers(0, 8)))
            G.patch(vs, rows, fl)
            if rng.random() < 0.6:                             # ... then new costs on some rows, in place; arrays and summary as a fresh upload's
                vs2 = np.sort(rng.choice(n, size=int(rng.integers(1, min(n, 12) + 1)), replace=False))
                rows2 = []
                for v in vs2.tolist():
                    c = G.col[G.row_ptr[v]:G.row_ptr[v + 1]].copy()
                    rows2.append((c, rng.integers(0, 6, len(c)).astype(np.uint32)))
                G.patch(vs2, rows2, G.vflags[vs2].copy())
                F = ctx.upload(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
                good = all(np.array_equal(G.export(k), F.export(k)) for k in built + derived)
                good = good and (int(G.export("build_mode")[0]) == 2 or int(np.diff(G.export("in_ptr")).max(initial=0)) > 256)
                F.free()
                runs += 1
                ok += good
                if not good:
                    print("COST PATCH MISMATCH", seed, rep, flush=True)
            if spf:                                            # SPF on whatever graph the arbitrary rows made
                gg = synth.CsrGraph(G.row_ptr, G.col, G.metric, G.vflags, g.max_path_metric)
                roots = rng.choice(n, size=int(rng.integers(1, min(n, 70) + 1)), replace=False).astype(np.uint32)
                flags = int(rng.choice([0, E.RUN_NET_NEXTHOPS, E.RUN_IGNORE_OVERLOAD, 3]))
                runs += 1
                ok += compare(ctx, G, gg, roots, flags, ("arbitrary", seed, rep, flags))
        G.free()
    if verbose:
        print(f"fuzz_layout: {ok}/{runs} layouts identical over {count} graphs in {time.time() - t0:.1f} s", flush=True)
    return ok, runs


def fuzz_routes(ctx, first, count, verbose=True):
    """hspf_routes_device (IS-IS rule, OSPF saturating add, OSPF last-min-replaces) on random prefix tables against a
    per-prefix restatement over the oracle's SPT tables."""
    import torch
    dev = torch.device("cuda:0")
    ok = runs = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(130_000 + seed)
        g = synth.random_lsdb(int(rng.integers(5, 120)), int(rng.integers(0, 8)), float(rng.uniform(1.5, 4.0)), 140_000 + seed,
                              metric_hi=int(rng.integers(1, 8)), lan_size=int(rng.choice([2, 5, 12])), max_path=0xFFFFFFFF)
        n = g.n
        R = int(rng.integers(1, min(n, 70) + 1))
        roots = rng.choice(n, size=R, replace=False).astype(np.uint32)
        rflags = int(rng.choice([0, E.RUN_NET_NEXTHOPS]))
        ref = go.run(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric, roots, rflags, go.MAP)
        W = ref.mask.shape[2]
        G = ctx.upload(g.row_ptr, g.col, g.metric, g.vflags, g.max_path_metric)
        dist = torch.empty((R, n), dtype=torch.int32, device=dev); hops = torch.empty((R, n), dtype=torch.int16, device=dev)
        fl = torch.empty((R, n), dtype=torch.int16, device=dev); mask = torch.empty((R, n, W), dtype=torch.int64, device=dev)
        ctx.run_device(G, roots, rflags, dist_ptr=dist.data_ptr(), hops_ptr=hops.data_ptr(), flags_ptr=fl.data_ptr(),
                       mask_ptr=mask.data_ptr(), mask_words=W)
        G.free()
        P = int(rng.integers(1, 60)); ne = int(rng.integers(0, 200))
        pfx = np.sort(rng.integers(0, P, ne)); vtx = rng.integers(0, n, ne)
        order = np.lexsort((vtx, pfx)); pfx, vtx = pfx[order], vtx[order].astype(np.uint32)
        met = rng.integers(0, 4, ne).astype(np.uint32)
        big = rng.random(ne) < 0.08
        met[big] = (0xFFFFFFFF - rng.integers(0, 6, int(big.sum()))).astype(np.uint32)
        ptr = np.zeros(P + 1, np.uint32); np.add.at(ptr, pfx + 1, 1); ptr = np.cumsum(ptr, dtype=np.uint64).astype(np.uint32)
        for mode in (0, E.PFX_SATURATING, E.PFX_LAST_MIN, E.PFX_SATURATING | E.PFX_LAST_MIN):
            bm = torch.empty((R, P), dtype=torch.int32, device=dev); be = torch.empty((R, P), dtype=torch.int32, device=dev)
            nm = torch.empty((R, P, W), dtype=torch.int64, device=dev)
            ctx.routes_device(n, R, W, dist.data_ptr(), fl.data_ptr(), mask.data_ptr(), ptr, vtx, met, best_metric_ptr=bm.data_ptr(),
                              best_entry_ptr=be.data_ptr(), nexthop_mask_ptr=nm.data_ptr(), flags=mode)
            torch.cuda.synchronize()
            hbm = bm.cpu().numpy().view(np.uint32); hbe = be.cpu().numpy().view(np.uint32); hnm = nm.cpu().numpy().view(np.uint64)
            sat, last = bool(mode & E.PFX_SATURATING), bool(mode & E.PFX_LAST_MIN)
            good = True
            for r in range(R):
                for p in range(P):
                    best, ent, acc = 0xFFFFFFFF, 0xFFFFFFFF, np.zeros(W, np.uint64)
                    for e in range(int(ptr[p]), int(ptr[p + 1])):
                        v = int(vtx[e])
                        if not ref.flags[r, v]:
                            continue
                        m = int(ref.dist[r, v]) + int(met[e])
                        m = min(m, 0xFFFFFFFF) if sat else m & 0xFFFFFFFF
                        if ent == 0xFFFFFFFF or m < best or (last and m == best):
                            best, ent, acc = m, e, ref.mask[r, v].copy()
                        elif m == best:
                            acc |= ref.mask[r, v]
                    good = good and hbm[r, p] == best and hbe[r, p] == ent and np.array_equal(hnm[r, p], acc)
            runs += 1
            ok += good
            if not good:
                print("ROUTES MISMATCH", seed, mode, flush=True)
    if verbose:
        print(f"fuzz_routes: {ok}/{runs} tables identical over {count} graphs in {time.time() - t0:.1f} s", flush=True)
    return ok, runs


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ctx = E.SpfContext(0)
    if os.environ.get("FUZZ_MID"):                                  # only the mid-size graphs (k_xcd's range)
        ok, runs = fuzz_mid(ctx, first, int(os.environ["FUZZ_MID"]))
        sys.exit(0 if ok == runs else 1)
    if os.environ.get("FUZZ_WIDE"):                                 # only the wide-mask graphs
        ok, runs = fuzz_wide(ctx, first, int(os.environ["FUZZ_WIDE"]))
        sys.exit(0 if ok == runs else 1)
    ok, runs = fuzz(ctx, first, count)
    ok2, runs2 = fuzz_layout(ctx, first, max(count // 4, 20), spf=bool(os.environ.get("FUZZ_ARBITRARY")))
    ok3, runs3 = fuzz_routes(ctx, first, max(count // 20, 5))
    sys.exit(0 if (ok == runs and ok2 == runs2 and ok3 == runs3) else 1)


if __name__ == "__main__":
    main()
