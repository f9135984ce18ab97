"""A slice of the randomised differential run of tools/gpu_fuzz.py inside the GPU suite: 150 adversarial LSDBs x 3 runs
(LANs up to 140 members = up to 3 mask words, every flag combination, ragged root lists up to 200 roots so that the
regrouping by state class kicks in, row patches in between) — the HIP engine against the CPU oracle, bit for bit."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from _engines import all_engines  # noqa: E402

pytestmark = [pytest.mark.gpu, all_engines]


@pytest.mark.parametrize("first", [0, 5000, 9000])
def test_random_lsdbs_runs_and_patches_against_the_oracle(spf_ctx, first):
    import gpu_fuzz
    ok, runs = gpu_fuzz.fuzz(spf_ctx, first, 50, verbose=False)
    assert ok == runs and runs >= 150          # 50 graphs x 3 runs, plus the repeated runs (learned sweep schedule)


def test_random_lsdbs_with_big_lans_up_to_15_mask_words(spf_ctx):
    import gpu_fuzz
    ok, runs = gpu_fuzz.fuzz_wide(spf_ctx, 0, 30, verbose=False)
    assert ok == runs and runs == 60


def test_random_layouts_and_arbitrary_row_patches_against_the_restatement(spf_ctx):
    import gpu_fuzz
    # spf=True: after every round of arbitrary row replacements (links between any two vertices, any flags) an SPF
    # run on whatever graph that made, against the oracle
    ok, runs = gpu_fuzz.fuzz_layout(spf_ctx, 0, 40, verbose=False, spf=True)
    assert ok == runs and runs >= 240          # 40 graphs x 3 rounds x (layout + SPF), plus the in-place cost patches


def test_random_prefix_tables_on_device_against_the_restatement(spf_ctx):
    import gpu_fuzz
    ok, runs = gpu_fuzz.fuzz_routes(spf_ctx, 0, 12, verbose=False)
    assert ok == runs and runs == 48


def test_random_isis_instances_and_ospf_areas_through_the_engine(spf_ctx):
    """The host twins on random protocol-level inputs (tests/_random_isis.py, tests/_random_ospf.py) with the HIP engine
    behind them, against the literal restatements: RIBs, and whole SPTs for a sample."""
    from holo_amd import isis as H
    from oracle import isis_ref as R
    from _random_isis import make as make_isis
    from _random_ospf import make as make_ospf
    from test_host_isis import check_spts_against_ref
    from test_host_ospf_random import check as check_ospf
    for seed in range(1000, 1060):
        vec = make_isis(seed)
        assert H.compute_spf(H.Instance.from_vector(vec), spf_ctx) == R.local_rib(vec), seed
        if seed % 6 == 0:
            check_spts_against_ref(vec, H.Instance.from_vector(vec), spf_ctx)
    for seed in range(1000, 1060):
        check_ospf(make_ospf(seed), spf_ctx)


def test_random_two_level_and_multi_topology_isis_instances_through_the_engine(spf_ctx):
    """level-all instances (one graph and SPT per level, L1 over L2 in the merge) and MT IPv6-unicast instances (a second topology
    with its own links, metrics and per-topology overload bits) from tests/_random_isis.py, HIP engine behind the twin, RIBs
    against the literal restatement.  (Written in a session without a GPU; the same instances pass with the oracle engine.)"""
    from holo_amd import isis as H
    from oracle import isis_ref as R
    from _random_isis import make_long, make_mt, make_two_level
    for seed in range(2000, 2040):
        for vec in (make_two_level(seed), make_mt(seed), make_long(seed)):     # make_long: path metrics beyond the maximum (spf.rs:637-641)
            assert H.compute_spf(H.Instance.from_vector(vec), spf_ctx) == R.local_rib(vec), (seed, vec["source"])


def test_random_isis_instances_with_zero_metrics_through_the_engine(spf_ctx):
    """Round 6: a third of the link metrics at 0 — the engine resolves the dynamic pop orders in parallel (k_repair, pop ranks from
    two sorts) and the twin's slot replay / first-hop lists follow those ranks: RIBs and whole SPTs against the literal loop."""
    from holo_amd import isis as H
    from oracle import isis_ref as R
    from _random_isis import make as make_isis
    from test_host_isis import check_spts_against_ref
    exact = 0
    for seed in range(5000, 5080):
        vec = make_isis(seed, zero=True)
        assert H.compute_spf(H.Instance.from_vector(vec), spf_ctx) == R.local_rib(vec), seed
        if seed % 3 == 0:
            check_spts_against_ref(vec, H.Instance.from_vector(vec), spf_ctx)
        st = spf_ctx.stats()
        exact += st["n_exact_roots"]
    assert exact == 0, "the sequential kernel ran"
    from _random_ospf import make as make_ospf
    from test_host_ospf_random import check as check_ospf
    for seed in range(7000, 7080):
        check_ospf(make_ospf(seed, zero=True), spf_ctx)
