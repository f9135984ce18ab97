"""Host arithmetic of the multi-GPU entry points of the C ABI (hspf_shard_bounds, hspf_plan_areas): no GPU needed.
holo_amd.shard.shard_bounds (used by the gloo tests and bench.py) IS the C function: there is one implementation."""
import numpy as np
import pytest

from holo_amd import engine as E
from holo_amd import shard


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_shard_bounds_covers_all_roots_in_whole_batches(world):
    for n_roots in [0, 1, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1000, 10007]:
        py = shard.shard_bounds(n_roots, world)
        c = [E.shard_bounds(n_roots, world, r) for r in range(world)]
        assert c == py
        # contiguous cover, whole batches except the ragged tail, sizes differ by at most one batch
        assert c[0][0] == 0 and c[-1][1] == n_roots
        for (a0, a1), (b0, b1) in zip(c, c[1:]):
            assert a1 == b0
        for lo, hi in c:
            assert lo == hi or (lo % 64 == 0 and (hi % 64 == 0 or hi == n_roots))
        sizes = [(hi - lo + 63) // 64 for lo, hi in c]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [1, 2, 3, 8, 16])
def test_plan_areas_partitions_area_batches_contiguously(world):
    rng = np.random.default_rng(world)
    for trial in range(20):
        n_areas = int(rng.integers(1, 12))
        rpa = rng.integers(0, 3000, n_areas).astype(np.uint32)
        plan = E.plan_areas(rpa, world)
        # every root of every area exactly once, in order
        seen = {a: 0 for a in range(n_areas)}
        last = (-1, -1, -1)
        load = [0] * world
        for rank, area, b, e in plan:
            assert 0 <= rank < world and b < e <= rpa[area]
            assert b == seen[area] and b % 64 == 0 and (e % 64 == 0 or e == rpa[area])
            seen[area] = e
            assert (rank, area, b) > last
            last = (rank, area, b)
            load[rank] += (e - b + 63) // 64
        assert all(seen[a] == rpa[a] for a in range(n_areas))
        total = sum(int(r + 63) // 64 for r in rpa)
        assert sum(load) == total and max(load) - min(load) <= 1


def test_plan_areas_multi_area_config_keeps_areas_whole_when_it_can():
    """BASELINE configs[3]: 10 areas x 1000 roots on 8 GPUs = 160 batches, 20 per rank: an area (16 batches) is cut at
    most once and a rank touches at most 3 areas."""
    plan = E.plan_areas([1000] * 10, 8)
    per_rank = {}
    for rank, area, b, e in plan:
        per_rank.setdefault(rank, []).append(area)
    assert len(per_rank) == 8 and all(len(v) <= 3 for v in per_rank.values())


def test_world_8_plans_of_the_baseline_configs_are_balanced_to_one_batch():
    """The 8-GPU shapes of BASELINE.json (no 8-GPU node has been available to any session: the plan is what can be checked):
    configs[2] weak scaling = 8 x 64 roots -> one batch per rank; configs[3] = 10 areas x 1000 roots -> hspf_plan_areas;
    configs[4] = 101 roots of the fat-tree -> two batches on two ranks, six ranks idle (the roots of one job do not fill
    eight GPUs: that config shards by class of root).  Per-rank batch counts differ by at most one, every
    root is covered exactly once, and the slices a rank receives are what hspf_multi_run computes (hspf_shard_bounds)."""
    from holo_amd import synth
    # configs[2], weak scaling
    c2 = [E.shard_bounds(8 * 64, 8, r) for r in range(8)]
    assert [(hi - lo) for lo, hi in c2] == [64] * 8
    # configs[3]: the real root lists
    areas = synth.ospf_multi_area()
    rpa = [len(g.meta["roots"]) for g in areas]
    plan = E.plan_areas(rpa, 8)
    load = [0] * 8
    covered = {a: [] for a in range(len(rpa))}
    for rank, area, b, e in plan:
        load[rank] += (e - b + 63) // 64
        covered[area].append((b, e))
    assert max(load) - min(load) <= 1 and sum(load) == sum((r + 63) // 64 for r in rpa)
    for a, segs in covered.items():
        assert segs[0][0] == 0 and segs[-1][1] == rpa[a] and all(x[1] == y[0] for x, y in zip(segs, segs[1:]))
    # configs[4]: 101 roots
    gf = synth.isis_fattree(100)
    c4 = [E.shard_bounds(len(gf.meta["roots"]), 8, r) for r in range(8)]
    sizes = [(hi - lo + 63) // 64 for lo, hi in c4]
    assert sum(hi - lo for lo, hi in c4) == 101 and max(sizes) - min(sizes) <= 1 and sizes.count(1) == 2
