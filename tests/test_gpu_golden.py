"""GPU parity against the reference's own recorded answers: holo_amd.isis on the HIP engine
(through the C ABI) must reproduce the `local-rib` of every IS-IS conformance fixture, and its
batched multi-root SPTs must equal the literal restatement of compute_spt vertex by vertex."""
import glob
import json
import os

import pytest

from holo_amd import isis as H
from oracle import isis_ref as R
from test_host_isis import check_spts_against_ref

from _engines import both_engines  # noqa: E402

pytestmark = [pytest.mark.gpu, both_engines]

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ISIS = sorted(glob.glob(os.path.join(GOLD, "isis", "*.json"))) + sorted(glob.glob(os.path.join(GOLD, "isis_steps", "*.json")))


@pytest.mark.parametrize("path", ISIS, ids=[os.path.basename(p)[:-5] for p in ISIS])
def test_isis_compute_spf_on_gpu_reproduces_reference_local_rib(spf_ctx, path):
    vec = json.load(open(path))
    inst = H.Instance.from_vector(vec)
    want = sorted(vec["rib"], key=lambda r: R._net_key(r["prefix"]))
    assert H.compute_spf(inst, spf_ctx) == want


@pytest.mark.parametrize("path", ISIS, ids=[os.path.basename(p)[:-5] for p in ISIS])
def test_isis_batched_roots_on_gpu_match_literal_restatement(spf_ctx, path):
    vec = json.load(open(path))
    check_spts_against_ref(vec, H.Instance.from_vector(vec), spf_ctx)


# ---- OSPFv2 -----------------------------------------------------------------------------------------
from test_host_ospf import OSPF, check_ospf_vector      # noqa: E402


@pytest.mark.parametrize("path", OSPF, ids=[os.path.basename(p)[:-5] for p in OSPF])
def test_ospfv2_run_area_on_gpu_reproduces_reference_intra_area_rib(spf_ctx, path):
    check_ospf_vector(json.load(open(path)), spf_ctx)


# ---- OSPFv3 -----------------------------------------------------------------------------------------
from test_host_ospf import OSPF3, check_ospfv3_vector      # noqa: E402


@pytest.mark.parametrize("path", OSPF3, ids=[os.path.basename(p)[:-5] for p in OSPF3])
def test_ospfv3_run_area_on_gpu_reproduces_reference_intra_area_rib(spf_ctx, path):
    check_ospfv3_vector(json.load(open(path)), spf_ctx)


# ---- incremental LSDB -> CSR: step tests replayed through device-resident, patched graphs ---------
from test_host_isis_incremental import STEPS as ISIS_STEP_FILES, replay_isis_step      # noqa: E402
from test_host_ospf_incremental import STEPS as OSPF_STEP_FILES, replay_ospf_step      # noqa: E402


@pytest.mark.parametrize("path", ISIS_STEP_FILES, ids=[os.path.basename(p)[:-5] for p in ISIS_STEP_FILES])
def test_isis_step_on_gpu_through_patched_graphs(spf_ctx, path):
    replay_isis_step(json.load(open(path)), spf_ctx)


@pytest.mark.parametrize("path", OSPF_STEP_FILES, ids=[os.path.basename(p)[:-5] for p in OSPF_STEP_FILES])
def test_ospfv2_step_on_gpu_through_patched_graphs(spf_ctx, path):
    replay_ospf_step(json.load(open(path)), spf_ctx)


# ---- flooding::manet: one batched hop-count run per level, reflood lists answered from it ----------------------------
from test_host_manet import check_reflood_lists      # noqa: E402

MANET_FILES = sorted(glob.glob(os.path.join(GOLD, "isis", "*.json")))[::2]


@pytest.mark.parametrize("path", MANET_FILES, ids=[os.path.basename(p)[:-5] for p in MANET_FILES])
def test_manet_reflood_lists_on_gpu_match_literal_restatement(spf_ctx, path):
    assert check_reflood_lists(json.load(open(path)), spf_ctx) > 0


# ---- the same lists with the is_on_path queries answered from device-built ancestor sets (hspf_ancestors_device) ------
from holo_amd import isis as H                       # noqa: E402
from oracle import isis_ref as R                     # noqa: E402
from test_host_manet import ALGOS                    # noqa: E402


def check_reflood_lists_device(vec, engine):
    inst = H.Instance.from_vector(vec)
    local = inst.config.system_id
    n_lists = 0
    for level in inst.config.levels():
        if level not in inst.lsdb:
            continue
        systems = sorted({l.system_id for l in inst.lsdb[level].iter()})
        lsp_ids = [(s, 0, 0) for s in systems] + [(systems[0], 3, 9), (systems[-1], 0, 17)]
        for name, algo_of in ALGOS.items():
            cache = H.manet_init_cache_device(level, inst, engine, algo_of)
            for tn in cache:
                for lsp_id in lsp_ids:
                    got = H.reflood_list_device(cache, local, tn, lsp_id)
                    want = R.reflood_list(vec, level, local, tn, lsp_id, algo_of)
                    assert got == want, (level, name, tn.hex(), lsp_id)
                    n_lists += 1
    return n_lists


@pytest.mark.parametrize("path", MANET_FILES, ids=[os.path.basename(p)[:-5] for p in MANET_FILES])
def test_manet_reflood_lists_from_device_ancestor_sets(spf_ctx, path):
    assert check_reflood_lists_device(json.load(open(path)), spf_ctx) > 0


@pytest.mark.parametrize("block", range(2))
def test_manet_reflood_lists_from_device_ancestor_sets_random_instances(spf_ctx, block):
    from _random_isis import make
    total = 0
    for seed in range(2000 + block * 15, 2000 + block * 15 + 15):
        total += check_reflood_lists_device(make(seed), spf_ctx)
    assert total > 50
