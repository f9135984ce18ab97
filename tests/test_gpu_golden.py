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
