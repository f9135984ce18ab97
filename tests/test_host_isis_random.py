"""Differential test of the IS-IS host twin on random instances (tests/_random_isis.py), CPU only: holo_amd.isis with
the oracle engine behind it against the literal restatement of compute_spt / compute_routes (oracle/isis_ref.py, itself
pinned to the reference's recorded RIBs): the local RIB, and for every system as root the whole SPT (distance, hops,
next hops with interface and addresses, parents, first / second hops in pop order, is_on_path)."""
import pytest

from holo_amd import isis as H
from oracle import isis_ref as R
from _oracle_engine import OracleEngine
from _random_isis import make
from test_host_isis import check_spts_against_ref


@pytest.mark.parametrize("block", range(8))
def test_random_instances_local_rib_and_all_roots(block):
    eng = OracleEngine()
    for seed in range(block * 25, block * 25 + 25):
        vec = make(seed)
        inst = H.Instance.from_vector(vec)
        assert H.compute_spf(inst, eng) == R.local_rib(vec), seed
        if seed % 5 == 0:
            check_spts_against_ref(vec, H.Instance.from_vector(vec), eng)
