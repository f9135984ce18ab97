"""The committed golden vectors (tests/golden/) are exactly what tools/make_golden.py + tools/make_golden_ospf.py +
tools/make_golden_wire.py extract
from the reference's conformance fixtures: re-run the extractors into a scratch directory and compare byte for byte.
Only where the reference is mounted (/root/reference: the build container); skipped on the GPU box."""
import filecmp
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (build container only)")
def test_committed_vectors_are_reproducible_from_the_reference(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    import make_golden_ospf
    import make_golden_wire
    old = (make_golden.OUT, make_golden_ospf.OUT, make_golden_wire.OUT)
    make_golden.OUT = make_golden_ospf.OUT = make_golden_wire.OUT = str(tmp_path)
    try:
        make_golden.make_isis()
        make_golden.make_isis_steps()
        make_golden_ospf.make_ospfv2()
        make_golden_ospf.make_ospfv2_steps()
        make_golden_ospf.make_ospfv3()
        make_golden_wire.make_wire()          # the cold-start wire states: every topology router's output/ibus.jsonl
    finally:
        make_golden.OUT, make_golden_ospf.OUT, make_golden_wire.OUT = old
    total = 0
    for sub in ("isis", "isis_steps", "ospfv2", "ospfv2_steps", "ospfv3", "wire/isis", "wire/ospfv2", "wire/ospfv3"):
        committed = os.path.join(ROOT, "tests", "golden", sub)
        fresh = os.path.join(str(tmp_path), sub)
        names = sorted(os.listdir(committed))
        assert names == sorted(os.listdir(fresh)), sub
        match, mismatch, errors = filecmp.cmpfiles(committed, fresh, names, shallow=False)
        assert not mismatch and not errors, (sub, mismatch[:3], errors[:3])
        total += len(match)
    assert total == 38 + 19 + 63 + 11 + 44 + (38 + 50 + 44)
