"""flooding::manet on the host side (holo_amd.isis: manet_init_cache -> reflood_list -> should_flood), CPU: the
engine is the oracle adapter, the answers are (1) the reference's own known-answer vectors for the flood-reduction hash
(holo-isis/src/flooding/manet.rs:205-232) and (2) a literal restatement of init_cache + reflood_list
(oracle/isis_ref.py) — the reference has no conformance test with Modified MANET enabled, so (2) is
"parity unpinned" beyond the hash."""
import glob
import json
import os

import pytest

from holo_amd import isis as H
from oracle import isis_ref as R
from _oracle_engine import OracleEngine

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ISIS = sorted(glob.glob(os.path.join(GOLD, "isis", "*.json")))

# (LSP id bytes, hash) — the reference's unit test, from draft-ietf-lsr-distoptflood-12 section 1.2.3
HASH_KAT = [
    ([0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x00, 0x00], 0x6215),
    ([0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x00, 0x07], 0x6215),
    ([0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x00, 0x0F], 0x6316),
    ([0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x00, 0x01], 0x410F),
]


@pytest.mark.parametrize("raw,want", HASH_KAT)
def test_flood_reduction_hash_known_answers(raw, want):
    lsp_id = (bytes(raw[:6]), raw[6], raw[7])
    assert H.flood_reduction_hash(lsp_id) == want
    assert R.flood_reduction_hash(lsp_id) == want


ALGOS = {
    "zero-pruner": None,
    "modified-manet": lambda sid: "modified-manet",
    "mixed": lambda sid: "modified-manet" if sid[-1] & 1 else "zero-pruner",
}


def check_reflood_lists(vec, engine):
    inst = H.Instance.from_vector(vec)
    local = inst.config.system_id
    n_lists = 0
    for level in inst.config.levels():
        if level not in inst.lsdb:
            continue
        systems = sorted({l.system_id for l in inst.lsdb[level].iter()})
        lsp_ids = [(s, 0, 0) for s in systems] + [(systems[0], 3, 9), (systems[-1], 0, 17)]
        for name, algo_of in ALGOS.items():
            cache = H.manet_init_cache(level, inst, engine, algo_of)
            for tn in cache:
                for lsp_id in lsp_ids:
                    got = H.reflood_list(cache, local, tn, lsp_id)
                    want = R.reflood_list(vec, level, local, tn, lsp_id, algo_of)
                    assert got == want, (level, name, tn.hex(), lsp_id)
                    for iface in inst.interfaces:
                        ups = {a.system_id for a in iface.adjacencies if a.state == "up"}
                        assert H.should_flood(iface, got) == bool(ups & set(want))
                    n_lists += 1
    return n_lists


@pytest.mark.parametrize("path", ISIS[::2], ids=[os.path.basename(p)[:-5] for p in ISIS[::2]])
def test_reflood_list_matches_literal_restatement(path):
    assert check_reflood_lists(json.load(open(path)), OracleEngine()) > 0


def test_reflood_list_without_cache_entry_is_empty():
    vec = json.load(open(ISIS[0]))
    inst = H.Instance.from_vector(vec)
    cache = H.manet_init_cache(inst.config.levels()[0], inst, OracleEngine())
    assert H.reflood_list(cache, inst.config.system_id, b"\xee" * 6, (b"\x01" * 6, 0, 0)) == []


@pytest.mark.parametrize("block", range(3))
def test_reflood_lists_on_random_instances(block):
    from _random_isis import make
    total = 0
    for seed in range(2000 + block * 20, 2000 + block * 20 + 20):
        total += check_reflood_lists(make(seed), OracleEngine())
    assert total > 100
