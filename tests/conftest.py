import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def _ctx_pool():
    pool = {}
    yield pool
    for ctx in pool.values():
        ctx.close()


@pytest.fixture
def spf_ctx(request, _ctx_pool):
    """Engine contexts shared by the GPU tests of a session.  "default" is the product configuration (small graphs
    take the one-workgroup-per-root kernel, k_single); "sweeps" has that kernel switched off (HSPF_SINGLE_MAX_N=0), so
    that the batched sweep engine keeps its coverage on the small adversarial graphs of the suite.  Tests choose with
    tests/_engines.py: `both_engines` / `sweeps_engine` (indirect parametrisation); unmarked tests get "default"."""
    mode = getattr(request, "param", "default")
    if mode not in _ctx_pool:
        from holo_amd.engine import SpfContext
        old = os.environ.get("HSPF_SINGLE_MAX_N")
        if mode == "sweeps":
            os.environ["HSPF_SINGLE_MAX_N"] = "0"
        try:
            _ctx_pool[mode] = SpfContext(0)          # the switch is read once, at hspf_init
        finally:
            if mode == "sweeps":
                if old is None:
                    del os.environ["HSPF_SINGLE_MAX_N"]
                else:
                    os.environ["HSPF_SINGLE_MAX_N"] = old
    return _ctx_pool[mode]
