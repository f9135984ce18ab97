import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # A test that asks for an absurd amount of host memory must fail with MemoryError, not take the machine down with it
    # (round 6: an `np.arange(1 << 40)` in a test helper cost two GPU boxes).  Private writable memory of the test process
    # is capped at 48 GiB (HIP's address-space reservations are not counted: they are not writable private mappings).
    try:
        import resource
        soft, hard = resource.getrlimit(resource.RLIMIT_DATA)
        cap = 48 << 30
        if soft == resource.RLIM_INFINITY or soft > cap:
            resource.setrlimit(resource.RLIMIT_DATA, (cap, hard))
    except Exception:      # noqa: BLE001  (platforms without the limit)
        pass


@pytest.fixture(scope="session")
def _ctx_pool():
    pool = {}
    yield pool
    for ctx in pool.values():
        ctx.close()


@pytest.fixture
def spf_ctx(request, _ctx_pool):
    """Engine contexts shared by the GPU tests of a session.  "default" is the product configuration (small graphs
    take the one-workgroup-per-root kernel, k_single, runs of a few roots on larger ones the lane = vertex kernel, k_lv);
    "sweeps" has both switched off (HSPF_SINGLE_MAX_N=0, HSPF_LV_MAX_ROOTS=0), so that the batched lane = root sweep
    engine keeps its coverage on the small adversarial graphs of the suite — since round 3 that is k_fused_lean wherever
    its 4-byte state fits, so "kfused" is "sweeps" with the lean sweep off (HSPF_VARIANT bit15): k_fused on both state
    widths; "twophase" additionally sends runs with more
    than 24 first-hop slots down the older k_relax + k_dag path instead of k_fw (HSPF_VARIANT bit6); "widemask" sends EVERY
    run down k_fw, the wide-mask fixed point (HSPF_VARIANT bit0: no packed state) — with the leaves of the graph left
    to the emit wherever it has any, which the adversarial graphs do (stub LANs, one-way links); "lanevertex" sends
    every run of up to 64 roots (with at most 24 first-hop slots) through k_lv (HSPF_SINGLE_MAX_N=0, HSPF_LV_MAX_ROOTS=64,
    HSPF_LV_MIN_N=0); "xcd" (round 5) has k_single and k_lv off, so that every run of at most eight roots — on the small
    adversarial graphs too — takes k_xcd, one XCD per root with the state replicated in every CU's LDS, whatever it measured
    before (HSPF_XCD_ALWAYS=1; in "default" the graphs between k_single's and 20 000 vertices take it or the sweep engine,
    whichever was faster last time; the four sweep configurations switch it off: HSPF_XCD_MAX_ROOTS=0);
    "hubsort" is the default engine with every graph built in hub mode (HSPF_HUB_DEG=0: two-way check and
    in-row order from device-wide sorts instead of per-link row scans; HSPF_TW_HOST_MAX=0: a structural patch fetches the
    two-way flags of its host mirror from the device, as it does for rows too long to scan, instead of keeping them itself);
    "patchfull" (round 6) is the default engine with every structural patch rebuilding the layout from the spliced rows
    (HSPF_PATCH_FULL=1: the path every such patch took before the incremental one, holo_amd/csrc/graph_patch.hip.h).
    Tests choose with tests/_engines.py (indirect parametrisation); unmarked tests get "default"."""
    mode = getattr(request, "param", "default")
    if mode not in _ctx_pool:
        from holo_amd.engine import SpfContext
        env = {"sweeps": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "0", "HSPF_XCD_MAX_ROOTS": "0"},
               "kfused": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "0", "HSPF_XCD_MAX_ROOTS": "0", "HSPF_VARIANT": "32768"},
               "twophase": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "0", "HSPF_XCD_MAX_ROOTS": "0", "HSPF_VARIANT": "64"},
               "widemask": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "0", "HSPF_XCD_MAX_ROOTS": "0", "HSPF_VARIANT": "1"},
               "lanevertex": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "64", "HSPF_LV_MIN_N": "0"},
               "xcd": {"HSPF_SINGLE_MAX_N": "0", "HSPF_LV_MAX_ROOTS": "0", "HSPF_XCD_ALWAYS": "1"},
               "hubsort": {"HSPF_HUB_DEG": "0", "HSPF_TW_HOST_MAX": "0"},
               "patchfull": {"HSPF_PATCH_FULL": "1"}}.get(mode, {})
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            _ctx_pool[mode] = SpfContext(0)          # the switches are read once, at hspf_init
            _ctx_pool[mode].mode = mode
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
    return _ctx_pool[mode]
