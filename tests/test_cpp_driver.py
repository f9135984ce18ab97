"""The C ABI used from compiled code (tests/cpp/capi_parity.cpp through include/holo_spf_hip.hpp): no Python,
no torch between the caller and libholo_spf_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "capi_parity")


def _build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(EXE + ".cpp"):
        from holo_amd import build as hb
        hb.build_lib()
        subprocess.check_call([hb.hipcc_path(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
                               EXE + ".cpp", "-L" + os.path.join(ROOT, "holo_amd"), "-lholo_spf_hip",
                               "-Wl,-rpath,$ORIGIN/../../holo_amd", "-ldl", "-o", EXE])


def test_cpp_driver_builds_and_reports_missing_device_as_a_code():
    """CPU container: the driver must build against the header and, without a GPU, see HSPF_E_NODEV from
    hspf_init (exit 77) — an error code, not a crash, and no CPU fallback."""
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    assert subprocess.run([EXE, ROOT]).returncode == 77


@pytest.mark.gpu
def test_cpp_driver_parity_on_gpu():
    from oracle import graph_oracle
    graph_oracle.build()
    _build()
    r = subprocess.run([EXE, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit-exact" in r.stdout
