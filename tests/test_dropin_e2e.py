"""tests/cpp/dropin_e2e: the drop-in end to end through the compiled C++ host side.

CPU: the driver on the oracle stand-in for the engine — the RIB of a down-sized twin of the synthetic isis-100k LSDB equals
what the literal restatement oracle/isis_ref.py (pinned to the reference's recorded RIBs) computes from the SAME instance,
the incrementally patched graph equals a graph derived from scratch, and the device-routes path gives the same rows.
GPU: the same through the product engine (C ABI), packed hand-off and hspf_run's full tables.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "dropin_e2e")


def _built():
    if not os.path.exists(EXE):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.build()
    return EXE


def _run_and_check(engine_args, n, tmp_path):
    from oracle import isis_ref
    from oracle import graph_oracle
    graph_oracle.build()
    vec_p, rib_p = str(tmp_path / "inst.json"), str(tmp_path / "rib.json")
    p = subprocess.run([_built(), *engine_args, "--n", str(n), "--reps", "1", "--batch", "5", "--dump-json", vec_p, "--dump-rib", rib_p],
                       capture_output=True, text=True, timeout=600)
    if p.returncode == 77:
        pytest.skip("no HIP device")
    assert p.returncode == 0, p.stderr
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert rep["lsdb_to_csr_incremental"]["patched_graph_identical"] is True
    assert rep["device_routes_path"]["same_rib"] is True
    if "oracle" in engine_args:          # (the twin's threaded rebuild; with the product engine the field is reported, bench.py `dropin_e2e`)
        assert rep["batch"]["same_as_one_root_at_a_time"] is True
    assert rep["one_root"]["spt_vertices"] == n and rep["one_root"]["rib_routes"] == n + (n + 4) // 5
    vec = json.load(open(vec_p))
    got = [{"prefix": r["prefix"], "metric": r["metric"], "level": r["level"], "nexthops": [list(x) for x in r["nexthops"]]} for r in json.load(open(rib_p))]
    want = isis_ref.local_rib(vec)
    assert got == want
    assert sum(1 for r in want if len(r["nexthops"]) > 1) > 0, "the twin has ECMP routes"
    return rep


@pytest.mark.parametrize("n", [60, 700])
def test_cpu_engine_rib_of_the_downsized_twin_matches_the_literal_restatement(n, tmp_path):
    _run_and_check(["--engine", "oracle"], n, tmp_path)


@pytest.mark.parametrize("threads", ["1", "7"])
def test_cpu_engine_prefix_table_on_threads_gives_the_same_rows(threads):
    """PrefixTable::build walks graphs of 4 096 vertices and more in ranges of vertices on threads (HSPF_KEYED_THREADS) and
    reads the LSDB through a forward cursor: the rows attached from that table (compute_spf_device_routes, RibPipeline's
    first step and its LSP-change steps) equal what the host rule makes of compute_spf's SPT on the same instance; the Spts
    of a batch of roots, rebuilt side by side, equal the ones rebuilt one root at a time."""
    p = subprocess.run([_built(), "--engine", "oracle", "--n", "6000", "--reps", "1", "--batch", "5"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSPF_KEYED_THREADS=threads))
    assert p.returncode == 0, p.stderr
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert rep["device_routes_path"]["same_rib"] is True
    assert rep["running_instance_pipeline"]["identical_to_host_rule"] is True
    assert rep["batch"]["same_as_one_root_at_a_time"] is True          # compute_spts: the roots rebuilt on threads
    assert rep["running_instance_pipeline"]["first_step_messages"] == rep["one_root"]["rib_routes"] - 2
    assert rep["one_root"]["rib_routes"] == 7200


def test_cpu_engine_threaded_host_paths_under_thread_sanitizer(tmp_path):
    """The host side's threaded paths (PrefixTable::build's ranges of vertices, compute_spts' roots side by side) built with
    g++ -fsanitize=thread and run on 7 threads with the oracle standing in for the engine: no data race reported, same rows."""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    from oracle import graph_oracle
    from holo_amd import build as hb
    graph_oracle.build()
    hb.build_lib()
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_e2e_tsan")          # (beside dropin_e2e: it finds oracle/liboracle_spf.so relative to itself)
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-w", "-pthread",
                        "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "dropin_e2e.cpp"),
                        "-L" + os.path.join(ROOT, "holo_amd"), "-lholo_spf_hip", "-Wl,-rpath," + os.path.join(ROOT, "holo_amd"),
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("libtsan not installed")
    assert r.returncode == 0, r.stderr[-2000:]

    def _no_caps():                          # TSan maps terabytes of shadow address space: the suite's memory cap (conftest.py) off
        import resource
        for lim in (resource.RLIMIT_DATA, resource.RLIMIT_AS):
            hard = resource.getrlimit(lim)[1]
            resource.setrlimit(lim, (hard, hard))
    try:
        p = subprocess.run([exe, "--engine", "oracle", "--n", "6000", "--reps", "1", "--batch", "5"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, HSPF_KEYED_THREADS="7", TSAN_OPTIONS="halt_on_error=0"), preexec_fn=_no_caps)
    finally:
        os.remove(exe)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert rep["device_routes_path"]["same_rib"] is True and rep["batch"]["same_as_one_root_at_a_time"] is True
    assert rep["running_instance_pipeline"]["identical_to_host_rule"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("args", [[], ["--no-packed"]])
@pytest.mark.parametrize("n", [700, 3000])
def test_gpu_engine_rib_of_the_downsized_twin_matches_the_literal_restatement(n, args, tmp_path):
    rep = _run_and_check(["--engine", "hip", *args], n, tmp_path)
    assert rep["packed_handoff"] is (not args)
