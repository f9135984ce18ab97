"""bench.py's multi-rank code at world size 2 on CPU (gloo), the engine stubbed FROM tests/ (tests/_bench_stub_engine.py):
first-contact insurance for the driver's N > 1 runs, which no session of this build could execute on hardware.  What runs
is the real file: launcher re-exec (`--gpus 2` without WORLD_SIZE), RANK / WORLD_SIZE handling, the communicator-id
broadcast, the C ABI's own hspf_shard_bounds, the in-flight step loop with `run_async` / `run_wait`, the in-place gather of
the distance table, the oracle check of the GATHERED table on rank 0, the `exchange` block and the one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env["HSPF_BENCH_STUB"] = "_bench_stub_engine"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, env.get("PYTHONPATH", "")])
    env["MASTER_PORT"] = str(port)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def _line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_two_ranks_through_the_launcher_reexec():
    """`python bench.py --gpus 2` started by hand becomes the launcher (torch.distributed.run, rendezvous on 127.0.0.1)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--min-timed-ms", "1",
                        "--no-cpu-baseline"], env=_env(29641), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 2
    assert d["verified_roots"] == 64 and d["verified_gathered_dist_roots"] == 128, "rank 0 checks the gathered table of all 2 x 64 roots"
    assert "STUB" in d["data"]
    assert "hspf_multi_run" in d["config"]["gather"]
    ex = d["exchange"]
    assert [r["rank"] for r in ex["per_rank"]] == [0, 1] and all(r["bytes_sent"] == 64 * d["config"]["n_vertices"] * 4 for r in ex["per_rank"])
    assert "cpu_baseline" not in d and "roofline" in d and d["roofline"]["frac"] > 0


def test_two_ranks_as_the_driver_launches_them_without_the_library_gather():
    """The driver's command line (`python -m torch.distributed.run ... bench.py --gpus 2 ...`), here with `--gather none`:
    each rank keeps its own rows (a one-rank engine per process)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29643",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--min-timed-ms", "1", "--no-cpu-baseline", "--gather", "none"]
    p = subprocess.run(cmd, env=_env(29643), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["verified_roots"] == 64 and d["verified_gathered_dist_roots"] == 64 and d["config"]["gather"] == "none"
    assert "exchange" not in d
