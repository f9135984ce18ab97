"""profiles/traffic.json (the PMC traffic bench.py quotes in its roofline block) must come from the binary that is
committed: it records the git revision it was measured at, and nothing the library is built from — holo_amd/csrc and the C ABI
header include/holo_spf_hip.h (the C++ host twins under include/ are header-only callers, not part of the binary) — may have
changed since."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _git(*args):
    return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True)


@pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, ".git")), reason="not a git checkout (GPU box snapshot)")
def test_traffic_json_was_measured_on_the_committed_kernels():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    rev = t.get("git_rev")
    assert rev and rev != "unknown", "profiles/traffic.json has no git_rev: regenerate with GIT_REV=$(git rev-parse --short HEAD) bash tools/gpu_profile.sh <tag>"
    assert _git("cat-file", "-e", rev + "^{commit}").returncode == 0, rev
    assert _git("merge-base", "--is-ancestor", rev, "HEAD").returncode == 0, rev
    changed = _git("diff", "--name-only", rev, "HEAD", "--", "holo_amd/csrc", "include/holo_spf_hip.h").stdout.split()
    assert not changed, f"kernel sources changed since profiles/traffic.json was measured at {rev}: {changed}"
    ps = t["per_step"]
    assert ps["hbm_bytes"] > 0 and "k_fused_lean" in ps["kernels"] and "k_emit_fused" in ps["kernels"]
