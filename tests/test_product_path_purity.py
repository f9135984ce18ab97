"""The oracle is test infrastructure: nothing the product is built from may import, link or load it, and the product path
fails loudly — no CPU fallback — when the HIP library is not there."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nothing_under_the_package_or_the_headers_reaches_for_the_oracle():
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|liboracle|oracle/|dlopen\([^)]*oracle", re.M)
    files = (glob.glob(os.path.join(ROOT, "holo_amd", "*.py")) + glob.glob(os.path.join(ROOT, "holo_amd", "csrc", "*")) +
             glob.glob(os.path.join(ROOT, "include", "*")) + glob.glob(os.path.join(ROOT, "rust", "**", "*.rs"), recursive=True))
    assert len(files) > 20
    hits = []
    for f in files:
        if os.path.isdir(f) or f.endswith((".o", ".so")):
            continue
        for m in pat.finditer(open(f, errors="replace").read()):
            line = m.group(0).strip()
            hits.append((os.path.relpath(f, ROOT), line))
    # comments may NAME the oracle files as the place a rule is restated; code may not load them
    hits = [(f, l) for f, l in hits if not l.startswith("oracle/")]
    assert not hits, hits


def test_bench_reaches_the_oracle_only_inside_its_checks_and_the_cpu_baseline():
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    for n in top:
        names = [a.name for a in n.names] + ([n.module] if isinstance(n, ast.ImportFrom) and n.module else [])
        assert not any(x.split(".")[0] == "oracle" for x in names), ast.dump(n)


def test_the_product_path_fails_loudly_without_the_library(monkeypatch, tmp_path):
    from holo_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libholo_spf_hip.so"))
    monkeypatch.setattr(_lib, "_lib", None)                 # (not the handle an earlier test of this process loaded)
    with pytest.raises((OSError, RuntimeError, FileNotFoundError)):
        _lib.load()
