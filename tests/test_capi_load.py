"""The C-ABI shared library must load without a GPU and export every symbol include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    """Every function the headers declare — without the `static inline` decode helpers of the packed results (defined in
    the header itself, nothing to export)."""
    names, inline = set(), set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(hspf_[a-z0-9_]+)\s*\(", src))
        inline |= set(re.findall(r"static inline [^;{(]*?\b(hspf_[a-z0-9_]+)\s*\(", src))
    assert inline == {"hspf_packed_word", "hspf_packed_in_spt", "hspf_packed_dist", "hspf_packed_hops", "hspf_packed_mask"}
    return names - inline


def test_library_loads_and_exports_every_declared_symbol():
    from holo_amd import build, _lib
    build.build_lib()
    lib = _lib.load()
    decl = declared_symbols()
    assert len(decl) >= 17
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert {n for n, _, _ in _lib.SYMBOLS} == decl, "ctypes binding table out of sync with the header"
    assert lib.hspf_abi_version() == 8
    assert lib.hspf_strerror(-7).decode().startswith("results do not fit")
    assert lib.hspf_strerror(-5).decode().startswith("too many")


def test_rust_sys_binds_every_declared_symbol():
    """rust/holo-spf-hip/src/sys.rs (generated from the header) has a `pub fn` for every
    function the header declares (the full comparison lives in tests/test_rust_side.py)."""
    text = open(os.path.join(ROOT, "rust", "holo-spf-hip", "src", "sys.rs")).read()
    bound = set(re.findall(r"pub fn (hspf_[a-z0-9_]+)\s*\(", text))
    assert bound == declared_symbols()


def test_recommend_cpu_rule():
    """hspf_recommend_cpu is pure arithmetic (no GPU): one root on a tiny LSDB -> CPU, the reference's 500-router
    benchmark and anything batched -> engine."""
    from holo_amd import _lib
    lib = _lib.load()
    assert lib.hspf_recommend_cpu(25, 80, 1) == 1 and lib.hspf_recommend_cpu(100, 360, 1) == 1
    assert lib.hspf_recommend_cpu(144, 528, 1) == 0 and lib.hspf_recommend_cpu(500, 1910, 1) == 0
    assert lib.hspf_recommend_cpu(100, 360, 2) == 0 and lib.hspf_recommend_cpu(50, 170, 2) == 1
    assert lib.hspf_recommend_cpu(25, 80, 8) == 0 or lib.hspf_recommend_cpu(25, 80, 8) == 1    # tiny graphs may stay on the CPU even batched
    assert lib.hspf_recommend_cpu(100000, 1000000, 1) == 0 and lib.hspf_recommend_cpu(0, 0, 1) == 1
    # dense graphs cost the CPU loop more per vertex
    assert lib.hspf_recommend_cpu(100, 360, 1) == 1 and lib.hspf_recommend_cpu(100, 8000, 1) == 0


def test_no_device_is_an_error_code_not_a_crash():
    """On the CPU-only build container hspf_init must fail with HSPF_E_NODEV (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    from holo_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.hspf_init(0, ctypes.byref(h)) == -2
    assert not h.value


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for dp, _dn, fns in os.walk(os.path.join(ROOT, "holo_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, fn
