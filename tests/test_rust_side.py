"""The Rust side of the binding (rust/): what can be checked without rustc.  SURVEY.md section 7 step 9 / north_star
"new holo-spf-hip crate": the crate, the two glue modules and the call-site patches are FILES; here: sys.rs is the
header, the wrapper and the glue only use names that exist, the patches apply to the reference tree."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
RUST = os.path.join(ROOT, "rust")
REF = "/root/reference"


def _hdr():
    import gen_rust_sys
    return gen_rust_sys.parse(open(gen_rust_sys.HEADER).read())


def test_sys_rs_is_generated_from_the_header_and_current():
    import gen_rust_sys
    assert open(gen_rust_sys.OUT).read() == gen_rust_sys.generate(), "rust/holo-spf-hip/src/sys.rs is stale: python tools/gen_rust_sys.py"


def test_sys_rs_declares_every_function_of_the_header_with_the_same_arity():
    consts, structs, opaques, funcs = _hdr()
    text = open(os.path.join(RUST, "holo-spf-hip", "src", "sys.rs")).read()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "holo_spf_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(hspf_[a-z0-9_]+)\s*\(", src))
    declared -= set(re.findall(r"static inline [^;{(]*?\b(hspf_[a-z0-9_]+)\s*\(", src))     # the header's own decode helpers: nothing to bind
    bound = dict((m.group(1), m.group(2)) for m in re.finditer(r"pub fn (hspf_[a-z0-9_]+)\(([^)]*)\)", text))
    assert set(bound) == declared
    for ret, name, params in funcs:
        n = len([p for p in bound[name].split(",") if p.strip()])
        assert n == len(params), name
    assert len(funcs) == len(declared)


def test_repr_c_structs_have_the_header_field_order():
    consts, structs, opaques, funcs = _hdr()
    text = open(os.path.join(RUST, "holo-spf-hip", "src", "sys.rs")).read()
    for name, fields in structs:
        m = re.search(r"#\[repr\(C\)\]\n#\[derive\(Clone, Copy\)\]\npub struct %s \{(.*?)\n\}" % name, text, flags=re.S)
        assert m, name
        got = re.findall(r"pub ([a-z_0-9]+):", m.group(1))
        assert got == [f[1] for f in fields], name
    for name in opaques:
        assert f"pub struct {name} {{ _private: [u8; 0] }}" in text
    for name, val, unsigned in consts:
        assert re.search(r"pub const %s: (u32|i32|usize) = %s;" % (name, re.escape(val)), text), name


@pytest.mark.parametrize("rel", ["holo-spf-hip/src/lib.rs", "holo-isis/src/spf/hip.rs", "holo-ospf/src/spf/hip.rs"])
def test_wrapper_and_glue_only_use_sys_names_that_exist(rel):
    sys_text = open(os.path.join(RUST, "holo-spf-hip", "src", "sys.rs")).read()
    have = set(re.findall(r"pub (?:fn|const|struct) ([A-Za-z_0-9]+)", sys_text))
    text = open(os.path.join(RUST, rel)).read()
    used = set(re.findall(r"\bsys::([A-Za-z_0-9]+)", text))
    assert used and used <= have, sorted(used - have)
    # balanced delimiters: the cheapest "does it parse" there is without a compiler
    code = re.sub(r"//[^\n]*", "", text)
    code = re.sub(r'"(?:\\.|[^"\\])*"', '""', code)
    code = re.sub(r"'(?:\\.|[^'\\])'", "' '", code)
    for a, b in ("()", "[]", "{}"):
        assert code.count(a) == code.count(b), (rel, a, code.count(a), code.count(b))


def test_glue_uses_only_wrapper_items_that_exist():
    lib = open(os.path.join(RUST, "holo-spf-hip", "src", "lib.rs")).read()
    pub = set(re.findall(r"pub (?:fn|struct|mod) ([A-Za-z_0-9]+)", lib))
    for rel in ("holo-isis/src/spf/hip.rs", "holo-ospf/src/spf/hip.rs"):
        text = open(os.path.join(RUST, rel)).read()
        imported = re.search(r"use holo_spf_hip::\{([^}]*)\}", text).group(1)
        for name in [x.strip() for x in imported.split(",")]:
            assert name in pub or name == "sys", (rel, name)
        for meth in ("recommend_cpu", "get_or_patch", "slot_table", "run", "from_env"):
            assert meth in pub
        for meth in re.findall(r"\beng\.([a-z_]+)\(", text) + re.findall(r"\bgraph\.([a-z_]+)\(", text) + re.findall(r"\bcache\.([a-z_]+)\(", text):
            assert meth in pub, (rel, meth)


def test_every_function_of_the_header_has_a_safe_wrapper_or_is_on_the_inspection_list():
    """33 of the 62 ABI functions (every hspf_multi_*, hspf_rib_fold_device) had no Rust wrapper.  Now every
    function of include/holo_spf_hip.h is called by the wrapper crate or the glue — or is one of the inspection / plumbing
    functions below, which a holo build has no use for."""
    consts, structs, opaques, funcs = _hdr()
    text = "".join(open(os.path.join(RUST, rel)).read() for rel in ("holo-spf-hip/src/lib.rs", "holo-isis/src/spf/hip.rs", "holo-ospf/src/spf/hip.rs"))
    used = set(re.findall(r"\bsys::(hspf_[a-z0-9_]+)", text))
    unbound = sorted(name for _, name, _ in funcs if name not in used)
    assert unbound == ["hspf_get_stream", "hspf_graph_n_edges", "hspf_graph_n_edges_kept", "hspf_graph_n_vertices", "hspf_set_stream"], unbound


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "holo-isis")), reason="reference tree not mounted")
def test_patches_are_current_and_apply_to_the_reference_tree(tmp_path):
    import make_rust_patches
    old = {p: open(os.path.join(RUST, "patches", p)).read() for p in os.listdir(os.path.join(RUST, "patches"))}
    make_rust_patches.OUT = str(tmp_path)
    sys.argv = ["make_rust_patches.py", REF]
    make_rust_patches.main()
    for p, text in old.items():
        assert open(os.path.join(str(tmp_path), p)).read() == text, f"rust/patches/{p} is stale"
        r = subprocess.run(["patch", "--dry-run", "-p1", "-d", REF, "-i", os.path.join(RUST, "patches", p)],
                           capture_output=True, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "holo-isis")), reason="reference tree not mounted")
def test_glue_calls_reference_items_that_exist_with_the_reference_arity():
    """The parent-module items the glue relies on: still there, still taking what the glue passes."""
    isis = open(os.path.join(REF, "holo-isis", "src", "spf.rs")).read()
    for fn, nargs in (("vertex_edges", 6), ("zeroth_lsp", 3), ("resolve_nexthop", 8)):
        m = re.search(r"fn %s(?:<[^>]*>)?\(\s*(.*?)\)\s*(?:->|\{)" % fn, isis, flags=re.S)
        assert m, fn
        assert len([a for a in m.group(1).split(",") if a.strip()]) == nargs, fn
    glue = open(os.path.join(RUST, "holo-isis", "src", "spf", "hip.rs")).read()
    assert glue.count("resolve_nexthop(&mut nexthop, level, mt_id, &parent, &link, &mut used_adjs, interfaces, adjacencies)") == 1
    assert "vertex_edges(vid, mt_id, metric_mode, metric_type, lsdb, lsp_entries)" in glue
    # round 5: the routes / wire step from device tables (hip::update_rib) and the MANET ancestor sets
    m = re.search(r"fn vertex_networks<'a>\(\s*(.*?)\)\s*->", isis, flags=re.S)
    assert m and len([a for a in m.group(1).split(",") if a.strip()]) == 11
    assert glue.count("vertex_networks(") == 1 and "Route::new(&vertex, network, level)" in glue
    route = open(os.path.join(REF, "holo-isis", "src", "route.rs")).read()
    for name in ("pub(crate) fn new(\n        vertex: &Vertex,\n        vertex_network: &VertexNetwork,\n        level: LevelNumber,", "const INSTALLED", "const CONNECTED",
                 "pub(crate) const fn distance(&self, config: &InstanceCfg)", "pub struct Nexthop {"):
        assert name in route, name
    tx = open(os.path.join(REF, "holo-isis", "src", "ibus", "tx.rs")).read()
    m = re.search(r"pub\(crate\) fn route_install\(\s*(.*?)\)\s*\{", tx, flags=re.S)
    assert m and len([a for a in m.group(1).split(",") if a.strip()]) == 6
    assert "ibus::tx::route_install(&instance.tx.ibus, &prefix, &route, old_sr_label, distance, interfaces)" in glue
    m = re.search(r"pub\(crate\) fn route_uninstall\(\s*(.*?)\)\s*\{", tx, flags=re.S)
    assert m and len([a for a in m.group(1).split(",") if a.strip()]) == 3
    inst = open(os.path.join(REF, "holo-isis", "src", "instance.rs")).read()
    assert "pub(crate) fn rib_mut(" in inst and "pub(crate) fn is_l2_attached_to_backbone(" in inst
    manet = open(os.path.join(REF, "holo-isis", "src", "flooding", "manet.rs")).read()
    assert manet.count("spt_hopcount.is_on_path(") + manet.count(".spt_hopcount\n                .is_on_path(") == 3     # the three queries the patch redirects
    ospf = open(os.path.join(REF, "holo-ospf", "src", "spf.rs")).read()
    for name in ("fn vertex_lsa_find(", "fn vertex_lsa_links<'a>(", "fn calc_nexthops(", "RouteRtr::new(", "is_vlink_endpoint()", "spf_run_count += 1"):
        assert name in ospf, name
    g2 = open(os.path.join(RUST, "holo-ospf", "src", "spf", "hip.rs")).read()
    assert "V::calc_nexthops(" in g2 and "V::vertex_lsa_links(" in g2 and "V::vertex_lsa_find(" in g2
