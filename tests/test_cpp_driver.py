"""The C ABI used from compiled code (tests/cpp/capi_parity.cpp through include/holo_spf_hip.hpp): no Python,
no torch between the caller and libholo_spf_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "capi_parity")


def _build():
    import glob as _glob
    # (the headers too: a driver built against an older include/holo_spf_hip.h passes a smaller hspf_stats to the library)
    deps = [EXE + ".cpp"] + _glob.glob(os.path.join(ROOT, "include", "*.h*"))
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
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


# ---- the C++ host side (include/holo_spf_isis.hpp, include/holo_spf_ospf.hpp) ---------------------------------------
import glob      # noqa: E402

HOST = os.path.join(ROOT, "tests", "cpp", "host_parity")
VECTORS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "isis", "*.json")) +
                 glob.glob(os.path.join(ROOT, "tests", "golden", "isis_steps", "*.json")) +
                 glob.glob(os.path.join(ROOT, "tests", "golden", "ospfv2", "*.json")) +
                 glob.glob(os.path.join(ROOT, "tests", "golden", "ospfv2_steps", "*.json")) +
                 glob.glob(os.path.join(ROOT, "tests", "golden", "ospfv3", "*.json")))


def _build_host():
    deps = [HOST + ".cpp", os.path.join(ROOT, "tests", "cpp", "mini_json.hpp"), os.path.join(ROOT, "tests", "cpp", "oracle_engine.hpp")] + glob.glob(os.path.join(ROOT, "include", "*.h*"))
    if not os.path.exists(HOST) or os.path.getmtime(HOST) < max(os.path.getmtime(d) for d in deps):
        from holo_amd import build as hb
        hb.build_lib()
        subprocess.check_call([hb.hipcc_path(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
                               HOST + ".cpp", "-L" + os.path.join(ROOT, "holo_amd"), "-lholo_spf_hip",
                               "-Wl,-rpath,$ORIGIN/../../holo_amd", "-ldl", "-o", HOST])


def test_cpp_host_side_reproduces_recorded_ribs_with_the_oracle_as_engine():
    """CPU: the compiled host side (LSDB -> CSR, slot replay through resolve_nexthop / calc_nexthops, route build) with
    the CPU oracle standing in for the ENGINE only; answers = the reference's recorded local RIBs (IS-IS 38 + 19 step
    vectors, OSPFv2 57 + 11, OSPFv3 38; the 6 + 6 virtual-link endpoints are skipped)."""
    from oracle import graph_oracle
    graph_oracle.build()
    _build_host()
    r = subprocess.run([HOST, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so"),
                        "--replay-steps", os.path.join(ROOT, "tests", "golden")] + VECTORS, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "163 vectors reproduce" in r.stdout and " 0 do not" in r.stdout
    # the reference's step tests replayed as snapshot + row patches (LevelGraph / AreaGraph refresh, GraphCache)
    assert "30 step tests replayed through patched graphs" in r.stdout and ", 0 differ" in r.stdout
    assert "57 IS-IS RIBs also derived with the prefix attachment on the engine, 0 differ" in r.stdout
    # the wire step (round 5): the reference's recorded RouteIpAdd / RouteIpDel sequences from the host rule, from engine
    # tables (comparison, compaction, packing on the engine) and, where the step kept the interfaces, the running pipeline
    assert "17 recorded ibus sequences" in r.stdout and ", 0 differ; 3 also through the running-instance pipeline" in r.stdout
    assert "11 recorded OSPFv2 ibus sequences" in r.stdout and "2 of them two-area instances folded into one RIB on the engine), 0 differ" in r.stdout
    assert "38 OSPFv3 RIBs also from the ordered fold on the engine, each with two wire steps" in r.stdout and "sequences with messages) and a Full / Partial / Full sequence of SpfState, 0 differ" in r.stdout
    assert "132 recorded cold-start ibus states" in r.stdout and "(116 through the device comparison and packing" in r.stdout and "records), 0 differ" in r.stdout   # topology output/ibus.jsonl, OSPFv3 included


def test_cpp_host_side_without_a_device_reports_it():
    import torch
    _build_host()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    assert subprocess.run([HOST, "--engine", "hip"] + VECTORS[:1]).returncode == 77      # no CPU fallback


@pytest.mark.gpu
def test_cpp_host_side_reproduces_recorded_ribs_on_gpu():
    _build_host()
    r = subprocess.run([HOST, "--engine", "hip", "--replay-steps", os.path.join(ROOT, "tests", "golden")] + VECTORS,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "163 vectors reproduce" in r.stdout and " 0 do not" in r.stdout
    assert "30 step tests replayed through patched graphs" in r.stdout and ", 0 differ" in r.stdout      # hspf_graph_patch
    assert "57 IS-IS RIBs also derived with the prefix attachment on the engine, 0 differ" in r.stdout     # hspf_routes_device
    assert "17 recorded ibus sequences" in r.stdout and ", 0 differ; 3 also through the running-instance pipeline" in r.stdout   # hspf_routes_diff_device + hspf_routes_pack
    assert "132 recorded cold-start ibus states" in r.stdout and "(116 through the device comparison and packing" in r.stdout and "records), 0 differ" in r.stdout   # topology output/ibus.jsonl, OSPFv3 included
    assert "11 recorded OSPFv2 ibus sequences" in r.stdout and "2 of them two-area instances folded into one RIB on the engine), 0 differ" in r.stdout   # hspf_rib_fold_device
    assert "38 OSPFv3 RIBs also from the ordered fold on the engine, each with two wire steps" in r.stdout and "sequences with messages) and a Full / Partial / Full sequence of SpfState, 0 differ" in r.stdout


def test_cpp_host_side_on_random_instances_against_the_literal_restatements(tmp_path):
    """The compiled host side on random protocol-level inputs (tests/_random_isis.py, tests/_random_ospf.py): expected
    rows = the literal restatements' RIBs (oracle/isis_ref.py, oracle/ospf_ref.py), engine = the CPU oracle."""
    import json
    from oracle import graph_oracle, isis_ref, ospf_ref, ospfv3_ref
    from _random_isis import add_sr, make as make_isis, make_long, make_mt, make_two_level
    from _random_ospf import make as make_ospf
    from _random_ospfv3 import make as make_ospfv3
    graph_oracle.build()
    _build_host()
    files, labelled = [], 0
    for seed in range(3000, 3150):
        v = make_isis(seed)
        v["rib"] = isis_ref.local_rib(v)
        p = tmp_path / f"isis_{seed}.json"; p.write_text(json.dumps(v)); files.append(str(p))
        if seed % 3 == 0:                                      # ... and with segment routing on: input and output labels (sr.rs)
            s = add_sr(make_isis(seed), seed)
            s["rib"] = isis_ref.local_rib(s)
            labelled += sum(1 for r in s["rib"] if r.get("sr_label") is not None or any(x is not None for x in r.get("nexthop_labels", [])))
            p = tmp_path / f"isis_sr_{seed}.json"; p.write_text(json.dumps(s)); files.append(str(p))
        t = make_two_level(seed)                               # level-all: two SPTs, the L1 / L2 merge, ATT-bit default routes
        t["rib"] = isis_ref.local_rib(t)
        p = tmp_path / f"isis_l12_{seed}.json"; p.write_text(json.dumps(t)); files.append(str(p))
        t = make_mt(seed)                                      # MT IPv6-unicast: a second topology with its own links and prefixes
        t["rib"] = isis_ref.local_rib(t)
        p = tmp_path / f"isis_mt_{seed}.json"; p.write_text(json.dumps(t)); files.append(str(p))
        t = make_long(seed)                                    # a long chain: path metrics beyond MAX_PATH_METRIC_STANDARD
        t["rib"] = isis_ref.local_rib(t)
        p = tmp_path / f"isis_long_{seed}.json"; p.write_text(json.dumps(t)); files.append(str(p))
        w = make_ospf(seed)
        w["rib"] = ospf_ref.intra_area_rib(w)
        p = tmp_path / f"ospf_{seed}.json"; p.write_text(json.dumps(w)); files.append(str(p))
        x = make_ospfv3(seed)
        x["rib"] = ospfv3_ref.intra_area_rib(x)
        p = tmp_path / f"ospfv3_{seed}.json"; p.write_text(json.dumps(x)); files.append(str(p))
    # the three-router chains of tests/test_host_isis_sr.py whose labels are worked out by hand there (P / E flags, an index
    # beyond the SRGB, an absolute label, a neighbour without the I flag, an advertiser without the SPF algorithm)
    from test_host_isis_sr import chain
    for k, kw in enumerate(({}, {"flags2": ("P",)}, {"flags2": ("P", "E")}, {"flags3": ("P", "E")}, {"srgb2": ((17000, 20),)},
                            {"sid3": {"flags": ["V", "L"], "label": 5555}}, {"cap2_flags": ("V",)}, {"algos3": (1,)})):
        c = chain(**kw)
        c["rib"] = isis_ref.local_rib(c)
        p = tmp_path / f"isis_sr_chain_{k}.json"; p.write_text(json.dumps(c)); files.append(str(p))
    r = subprocess.run([HOST, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so")] + files,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "958 vectors reproduce" in r.stdout and " 0 do not" in r.stdout
    assert labelled > 100                                      # (the SR vectors do carry labels: 186 rows)


def _manet_case_files(tmp_path):
    """Vectors with "manet" cases (expected reflood lists from oracle/isis_ref.py) for the C++ driver."""
    import json
    from oracle import isis_ref
    from _random_isis import make as make_isis
    algos = {"zero-pruner": None, "modified-manet": lambda s: "modified-manet",
             "mixed": lambda s: "modified-manet" if s[-1] & 1 else "zero-pruner"}
    vecs = [json.load(open(p)) for p in VECTORS if os.sep + "isis" + os.sep in p][::3] + [make_isis(s) for s in range(4000, 4040)]
    files, n_cases = [], 0
    for i, v in enumerate(vecs):
        local = bytes.fromhex(v["config"]["system_id"].replace(".", ""))
        cases = []
        for level_s, lsps in v["lsdb"].items():
            level = int(level_s)
            systems = sorted({l["id"][:14] for l in lsps})
            nbrs = sorted({a["system_id"] for f in v["interfaces"] for a in f["adjacencies"] if a["state"] == "up"})
            for algo, fn in algos.items():
                for tn in nbrs:
                    for sysid in systems[:6]:
                        lsp = (bytes.fromhex(sysid.replace(".", "")), 0, 9)
                        want = isis_ref.reflood_list(v, level, local, bytes.fromhex(tn.replace(".", "")), lsp, fn)
                        fmt = lambda b: f"{b[:2].hex()}.{b[2:4].hex()}.{b[4:].hex()}"      # noqa: E731
                        cases.append({"level": level, "algo": algo, "tn": tn, "lsp": [sysid, 0, 9], "want": [fmt(w) for w in want]})
        v = dict(v); v["manet"] = cases
        if "rib" not in v or not v["rib"]:
            v["rib"] = isis_ref.local_rib(v)
        n_cases += len(cases)
        p = tmp_path / f"m{i}.json"; p.write_text(json.dumps(v)); files.append(str(p))
    return files, n_cases


def test_cpp_flooding_manet_reflood_lists_against_the_literal_restatement(tmp_path):
    """flooding::manet in the compiled host side (init_cache = one batched hop-count run, reflood_list, the hash pinned by
    the reference's unit-test vectors inside the driver): expected lists = oracle/isis_ref.py on recorded topologies and on
    random instances; engine = the CPU oracle."""
    from oracle import graph_oracle
    graph_oracle.build()
    _build_host()
    files, n_cases = _manet_case_files(tmp_path)
    r = subprocess.run([HOST, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so")] + files,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert f"{n_cases} reflood lists checked, 0 differ" in r.stdout and n_cases > 500


@pytest.mark.gpu
def test_cpp_flooding_manet_reflood_lists_on_gpu(tmp_path):
    """The same reflood lists with the hop-count SPTs of every neighbour computed by ONE batched run of the product engine."""
    _build_host()
    files, n_cases = _manet_case_files(tmp_path)
    r = subprocess.run([HOST, "--engine", "hip"] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert f"{n_cases} reflood lists checked, 0 differ" in r.stdout


def _random_step_files(tmp_path, seeds, two_level=False):
    """Random (snapshot, step) pairs in the layout of tests/golden (isis/<topo>_<rt>.json + a step vector whose `source` names the
    snapshot): a random instance (tests/_random_isis.py), then LSP-level changes incl. prefixes gained, lost and re-priced and
    fragments purged; `rib`, `rib_before` and `ibus_routes` from the literal restatement (oracle/isis_ref.py)."""
    import copy
    import json
    import numpy as np
    from oracle import isis_ref
    from _random_isis import make as make_isis, make_two_level
    from test_host_isis_random import mutate
    gdir = tmp_path / "golden"
    (gdir / "isis").mkdir(parents=True)
    files = []
    def change(vec, rng):
        if "1" in vec["lsdb"]:                                             # two levels: the same kinds of change on one of them
            lv = "1" if rng.random() < 0.5 else "2"
            one = dict(vec)
            one["lsdb"] = {"2": vec["lsdb"][lv]}
            one = change(one, rng)
            step = copy.deepcopy(vec)
            step["lsdb"][lv] = one["lsdb"]["2"]
            step["rib"] = isis_ref.local_rib(step)
            return step
        step = mutate(vec, rng) if rng.random() < 0.6 else copy.deepcopy(vec)
        for _ in range(int(rng.integers(1, 4))):                           # prefix changes: the pipeline's "did an LSP's prefixes change" path
            zeroth = [l for l in step["lsdb"]["2"] if l["id"].endswith("-00") and (l["ipv4_int"] or l["ext_ipv4"] or l["ipv6"])]
            if not zeroth:
                break
            l = zeroth[int(rng.integers(0, len(zeroth)))]
            what = int(rng.integers(0, 4))
            for key in ("ipv4_int", "ext_ipv4", "ipv6"):
                if not l[key]:
                    continue
                if what == 0:
                    i = int(rng.integers(0, len(l[key])))
                    l[key][i] = [l[key][i][0], int(rng.integers(0, 30))] + l[key][i][2:]
                elif what == 1 and len(l[key]) > 1:
                    l[key].pop(int(rng.integers(0, len(l[key]))))
                elif what == 2:
                    new = f"10.9.{int(rng.integers(0, 3))}.0/24" if key != "ipv6" else f"fc09:{int(rng.integers(0, 3))}::/64"
                    l[key].append([new, int(rng.integers(0, 30))] + ([False] if key != "ipv4_int" else []))
        for l in step["lsdb"]["2"]:                                        # an LSP that was purged comes back now and then
            if l.get("lifetime") == 0 and rng.random() < 0.3:
                del l["lifetime"]
        step["rib"] = isis_ref.local_rib(step)
        return step

    for seed in seeds:
        rng = np.random.default_rng(seed)
        base = make_two_level(seed) if two_level else make_isis(seed, zero=(seed % 4 == 3))
        base["rib"] = isis_ref.local_rib(base)
        (gdir / "isis" / f"rnd{seed}_rt0.json").write_text(json.dumps(base))
        step = change(base, rng)
        ifindex = {f["name"]: i + 1 for i, f in enumerate(sorted(step["interfaces"], key=lambda f: f["name"]))}
        step["rib_before"] = base["rib"]
        step["ifindex"] = ifindex
        step["ibus_routes"] = isis_ref.update_global_rib(step["rib"], base["rib"], ifindex)
        step["source"] = f"random step (snapshot rnd{seed}/rt0, seed {seed})"
        step["random"] = True
        chain, cur = [], step
        for _ in range(3):                                                 # three more events on the same running instance
            cur = change(cur, rng)
            chain.append({k: cur[k] for k in ("proto", "source", "config", "interfaces", "lsdb", "rib")})
        step["next"] = chain
        p = tmp_path / f"step{seed}.json"
        p.write_text(json.dumps(step))
        files.append(str(p))
    return str(gdir), files


def test_cpp_running_instance_pipeline_on_random_lsp_changes(tmp_path):
    """RibPipeline (resident graph, prefix table with per-vertex signatures, record expansion) beyond the 17 recorded step
    fixtures: 600 random instances, each changed at LSP level four times in a row (metrics, overload bits, neighbours dropped,
    fragments purged and back, prefixes gained / lost / re-priced) — the messages of every step of the pipeline, of the one-shot
    device form and of the host rule equal the literal restatement's update_global_rib; the graph patched from the changed
    LSPs equals a fresh one.  (Its first run found two defects of the pipeline's reset paths: no withdrawals when the root's
    LSP or the last prefix went away, and withdrawals out of RIB order after a rebuild of the prefix table.)"""
    from oracle import graph_oracle
    graph_oracle.build()
    _build_host()
    gdir, files = _random_step_files(tmp_path, range(7000, 7600))
    r = subprocess.run([HOST, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so"), "--replay-steps", gdir] + files,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    import re
    m = re.search(r"(\d+) recorded ibus sequences .* (\d+) differ; (\d+) also through the running-instance pipeline", r.stdout)
    assert m and int(m.group(1)) == 600 and int(m.group(2)) == 0 and int(m.group(3)) == 600, r.stdout
    assert "600 step tests replayed through patched graphs" in r.stdout and ", 0 differ" in r.stdout


def _random_ospf_wire_files(tmp_path, seeds, chains=False):
    """Random OSPFv2 wire steps in the schema of tests/golden/ospfv2_steps: a random instance (tests/_random_ospf.py) before and
    after LSA-level changes of the OTHER routers (link costs, links withdrawn, Router- and Network-LSAs aged out and back);
    `rib_before` / `rib` / `ibus_routes` from the literal restatement (oracle/ospf_ref.py).  Instances whose RIB holds a route
    with addressed AND unaddressed next hops at once are left out (the restatement's message rule does not model them)."""
    import copy
    import json
    import random
    from oracle import ospf_ref
    from _random_ospf import make
    files = []

    def change(v0, rng, zero):
        v1 = copy.deepcopy(v0)
        for area in v1["areas"]:
            for r in area["routers"]:
                if r["adv_rtr"] == v1["router_id"]:
                    continue
                what = rng.random()
                if what < 0.3:
                    for l in r["links"]:
                        if rng.random() < 0.5:
                            l["metric"] = rng.randint(0 if zero else 1, 12)
                elif what < 0.38:
                    r["maxage"] = not r.get("maxage", False)
                elif what < 0.5 and r["links"]:
                    r["links"].pop(rng.randrange(len(r["links"])))
            for nl in area["networks"]:
                if rng.random() < 0.15:
                    nl["maxage"] = not nl["maxage"]
        v1["rib"] = ospf_ref.intra_area_rib(v1)
        return v1

    if chains:
        (tmp_path / "golden" / "ospfv2").mkdir(parents=True)
    for seed in seeds:
        rng = random.Random(seed)
        v0 = make(seed, zero=(seed % 5 == 4))
        if seed % 3 == 0:                                      # two or three areas sharing the local router: overlapping prefixes across areas
            from test_gpu_routes import _multi_area_instance
            rid, mp, areas = _multi_area_instance(make, seed, rng, 2 + seed % 2)
            v0 = dict(v0, router_id=rid, max_paths=mp, areas=areas, source=f"random multi-area instance {seed}")
        before = ospf_ref.intra_area_rib(v0)
        v1 = change(v0, rng, seed % 5 == 4)
        if any(len({a is None for a, _ in r["nexthops"]}) > 1 for r in before + v1["rib"]):
            continue
        v1["rib_before"] = before
        v1["ifindex"] = {nm: k + 2 for k, nm in enumerate(sorted({i["name"] for a in v1["areas"] for i in a["interfaces"]}))}
        v1["ibus_routes"] = ospf_ref.update_global_rib(v1["rib"], before, v1["ifindex"])
        v1["source"] = f"random ospf wire step {seed}"
        if chains:                                             # snapshot + three more events on the same graph cache (replay_ospf_step)
            v0["rib"] = before
            (tmp_path / "golden" / "ospfv2" / f"rnd{seed}_rt0.json").write_text(json.dumps(v0))
            v1["source"] = f"random ospf step (snapshot rnd{seed}/rt0, seed {seed})"
            cur, nxt = v1, []
            for _ in range(3):
                cur = change(cur, rng, seed % 5 == 4)
                nxt.append({k: cur[k] for k in ("proto", "source", "router_id", "max_paths", "has_vlinks", "areas", "rib")})
            v1["next"] = nxt
        p = tmp_path / f"ospf_wire_{seed}.json"
        p.write_text(json.dumps(v1))
        files.append(str(p))
    return files


def test_cpp_ospf_wire_step_on_random_lsa_changes(tmp_path):
    """The OSPFv2 wire step of the compiled host side beyond the 11 recorded sequences: ~390 random instances, a third of them
    with two or three areas, before / after LSA changes — the host rule and the one-shot device form (areas folded into one RIB,
    compared with `rib_before`, packed on the engine) give the literal restatement's RouteIpAdd / RouteIpDel sequence; engine =
    the CPU oracle."""
    import re
    from oracle import graph_oracle
    graph_oracle.build()
    _build_host()
    files = _random_ospf_wire_files(tmp_path, range(9000, 9400))
    assert len(files) > 350
    r = subprocess.run([HOST, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so")] + files,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    m = re.search(r"(\d+) recorded OSPFv2 ibus sequences reproduced .*\), (\d+) differ", r.stdout)
    assert m and int(m.group(1)) == len(files) and int(m.group(2)) == 0, r.stdout
    assert f"{len(files)} vectors reproduce" in r.stdout and " 0 do not" in r.stdout
    multi = int(re.search(r"(\d+) of them two-area instances", r.stdout).group(1))
    assert multi > 100, r.stdout


def test_cpp_wire_step_and_graph_cache_on_random_two_level_lsp_changes(tmp_path):
    """level-all instances (two tables: the device forms do not apply): the host rule's messages on compute_spf's merged RIB
    against the restatement's, and both levels' graphs patched from the changed LSPs against fresh ones."""
    import re
    from oracle import graph_oracle
    graph_oracle.build()
    _build_host()
    gdir, files = _random_step_files(tmp_path, range(8000, 8200), two_level=True)
    r = subprocess.run([HOST, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so"), "--replay-steps", gdir] + files,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    m = re.search(r"(\d+) recorded ibus sequences .* (\d+) differ; (\d+) also through the running-instance pipeline", r.stdout)
    assert m and int(m.group(1)) == 200 and int(m.group(2)) == 0, r.stdout
    assert "200 step tests replayed through patched graphs" in r.stdout and ", 0 differ" in r.stdout


def test_cpp_ospf_graph_cache_on_random_chains_of_lsa_changes(tmp_path):
    """OSPFv2 area graphs kept current from the changed LSAs (GraphCache: rows patched, spliced in place) over chains of four
    events on the same cache: the RIB of every event equals the restatement's, every patched graph a fresh one."""
    import re
    from oracle import graph_oracle
    graph_oracle.build()
    _build_host()
    files = _random_ospf_wire_files(tmp_path, range(12000, 12300), chains=True)
    r = subprocess.run([HOST, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so"), "--replay-steps", str(tmp_path / "golden")] + files,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    m = re.search(r"(\d+) step tests replayed through patched graphs \((\d+) row-patch refreshes\), (\d+) differ", r.stdout)
    assert m and int(m.group(1)) == len(files) and int(m.group(3)) == 0 and int(m.group(2)) > len(files), r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("HSPF_RANDOM_CHAINS_GPU") != "1",
                    reason="opt-in (HSPF_RANDOM_CHAINS_GPU=1): written in a session without a GPU, never run on hardware yet")
def test_cpp_running_instance_pipeline_on_random_lsp_changes_on_gpu(tmp_path):
    """The same chains with the product engine behind the pipeline (hspf_routes_device / _diff_device / _pack / graph patch)."""
    _build_host()
    gdir, files = _random_step_files(tmp_path, range(7000, 7200))
    r = subprocess.run([HOST, "--engine", "hip", "--replay-steps", gdir] + files, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "200 recorded ibus sequences" in r.stdout and ", 0 differ; 200 also through the running-instance pipeline" in r.stdout


def test_cpp_host_side_under_asan_ubsan():
    """the compiled host side (include/holo_spf_{host,isis,ospf}.hpp through tests/cpp/host_parity.cpp) built
    with g++ -fsanitize=address,undefined (-fno-sanitize-recover: any report aborts) and run over every recorded fixture
    with the oracle standing in for the engine.  Leak detection is off: the HIP runtime the C ABI library pulls in keeps
    process-lifetime allocations."""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    from oracle import graph_oracle
    from holo_amd import build as hb
    graph_oracle.build()
    hb.build_lib()
    exe = HOST + "_asan"
    deps = [HOST + ".cpp", os.path.join(ROOT, "tests", "cpp", "mini_json.hpp"), os.path.join(ROOT, "tests", "cpp", "oracle_engine.hpp")] + glob.glob(os.path.join(ROOT, "include", "*.h*"))
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-D__HIP_PLATFORM_AMD__", "-w", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                            HOST + ".cpp", "-L" + os.path.join(ROOT, "holo_amd"), "-lholo_spf_hip", "-Wl,-rpath,$ORIGIN/../../holo_amd",
                            "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-o", exe], capture_output=True, text=True)
        if r.returncode != 0 and "sanitizer" in (r.stderr or "").lower() and "cannot find" in r.stderr:
            pytest.skip("libasan / libubsan not installed")
        assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")

    def _no_data_cap():                     # ASan reserves terabytes of shadow address space: the suite's memory cap (conftest.py) off
        import resource
        hard = resource.getrlimit(resource.RLIMIT_DATA)[1]
        resource.setrlimit(resource.RLIMIT_DATA, (hard, hard))
    r = subprocess.run([exe, "--engine", "oracle", "--oracle-so", os.path.join(ROOT, "oracle", "liboracle_spf.so"),
                        "--replay-steps", os.path.join(ROOT, "tests", "golden")] + VECTORS, capture_output=True, text=True, timeout=1200, env=env,
                       preexec_fn=_no_data_cap)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "163 vectors reproduce" in r.stdout and " 0 do not" in r.stdout and ", 0 differ" in r.stdout
    assert "132 recorded cold-start ibus states" in r.stdout and "(116 through the device comparison and packing" in r.stdout and "records), 0 differ" in r.stdout   # topology output/ibus.jsonl, OSPFv3 included
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
