#!/usr/bin/env python3
"""tools/make_golden.py — extract known-answer vectors for the SPF path from the reference's own
conformance fixtures (runs only in the build container, where /root/reference exists).

The reference has no unit test of SPF; what pins `compute_spt` / `run_area` are the recorded
topology fixtures: each `<topo>/<rt>/output/northbound-state.json`
holds the decoded LSDB, the adjacency tables *and* the resulting `local-rib`, i.e. input and
answer of the SPF path in one file.  This script copies only the fields the path reads into
small JSON vectors under tests/golden/{isis,ospfv2}/ so the tests can run where /root/reference
does not exist (the GPU box).  No reference source code is copied, only test data.

    python tools/make_golden.py            # regenerate everything
"""
from __future__ import annotations

import glob
import json
import os
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def _proto(doc, key):
    for p in doc["ietf-routing:routing"]["control-plane-protocols"]["control-plane-protocol"]:
        if key in p:
            return p[key]
    raise KeyError(key)


# --------------------------------------------------------------------------------------------
# IS-IS   (fixture layout: holo-isis/tests/conformance/topologies/<topo>/<rt>/)
# --------------------------------------------------------------------------------------------

def _isis_metric_type(cfg, level):
    """`metric-type { value; level-1 { value }; level-2 { value } }` (ietf-isis.yang), resolved as
    holo-isis/src/northbound/configuration.rs LevelsCfg::get does: level override, else global,
    else the YANG default wide-only."""
    mt = cfg.get("metric-type", {})
    v = mt.get(f"level-{level}", {}).get("value", mt.get("value", "wide-only"))
    return {"wide-only": "wide", "old-only": "standard", "both": "both"}[v]


def _isis_iface_metric(icfg, level):
    m = icfg.get("metric", {})
    return int(m.get(f"level-{level}", {}).get("value", m.get("value", 10)))


def _reach_v4_old(tlv):
    return [[f'{p["ip-prefix"]}/{p["prefix-len"]}', int(p["default-metric"]["metric"])]
            for p in (tlv or {}).get("prefixes", [])]


def isis_vector(rt_dir: str) -> dict:
    cfg_doc = json.load(open(os.path.join(rt_dir, "config.json")))
    st_doc = json.load(open(os.path.join(rt_dir, "output", "northbound-state.json")))
    cfg = _proto(cfg_doc, "ietf-isis:isis")
    st = _proto(st_doc, "ietf-isis:isis")

    afs = {"ipv4": True, "ipv6": True}
    for a in cfg.get("address-families", {}).get("address-family-list", []):
        afs[a["address-family"]] = bool(a.get("enabled", True))
    mt_ipv6 = False
    for t in cfg.get("topologies", {}).get("topology", []):
        if t["name"].endswith("ipv6-unicast"):
            mt_ipv6 = bool(t.get("enabled", True))
    config = {
        "system_id": cfg["system-id"],
        "level_type": cfg.get("level-type", "level-all"),
        "metric_type": {"1": _isis_metric_type(cfg, 1), "2": _isis_metric_type(cfg, 2)},
        "afs": afs,
        "mt_ipv6_unicast": mt_ipv6,
        "max_paths": int(cfg.get("spf-control", {}).get("paths", 16)),
        "att_ignore": bool((cfg.get("attached-bit") or cfg.get("holo-isis:attached-bit") or {}).get("ignore-reception", False)),
        "area_addrs": cfg.get("area-address", []),
    }

    icfgs = {i["name"]: i for i in cfg.get("interfaces", {}).get("interface", [])}
    ifaces = []
    for i in st.get("interfaces", {}).get("interface", []):
        ic = icfgs.get(i["name"], {})
        adjs = []
        for a in i.get("adjacencies", {}).get("adjacency", []):
            adjs.append({
                "system_id": a["neighbor-sysid"],
                "usage": a["usage"],
                "state": a["state"],
                "ipv4": a.get("holo-isis:ipv4-addresses", []),
                "ipv6": a.get("holo-isis:ipv6-addresses", []),
                "topologies": a.get("holo-isis:topologies", []),
                "area_addrs": a.get("holo-isis:area-addresses", []),
            })
        ifaces.append({
            "name": i["name"],
            "type": ic.get("interface-type", "broadcast"),
            "metric": {"1": _isis_iface_metric(ic, 1), "2": _isis_iface_metric(ic, 2)},
            "adjacencies": adjs,
        })

    lsdb = {}
    for lv in st.get("database", {}).get("levels", []):
        lsps = []
        for l in lv.get("lsp", []):
            flags = []
            for f in l.get("attributes", {}).get("lsp-flags", []):
                if f == "lsp-overload-flag":
                    flags.append("ol")
                if f == "lsp-attached-default-metric-flag":
                    flags.append("att")
            mts = []
            for t in (l.get("mt-entries") or {}).get("topology", []):
                tf = []
                for f in t.get("attributes", {}).get("flags", []):
                    if "overload" in f:
                        tf.append("ol")
                    if "attached" in f:
                        tf.append("att")
                mts.append({"id": int(t["mt-id"]), "flags": tf})

            def nbrs(tlv, metric_of):
                out = []
                for nb in (tlv or {}).get("neighbor", []):
                    for inst in nb["instances"]["instance"]:
                        e = [nb["neighbor-id"], int(metric_of(inst))]
                        if "mt-id" in nb:
                            e = [int(nb["mt-id"])] + e
                        out.append(e)
                return out

            def reach_wide(tlv, has_mt=False):
                out = []
                for p in (tlv or {}).get("prefixes", []):
                    e = [f'{p["ip-prefix"]}/{p["prefix-len"]}', int(p["metric"]),
                         bool(p.get("external-prefix-flag", False))]
                    if has_mt:
                        e = [int(p["mt-id"])] + e
                    out.append(e)
                return out

            extra = {}
            if l.get("remaining-lifetime", 1) == 0:       # expired / purged, still in the LSDB (spf.rs:1025)
                extra["lifetime"] = 0
            lsps.append({
                **extra,
                "id": l["lsp-id"],
                "flags": flags,
                "protocols": l.get("protocol-supported"),
                "mt": mts,
                "is_reach": nbrs(l.get("is-neighbor"), lambda i: i["default-metric"]["metric"]),
                "ext_is_reach": nbrs(l.get("extended-is-neighbor"), lambda i: i["metric"]),
                "mt_is_reach": nbrs(l.get("mt-is-neighbor"), lambda i: i["metric"]),
                "ipv4_int": _reach_v4_old(l.get("ipv4-internal-reachability")),
                "ipv4_ext": _reach_v4_old(l.get("ipv4-external-reachability")),
                "ext_ipv4": reach_wide(l.get("extended-ipv4-reachability")),
                "ipv6": reach_wide(l.get("ipv6-reachability")),
                "mt_ipv6": reach_wide(l.get("mt-ipv6-reachability"), has_mt=True),
            })
        lsdb[str(lv["level"])] = lsps

    rib = _rib_rows(st)
    return {"source": os.path.relpath(rt_dir, REF), "proto": "isis", "config": config,
            "interfaces": ifaces, "lsdb": lsdb, "rib": rib}


def _rib_rows(st):
    rib = []
    for r in st.get("local-rib", {}).get("route", []):
        nhs = [[n.get("next-hop"), n.get("outgoing-interface")]
               for n in r.get("next-hops", {}).get("next-hop", [])]
        rib.append({"prefix": r["prefix"], "metric": int(r["metric"]), "level": int(r["level"]),
                    "nexthops": nhs})
    return rib


def _ibus_routes(path):
    """RouteIpAdd / RouteIpDel messages of one recorded ibus output, in order."""
    out = []
    for line in open(path):
        m = json.loads(line)
        if "RouteIpAdd" in m:
            a = m["RouteIpAdd"]
            out.append({"op": "add", "prefix": a["prefix"], "metric": a["metric"], "distance": a["distance"],
                        "nexthops": [[n["Address"]["ifindex"], n["Address"]["addr"]] for n in a["nexthops"] if "Address" in n]})
        elif "RouteIpDel" in m:
            out.append({"op": "del", "prefix": m["RouteIpDel"]["prefix"]})
    return out


def make_isis():
    base = os.path.join(REF, "holo-isis/tests/conformance/topologies")
    out = os.path.join(OUT, "isis")
    os.makedirs(out, exist_ok=True)
    n = 0
    for rt in sorted(glob.glob(os.path.join(base, "topo*", "rt*"))):
        v = isis_vector(rt)
        name = f"{os.path.basename(os.path.dirname(rt))}_{os.path.basename(rt)}.json"
        json.dump(v, open(os.path.join(out, name), "w"), separators=(",", ":"), sort_keys=True)
        n += 1
    print(f"isis: {n} vectors -> {out}")


# --------------------------------------------------------------------------------------------
# IS-IS step tests (holo-isis/tests/conformance/<name>/NN-{input,output}-*): a topology snapshot,
# then inputs (PDUs, config changes, interface events).  The LAST recorded state is again LSDB +
# adjacencies + local-rib, i.e. a known-answer vector — the ones whose last step ran SPF pin the
# overload / ATT / max-paths / metric gates with the reference's own answers.
# --------------------------------------------------------------------------------------------

def _merge_cfg(base, patch):
    for k, v in patch.items():
        if k.startswith("@"):
            continue
        kk = k.split(":", 1)[1] if (":" in k and k not in base and k.split(":", 1)[1] in base) else k
        if isinstance(v, dict):
            base[kk] = _merge_cfg(base.get(kk, {}) if isinstance(base.get(kk), dict) else {}, v)
        elif isinstance(v, list) and v and isinstance(v[0], dict) and "name" in v[0]:
            cur = {e["name"]: e for e in base.get(kk, [])}
            for e in v:
                cur[e["name"]] = _merge_cfg(cur.get(e["name"], {}), e)
            base[kk] = list(cur.values())
        else:
            base[kk] = v
    return base


def make_isis_steps():
    import re
    import tempfile
    import shutil
    base = os.path.join(REF, "holo-isis/tests/conformance")
    src = open(os.path.join(base, "mod.rs")).read()
    out = os.path.join(OUT, "isis_steps")
    os.makedirs(out, exist_ok=True)
    n = 0
    for name, topo, rt in re.findall(r'run_test::<Instance>\(\s*"([^"]+)",\s*"([^"]+)",\s*"([^"]+)"', src):
        d = os.path.join(base, name)
        states = sorted(glob.glob(os.path.join(d, "*-output-northbound-state.json")))
        if not states:
            continue
        # Only steps after which the reference demonstrably re-ran SPF: the same step also recorded
        # route (re)installations on the ibus.  (Other final states still show the RIB of the
        # previous SPF run: the delay timer had not fired yet when the step was recorded.)
        ibus = states[-1].replace("-output-northbound-state.json", "-output-ibus.jsonl")
        if not os.path.exists(ibus) or "RouteIp" not in open(ibus).read():
            continue
        cfg = json.load(open(os.path.join(base, "topologies", topo, rt, "config.json")))
        for ch in sorted(glob.glob(os.path.join(d, "*-input-northbound-config-change.json"))):
            _merge_cfg(cfg, json.load(open(ch)))
        tmp = tempfile.mkdtemp()
        try:
            os.makedirs(os.path.join(tmp, "output"))
            json.dump(cfg, open(os.path.join(tmp, "config.json"), "w"))
            shutil.copy(states[-1], os.path.join(tmp, "output", "northbound-state.json"))
            try:
                v = isis_vector(tmp)
            except Exception as e:      # noqa: BLE001
                print("skip", name, repr(e)[:80])
                continue
        finally:
            shutil.rmtree(tmp)
        v["source"] = f"holo-isis/tests/conformance/{name} (snapshot {topo}/{rt}, state {os.path.basename(states[-1])})"
        # The wire step after the path: the route messages this step put on the ibus
        # (update_global_rib, holo-isis/src/route.rs:254-312 -> ibus::tx::route_install / route_uninstall), the local RIB
        # BEFORE the step (the previous recorded state of the test, else the topology snapshot) and the interface
        # indices the harness had announced (InterfaceUpd events of the snapshot and of the test's own ibus inputs).
        prev = states[-2] if len(states) > 1 else os.path.join(base, "topologies", topo, rt, "output", "northbound-state.json")
        v["rib_before"] = _rib_rows(_proto(json.load(open(prev)), "ietf-isis:isis"))
        v["ibus_routes"] = _ibus_routes(ibus)
        ifx = {}
        for evf in [os.path.join(base, "topologies", topo, rt, "events.jsonl")] + sorted(glob.glob(os.path.join(d, "*-input-ibus.jsonl"))):
            for line in open(evf):
                ev = json.loads(line)
                ev = ev.get("Ibus", ev)
                if isinstance(ev, dict) and "InterfaceUpd" in ev:
                    ifx[ev["InterfaceUpd"]["ifname"]] = ev["InterfaceUpd"]["ifindex"]
        v["ifindex"] = ifx
        json.dump(v, open(os.path.join(out, f"{name}.json"), "w"), separators=(",", ":"), sort_keys=True)
        n += 1
    print(f"isis step tests: {n} vectors -> {out}")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    make_isis()
    make_isis_steps()
    try:
        from make_golden_ospf import make_ospfv2, make_ospfv2_steps, make_ospfv3   # noqa
        make_ospfv2()
        make_ospfv2_steps()
        make_ospfv3()
    except ImportError:
        pass

