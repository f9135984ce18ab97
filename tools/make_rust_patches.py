"""rust/patches/*.patch — the edits of EXISTING holo files that bind the holo-spf-hip crate, as unified diffs against
the reference tree (holo v0.9.0), generated from the edit list below so that they are reproducible:

    python tools/make_rust_patches.py [/root/reference]

New files are not in the patches: rust/holo-spf-hip/ (the crate), rust/holo-isis/src/spf/hip.rs and
rust/holo-ospf/src/spf/hip.rs are copied into the workspace as they are (rust/README.md).  tests/test_rust_side.py
applies the patches with `patch --dry-run` where the reference tree is mounted.
"""
import difflib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust", "patches")

# (patch file, reference file, [(anchor text that must occur exactly once, replacement)])
EDITS = [
    ("workspace.patch", "Cargo.toml", [
        ('  "holo-routing",\n', '  "holo-routing",\n  "holo-spf-hip",\n'),
    ]),
    ("holo-isis.patch", "holo-isis/Cargo.toml", [
        ('holo-utils = { path = "../holo-utils" }\nholo-yang', 'holo-spf-hip = { path = "../holo-spf-hip" }\nholo-utils = { path = "../holo-utils" }\nholo-yang'),
    ]),
    ("holo-isis.patch", "holo-isis/src/spf.rs", [
        ("use crate::{flooding, route, sr, tasks};\n",
         "use crate::{flooding, route, sr, tasks};\n\n"
         "// MI355X path of compute_spt (holo-spf-hip): LSDB -> CSR, the SPT loop on the GPU, next hops through the\n"
         "// unchanged resolve_nexthop.  `None` from it = no engine / too small / engine error: the loop below runs.\n"
         "mod hip;\n"
         "pub(crate) use hip::compute_spts as compute_spts_hip;\n"
         "pub(crate) use hip::{ManetSets, manet_sets as manet_sets_hip};\n"),
        ("    let mut used_adjs = BTreeSet::new();\n\n    // Get root vertex.\n",
         "    let mut used_adjs = BTreeSet::new();\n\n"
         "    // MI355X path (HOLO_SPF_HIP_DEVICE): same SPT, computed by the engine.\n"
         "    if let Some(spt) = hip::compute_spt(\n"
         "        level,\n        root_system_id,\n        local,\n        mt_id,\n        metric_mode,\n"
         "        instance,\n        interfaces,\n        adjacencies,\n        lsp_entries,\n    ) {\n"
         "        return spt;\n    }\n\n"
         "    // Get root vertex.\n"),
        ("    let mut new_rib = BTreeMap::new();\n"
         "    for mt_id in [MtId::Standard, MtId::Ipv6Unicast] {\n"
         "        if instance.config.is_topology_enabled(mt_id) {\n"
         "            compute_routes(\n"
         "                level,\n                mt_id,\n                instance,\n                interfaces,\n"
         "                adjacencies,\n                lsp_entries,\n                &mut new_rib,\n            );\n"
         "        }\n    }\n\n"
         "    // Update the local RIB and global RIB.\n"
         "    route::update_rib(level, new_rib, instance, interfaces);\n",
         "    //\n"
         "    // MI355X path (spf/hip.rs `update_rib`): SPT, prefix attachment and the comparison with the RIB held\n"
         "    // before on the device; a Route is built only for the routes that changed, and route_install /\n"
         "    // route_uninstall are called for exactly those.  `None` = not applicable or engine error: as before.\n"
         "    let trigger_lans = trigger_lsps\n"
         "        .keys()\n"
         "        .map(|lsp_id| LanId::from((lsp_id.system_id, lsp_id.pseudonode)))\n"
         "        .collect();\n"
         "    if hip::update_rib(\n"
         "        level,\n        &trigger_lans,\n        instance,\n        interfaces,\n        adjacencies,\n        lsp_entries,\n    )\n"
         "    .is_none()\n"
         "    {\n"
         "        let mut new_rib = BTreeMap::new();\n"
         "        for mt_id in [MtId::Standard, MtId::Ipv6Unicast] {\n"
         "            if instance.config.is_topology_enabled(mt_id) {\n"
         "                compute_routes(\n"
         "                    level,\n                    mt_id,\n                    instance,\n                    interfaces,\n"
         "                    adjacencies,\n                    lsp_entries,\n                    &mut new_rib,\n                );\n"
         "            }\n        }\n\n"
         "        // Update the local RIB and global RIB.\n"
         "        route::update_rib(level, new_rib, instance, interfaces);\n"
         "    }\n"),
    ]),
    ("holo-isis.patch", "holo-isis/src/flooding/manet.rs", [
        ("    pub remote_nbr_list: BTreeMap<SystemId, FloodingAlgo>,\n}\n",
         "    pub remote_nbr_list: BTreeMap<SystemId, FloodingAlgo>,\n"
         "    // MI355X path: ancestor bit sets of the batched hop-count run and this neighbor's row in them\n"
         "    // (spf/hip.rs `ManetSets`): every is_on_path of reflood_list is one bit test.\n"
         "    pub hip_sets: Option<(std::sync::Arc<spf::ManetSets>, u32)>,\n}\n"),
        ("        let mut cache = NeighborCache::default();\n",
         "        let mut cache = NeighborCache::default();\n"
         "        cache.hip_sets = hip_sets.clone().map(|sets| (sets, hip_row));\n"
         "        hip_row += 1;\n"),
        ("    if cache.remote_nbr_list.is_empty() {\n        return BTreeSet::default();\n    }\n",
         "    if cache.remote_nbr_list.is_empty() {\n        return BTreeSet::default();\n    }\n\n"
         "    // Spt::is_on_path from the device's ancestor sets where they answer it (MI355X path), else the DFS.\n"
         "    let on_path = |ancestor: SystemId, descendant: SystemId| -> bool {\n"
         "        cache\n"
         "            .hip_sets\n"
         "            .as_ref()\n"
         "            .and_then(|(sets, row)| sets.is_on_path(*row, ancestor, descendant))\n"
         "            .unwrap_or_else(|| cache.spt_hopcount.is_on_path(ancestor, descendant))\n"
         "    };\n"),
        ("            !cache\n                .spt_hopcount\n                .is_on_path(vertex.id.lan_id.system_id, lsp_id.system_id)\n",
         "            !on_path(vertex.id.lan_id.system_id, lsp_id.system_id)\n"),
        ("                if cache.spt_hopcount.is_on_path(*rnl, thl_node) {\n",
         "                if on_path(*rnl, thl_node) {\n"),
        ("            .retain(|thl_node| !cache.spt_hopcount.is_on_path(*rnl, *thl_node));\n",
         "            .retain(|thl_node| !on_path(*rnl, *thl_node));\n"),
        ("    // Process all adjacencies on active interfaces.\n    for adj in interfaces\n",
         "    // MI355X path: the hop-count SPTs of ALL Up adjacencies in ONE engine run (holo-spf-hip); when that is not\n"
         "    // available each neighbor is computed by spf::compute_spt below, as before.\n"
         "    let nbrs: Vec<SystemId> = interfaces\n"
         "        .iter()\n"
         "        .filter(|iface| iface.state.active)\n"
         "        .flat_map(|iface| {\n"
         "            iface\n"
         "                .adjacencies(adjacencies)\n"
         "                .filter(|adj| adj.state == AdjacencyState::Up)\n"
         "        })\n"
         "        .map(|adj| adj.system_id)\n"
         "        .collect();\n"
         "    let mut batched = spf::compute_spts_hip(\n"
         "        level,\n        &nbrs,\n        false,\n        None,\n        MetricMode::HopCount,\n"
         "        instance,\n        interfaces,\n        adjacencies,\n        lsp_entries,\n    )\n"
         "    .map(|spts| spts.into_iter());\n"
         "    // ... and, from the same graph, the ancestor sets that answer reflood_list's is_on_path queries\n"
         "    let hip_sets = spf::manet_sets_hip(level, &nbrs, instance, lsp_entries)\n"
         "        .map(std::sync::Arc::new);\n"
         "    let mut hip_row = 0u32;\n\n"
         "    // Process all adjacencies on active interfaces.\n    for adj in interfaces\n"),
        ("        cache.spt_hopcount = spf::compute_spt(\n"
         "            level,\n            adj.system_id,\n            false,\n            None,\n"
         "            MetricMode::HopCount,\n            instance,\n            interfaces,\n"
         "            adjacencies,\n            lsp_entries,\n        );\n",
         "        cache.spt_hopcount = match batched.as_mut().and_then(|spts| spts.next())\n"
         "        {\n"
         "            Some(spt) => spt,\n"
         "            None => spf::compute_spt(\n"
         "                level,\n                adj.system_id,\n                false,\n                None,\n"
         "                MetricMode::HopCount,\n                instance,\n                interfaces,\n"
         "                adjacencies,\n                lsp_entries,\n            ),\n"
         "        };\n"),
    ]),
    ("holo-ospf.patch", "holo-ospf/Cargo.toml", [
        ('holo-utils = { path = "../holo-utils" }\n', 'holo-spf-hip = { path = "../holo-spf-hip" }\nholo-utils = { path = "../holo-utils" }\n'),
    ]),
    ("holo-ospf.patch", "holo-ospf/src/spf.rs", [
        ("// Maximum size of the SPF log record.\n",
         "// MI355X path of run_area (holo-spf-hip): area LSDB -> CSR, the SPT loop on the GPU, next hops through the\n"
         "// unchanged V::calc_nexthops.  `None` from it = no engine / too small / engine error: the loop below runs.\n"
         "mod hip;\n\n"
         "// Maximum size of the SPF log record.\n"),
        ("    // Clear router's routing table.\n    area.state.routers.clear();\n",
         "    // Clear router's routing table.\n    area.state.routers.clear();\n\n"
         "    // MI355X path (HOLO_SPF_HIP_DEVICE): same SPT, router table and TransitCapability, computed by the engine.\n"
         "    if let Some(spt) = hip::run_area::<V>(\n"
         "        area,\n        root_vid,\n        instance,\n        interfaces,\n        neighbors,\n        lsa_entries,\n    ) {\n"
         "        hip::finish_area(area, spt);\n        return;\n    }\n"),
    ]),
]


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    os.makedirs(OUT, exist_ok=True)
    patches = {}
    for pname, rel, edits in EDITS:
        src = open(os.path.join(ref, rel)).read()
        new = src
        for anchor, repl in edits:
            if new.count(anchor) != 1:
                raise SystemExit(f"{rel}: anchor occurs {new.count(anchor)} times: {anchor[:60]!r}")
            new = new.replace(anchor, repl)
        diff = difflib.unified_diff(src.splitlines(keepends=True), new.splitlines(keepends=True),
                                    fromfile="a/" + rel, tofile="b/" + rel, n=2)
        patches.setdefault(pname, []).append("".join(diff))
    for pname, parts in patches.items():
        with open(os.path.join(OUT, pname), "w") as f:
            f.write("".join(parts))
        print(os.path.join(OUT, pname))


if __name__ == "__main__":
    main()
