//
// SPDX-License-Identifier: MIT
//
// holo-isis/src/spf/hip.rs — the MI355X path of `compute_spt` (child module of spf.rs: it uses the parent's private
// `vertex_edges`, `zeroth_lsp`, `resolve_nexthop` and `Spt` unchanged).
//
// What runs where:
//   here, on the instance thread   LSDB -> CSR (one walk of the live LSPs with the unchanged vertex_edges), the
//                                  per-vertex gates of the loop as vertex flags, first-hop slots -> next hops through
//                                  the unchanged resolve_nexthop, Spt rebuilt in pop order;
//   on the GPU (libholo_spf_hip)   the SPT loop itself: distances, hop counts, ECMP first-hop masks for every root.
//
// Python twin with the same structure, checked against the reference's recorded RIBs:
// holo_amd/isis.py (LevelGraph, compute_spts, _slot_nexthops) of the engine repository.
//
// NOT compiled in the engine repository's image (no cargo / rustc): a mechanical translation of the tested twin.

use std::cell::RefCell;
use std::collections::{BTreeMap, BTreeSet};
use std::net::IpAddr;

use holo_spf_hip::{AncestorSets, Csr, CsrCache, DeviceRoutes, Engine, PrefixTable, RouteRecord, Tables, sys};

use super::*;
use crate::ibus;
use crate::route::{Nexthop, RouteFlags};

const VF_NETWORK: u8 = sys::HSPF_VF_NETWORK as u8;
const VF_NO_TRANSIT: u8 = sys::HSPF_VF_NO_TRANSIT as u8;
const VF_NO_EXPAND: u8 = sys::HSPF_VF_NO_EXPAND as u8;

type CacheKey = (LevelNumber, Option<MtId>, bool);

// One engine per instance thread (holo-protocol/src/lib.rs:427-430), created on first use from
// HOLO_SPF_HIP_DEVICE; the graphs of this thread's (level, topology, metric mode) triples next to it.
// The engine is leaked on purpose: it lives as long as the instance thread, and the cache borrows it.
thread_local! {
    static ENGINE: Option<&'static Engine> = Engine::from_env().map(|e| &*Box::leak(Box::new(e)));
    static GRAPHS: RefCell<BTreeMap<CacheKey, CsrCache<'static, VertexId>>> = RefCell::new(BTreeMap::new());
}

// LSDB -> (vertex ids in VertexId order, CSR).  A vertex = a LAN id that owns at least one live LSP fragment; links to
// LAN ids without one are dropped (they can never pass the two-way check).  Rows in fragment, then TLV order: exactly
// what vertex_edges yields.  Flags = the gates of the loop (spf.rs:557-604).
fn level_csr(
    level: LevelNumber,
    mt_id: Option<MtId>,
    metric_mode: MetricMode,
    instance: &InstanceUpView<'_>,
    lsp_entries: &Arena<LspEntry>,
) -> (Vec<VertexId>, Csr) {
    let lsdb = instance.state.lsdb.get(level);
    let metric_type = instance.config.metric_type.get(level);
    let vids: Vec<VertexId> = lsdb
        .iter(lsp_entries)
        .map(|lse| &lse.data)
        .filter(|lsp| lsp.seqno != 0 && lsp.rem_lifetime != 0)
        .map(|lsp| VertexId::from(LanId::from((lsp.lsp_id.system_id, lsp.lsp_id.pseudonode))))
        .collect::<BTreeSet<_>>()
        .into_iter()
        .collect();
    let mut csr = Csr {
        row_ptr: vec![0],
        max_path_metric: match metric_type {
            MetricType::Wide | MetricType::Both => MAX_PATH_METRIC_WIDE,
            MetricType::Standard => MAX_PATH_METRIC_STANDARD,
        },
        ..Default::default()
    };
    for vid in &vids {
        for link in vertex_edges(vid, mt_id, metric_mode, metric_type, lsdb, lsp_entries) {
            if let Ok(j) = vids.binary_search(&link.id) {
                csr.col.push(j as u32);
                csr.metric.push(link.cost);
            }
        }
        csr.row_ptr.push(csr.col.len() as u32);
        let pseudonode = vid.lan_id.is_pseudonode();
        let mut f = if pseudonode { VF_NETWORK } else { 0 };
        match zeroth_lsp(vid.lan_id, lsdb, lsp_entries) {
            None => f |= VF_NO_EXPAND,
            Some(z) => {
                if !pseudonode
                    && let Some(mt_id) = mt_id
                    && z.overload_bit(mt_id)
                {
                    f |= VF_NO_TRANSIT;
                }
                if mt_id == Some(MtId::Standard) && !pseudonode {
                    let supported = z.tlvs.protocols_supported.as_ref().is_some_and(|ps| {
                        [AddressFamily::Ipv4, AddressFamily::Ipv6]
                            .into_iter()
                            .all(|af| !instance.config.is_af_enabled(af) || ps.contains(Nlpid::from(af)))
                    });
                    if !supported {
                        f |= VF_NO_EXPAND;
                    }
                }
            }
        }
        csr.vflags.push(f);
    }
    (vids, csr)
}

// May vertex u be expanded in the SPT of this run (the gates, from the flags)?
fn expandable(csr: &Csr, t: &Tables<'_>, r: u32, u: u32, ignore_overload: bool) -> bool {
    let f = csr.vflags[u as usize];
    if f & VF_NO_EXPAND != 0 {
        return false;
    }
    !(t.hops(r, u) != 0 && f & VF_NETWORK == 0 && !ignore_overload && f & VF_NO_TRANSIT != 0)
}

// The reference's pop order is (distance, VertexId) = (distance, index) — except on hop-count graphs, where a
// pseudonode (cost 0 from every router) is popped right after the lowest-numbered router of its own distance that
// lists it — and for roots the engine flagged HSPF_RF_EXACT, whose ranks come from `Engine::pop_ranks`.
fn rank_key(csr: &Csr, t: &Tables<'_>, r: u32, v: u32, hopcount: bool, exact: Option<&[u32]>) -> (u32, u32, u32, u32) {
    if let Some(rank) = exact {
        return (rank[(r * t.n_vertices + v) as usize], 0, 0, 0);
    }
    let d = t.dist(r, v);
    if hopcount && csr.vflags[v as usize] & VF_NETWORK != 0 {
        let first = csr
            .row(v)
            .0
            .iter()
            .copied()
            .filter(|&u| {
                t.in_spt(r, u) && t.dist(r, u) == d && csr.vflags[u as usize] & VF_NO_EXPAND == 0 && csr.links_back(u, v)
            })
            .min();
        if let Some(u) = first {
            return (d, u, 1, v);
        }
    }
    (d, v, 0, 0)
}

// Replays the relaxations made from hops == 0 vertices in the reference's order and gives every first-hop slot its
// VertexNexthop (spf.rs:680-701).  resolve_nexthop is order dependent through `used_adjs`, and the reference calls
// it for EVERY relaxation that is not Ordering::Greater at that moment — also for candidates a later, shorter path
// replaces —, so the replay evaluates the candidate-list state at each of those moments from the final distances.
#[allow(clippy::too_many_arguments)]
fn slot_nexthops(
    csr: &Csr,
    vids: &[VertexId],
    slot_table: &[(u32, u32)],
    t: &Tables<'_>,
    r: u32,
    key: &dyn Fn(u32) -> (u32, u32, u32, u32),
    local: bool,
    level: LevelNumber,
    mt_id: Option<MtId>,
    interfaces: &Interfaces,
    adjacencies: &Arena<Adjacency>,
) -> BTreeMap<u32, VertexNexthop> {
    let ignore_overload = mt_id.is_none();
    let mut parents: Vec<(u32, u32)> =
        slot_table.iter().copied().filter(|&(p, _)| t.in_spt(r, p) && t.hops(r, p) == 0).collect();
    parents.sort_by_key(|&(p, _)| key(p));
    let mut used_adjs = BTreeSet::new();
    let mut out = BTreeMap::new();
    // distance of `target` on the candidate list just before link `upto` of `p` is processed
    let cand_before = |target: u32, p: u32, upto: usize| -> Option<u32> {
        let mut best: Option<u32> = None;
        let sources: BTreeSet<u32> = csr.row(target).0.iter().copied().collect(); // two-way => u lists target
        for u in sources {
            if !t.in_spt(r, u) || !expandable(csr, t, r, u, ignore_overload) || key(u) > key(p) {
                continue;
            }
            let a = csr.row_ptr[u as usize] as usize;
            let (col, met) = csr.row(u);
            for (k, (&c, &m)) in col.iter().zip(met).enumerate() {
                if c != target {
                    continue;
                }
                if u == p && a + k >= upto {
                    break;
                }
                let d = t.dist(r, u).saturating_add(m);
                if d <= csr.max_path_metric && best.is_none_or(|b| d < b) {
                    best = Some(d);
                }
            }
        }
        best
    };
    for (p, base) in parents {
        if !expandable(csr, t, r, p, ignore_overload) {
            continue;
        }
        let a = csr.row_ptr[p as usize] as usize;
        let (col, met) = csr.row(p);
        for (j, (&target, &cost)) in col.iter().zip(met).enumerate() {
            if !csr.links_back(target, p) {
                continue;
            }
            if t.in_spt(r, target) && key(target) < key(p) {
                continue; // already on the SPT
            }
            let d = t.dist(r, p).saturating_add(cost);
            if d > csr.max_path_metric {
                continue;
            }
            if cand_before(target, p, a + j).is_some_and(|cur| d > cur) {
                continue; // Ordering::Greater
            }
            if csr.vflags[target as usize] & VF_NETWORK != 0 {
                continue; // pseudonode target: no next hop
            }
            let target_vid = vids[target as usize];
            let mut nexthop = VertexNexthop::new(target_vid.lan_id.system_id, None, None, None);
            if local && let Some(mt_id) = mt_id {
                let parent = Vertex::new(vids[p as usize], t.dist(r, p), 0);
                let link = VertexEdge::new(target_vid, cost);
                resolve_nexthop(&mut nexthop, level, mt_id, &parent, &link, &mut used_adjs, interfaces, adjacencies);
            }
            out.insert(base + j as u32, nexthop);
        }
    }
    out
}

// Tables of root row `r` -> Spt: vertices inserted in pop order (Spt::insert keeps first_hops / second_hops in that
// order, spf.rs:224-242), next hops = the slots of the vertex's mask, parents = the tight links in relaxation order.
#[allow(clippy::too_many_arguments)]
fn spt_from_tables(
    csr: &Csr,
    vids: &[VertexId],
    slot_table: &[(u32, u32)],
    t: &Tables<'_>,
    r: u32,
    exact: Option<&[u32]>,
    local: bool,
    level: LevelNumber,
    mt_id: Option<MtId>,
    metric_mode: MetricMode,
    interfaces: &Interfaces,
    adjacencies: &Arena<Adjacency>,
) -> Spt {
    let hopcount = matches!(metric_mode, MetricMode::HopCount);
    let ignore_overload = mt_id.is_none();
    let key = |v: u32| rank_key(csr, t, r, v, hopcount, exact);
    let slots = slot_nexthops(csr, vids, slot_table, t, r, &key, local, level, mt_id, interfaces, adjacencies);
    let mut members: Vec<u32> = (0..t.n_vertices).filter(|&v| t.in_spt(r, v)).collect();
    members.sort_by_key(|&v| key(v));
    let mut spt = Spt::default();
    let mut arena_idx = BTreeMap::new();
    for &v in &members {
        let mut vertex = Vertex::new(vids[v as usize], t.dist(r, v), t.hops(r, v));
        vertex.nexthops = t.slots(r, v).filter_map(|s| slots.get(&s).cloned()).collect();
        arena_idx.insert(v, spt.insert(vertex));
    }
    // `Vertex.parents` (spf.rs:677): every relaxation that reached the vertex at its final distance, parents in pop
    // order, their links in LSP order, parallel links repeated.
    for &u in &members {
        if !expandable(csr, t, r, u, ignore_overload) {
            continue;
        }
        let (col, met) = csr.row(u);
        for (&target, &cost) in col.iter().zip(met) {
            if !t.in_spt(r, target) || key(target) <= key(u) || !csr.links_back(target, u) {
                continue;
            }
            if t.dist(r, u).saturating_add(cost) == t.dist(r, target) {
                let parent = arena_idx[&u];
                spt.arena[arena_idx[&target]].parents.push(parent);
            }
        }
    }
    spt
}

// Several SPTs of one (level, topology, metric mode) in ONE engine run: `compute_spt` calls it with one root,
// `flooding::manet::init_cache` with every Up adjacency.  None = no engine, too small to pay, or an engine error
// (logged): the caller runs the existing loop.
#[allow(clippy::too_many_arguments)]
pub(crate) fn compute_spts(
    level: LevelNumber,
    root_system_ids: &[SystemId],
    local: bool,
    mt_id: Option<MtId>,
    metric_mode: MetricMode,
    instance: &InstanceUpView<'_>,
    interfaces: &Interfaces,
    adjacencies: &Arena<Adjacency>,
    lsp_entries: &Arena<LspEntry>,
) -> Option<Vec<Spt>> {
    let eng = ENGINE.with(|e| *e)?;
    let hopcount = matches!(metric_mode, MetricMode::HopCount);
    let (vids, csr) = level_csr(level, mt_id, metric_mode, instance, lsp_entries);
    if Engine::recommend_cpu(csr.n_vertices(), csr.col.len() as u32, root_system_ids.len() as u32) {
        return None;
    }
    GRAPHS.with(|graphs| {
        let mut graphs = graphs.borrow_mut();
        let cache = graphs.entry((level, mt_id, hopcount)).or_default();
        let graph = cache.get_or_patch(eng, vids, csr).map_err(|e| e.log()).ok()?;
        let (vids, csr) = (&cache.keys, &cache.csr);
        // A root without any LSP is put into the SPT and not expanded (spf.rs:552-561): the loop handles it.
        let roots: Vec<u32> = root_system_ids
            .iter()
            .map(|sid| vids.binary_search(&VertexId::from(*sid)).ok().map(|i| i as u32))
            .collect::<Option<_>>()?;
        let run_flags = if mt_id.is_none() { sys::HSPF_RUN_IGNORE_OVERLOAD } else { 0 };
        // (packed hand-off: one word per (root, vertex) into page-locked memory — a quarter of the bytes of the four arrays)
        let t = eng.run(graph, &roots, run_flags, None).map_err(|e| e.log()).ok()?;
        // roots whose pop order is dynamic (zero-cost plateaus): their exact pop ranks, one more run
        let any_exact = (0..t.n_roots).any(|r| (0..t.n_vertices).any(|v| t.exact(r, v)));
        let exact = if any_exact { Some(eng.pop_ranks(graph, &roots, run_flags).map_err(|e| e.log()).ok()?) } else { None };
        let mut spts = Vec::with_capacity(roots.len());
        for (r, &root) in roots.iter().enumerate() {
            let slot_table = graph.slot_table(root).map_err(|e| e.log()).ok()?;
            let r = r as u32;
            let exact = exact.as_deref().filter(|_| (0..t.n_vertices).any(|v| t.exact(r, v)));
            spts.push(spt_from_tables(
                csr, vids, &slot_table, &t, r, exact, local, level, mt_id, metric_mode, interfaces, adjacencies,
            ));
        }
        Some(spts)
    })
}

#[allow(clippy::too_many_arguments)]
pub(crate) fn compute_spt(
    level: LevelNumber,
    root_system_id: SystemId,
    local: bool,
    mt_id: Option<MtId>,
    metric_mode: MetricMode,
    instance: &InstanceUpView<'_>,
    interfaces: &Interfaces,
    adjacencies: &Arena<Adjacency>,
    lsp_entries: &Arena<LspEntry>,
) -> Option<Spt> {
    compute_spts(level, &[root_system_id], local, mt_id, metric_mode, instance, interfaces, adjacencies, lsp_entries)?
        .pop()
}


// ---- routes on the device and the wire step -----------------------------------------------------
//
// compute_routes (spf.rs:840-949) + route::update_rib / update_global_rib (route.rs:185-312) for a RUNNING instance with
// one RIB table (level_type != All, one topology, SR off): nobody walks 100 000 SPT vertices or rebuilds 120 000 Route
// objects per SPF event.  What stays resident between events:
//   the level graph on the device (CsrCache: rows of the changed LSPs are patched),
//   the prefix table on the device (PrefixTable, HSPF_PFX_RESIDENT; rebuilt when a changed LSP's prefixes differ),
//   the route tables of the PREVIOUS event on the device = the "RIB held before".
// Per event: run_device -> routes_device -> routes_changed (comparison + ordered compaction + two small copies) ->
// a `Route` is built ONLY for the records, the stored RIB is updated in place, route_install / route_uninstall are called
// for them in the reference's order.  A mechanical translation of the COMPILED AND TESTED C++ form,
// include/holo_spf_isis.hpp `RibPipeline` of the engine repository (its messages equal the reference's recorded ibus
// sequences: tests/cpp/host_parity.cpp; measured 0.94 ms per LSP change at 100 000 routers: tests/cpp/dropin_e2e.cpp).
// NOT compiled here.
pub(crate) struct RibPipeline {
    prefixes: Vec<IpNetwork>,                  // BTreeMap<IpNetwork, _> order = the table's prefix order
    table: PrefixTable,
    entries: Vec<(u32, VertexNetwork)>,        // per table entry: advertising vertex index, what Route::new needs of the network
    signature: BTreeMap<LanId, Vec<(IpNetwork, u32, bool)>>,
    resident: bool,
    prev: Option<DeviceRoutes<'static>>,
    slot_nh: BTreeMap<u32, VertexNexthop>,
}

thread_local! {
    static RIBS: RefCell<BTreeMap<(LevelNumber, MtId), RibPipeline>> = RefCell::new(BTreeMap::new());
}

fn networks_of(
    level: LevelNumber,
    mt_id: MtId,
    vid: &VertexId,
    instance: &InstanceUpView<'_>,
    interfaces: &Interfaces,
    adjacencies: &Arena<Adjacency>,
    lsp_entries: &Arena<LspEntry>,
) -> Vec<VertexNetwork> {
    let lsdb = instance.state.lsdb.get(level);
    let Some(zeroth) = zeroth_lsp(vid.lan_id, lsdb, lsp_entries) else {
        return vec![];
    };
    let att_bit = !instance.config.att_ignore && zeroth.att_bit(mt_id) && !zeroth.overload_bit(mt_id);
    let ipv4_enabled = instance.config.is_af_enabled(AddressFamily::Ipv4) && mt_id == MtId::Standard;
    let ipv6_enabled = instance.config.is_af_enabled(AddressFamily::Ipv6)
        && match mt_id {
            MtId::Standard => !instance.config.is_topology_enabled(MtId::Ipv6Unicast),
            MtId::Ipv6Unicast => true,
        };
    let vertex = Vertex::new(*vid, 0, 0);
    vertex_networks(
        instance.config.level_type,
        level,
        mt_id,
        &vertex,
        att_bit,
        instance.is_l2_attached_to_backbone(mt_id, interfaces, adjacencies),
        instance.config.metric_type.get(level),
        ipv4_enabled,
        ipv6_enabled,
        lsdb,
        lsp_entries,
    )
    .collect()
}

impl RibPipeline {
    // Every (vertex, prefix, metric) the unchanged vertex_networks yields, vertices in VertexId order, CSR by prefix.
    fn build(
        level: LevelNumber,
        mt_id: MtId,
        vids: &[VertexId],
        instance: &InstanceUpView<'_>,
        interfaces: &Interfaces,
        adjacencies: &Arena<Adjacency>,
        lsp_entries: &Arena<LspEntry>,
    ) -> RibPipeline {
        let mut by_prefix: BTreeMap<IpNetwork, Vec<(u32, VertexNetwork)>> = BTreeMap::new();
        let mut signature = BTreeMap::new();
        for (v, vid) in vids.iter().enumerate() {
            let nets = networks_of(level, mt_id, vid, instance, interfaces, adjacencies, lsp_entries);
            signature.insert(vid.lan_id, nets.iter().map(|n| (n.prefix, n.metric, n.external)).collect());
            for n in nets {
                by_prefix.entry(n.prefix).or_default().push((v as u32, n));
            }
        }
        let mut table = PrefixTable { pfx_ptr: vec![0], ..Default::default() };
        let (mut prefixes, mut entries) = (Vec::new(), Vec::new());
        for (prefix, list) in by_prefix {
            prefixes.push(prefix);
            for (v, n) in list {
                table.pfx_vertex.push(v);
                table.pfx_metric.push(n.metric);
                entries.push((v, n));
            }
            table.pfx_ptr.push(table.pfx_vertex.len() as u32);
        }
        RibPipeline { prefixes, table, entries, signature, resident: false, prev: None, slot_nh: BTreeMap::new() }
    }

    // The next hops a slot mask resolves to, as Route::build_nexthops would make them from a vertex's nexthop list
    // (route.rs:118-142), truncated to max-paths like compute_routes (spf.rs:920-929).
    fn nexthops_of(&self, mask: &[u64], af: AddressFamily, max_paths: u16) -> BTreeMap<IpAddr, Nexthop> {
        RouteRecord::slots(mask)
            .filter_map(|s| self.slot_nh.get(&s))
            .filter_map(|nh| {
                let addr = match af {
                    AddressFamily::Ipv4 => nh.ipv4.map(IpAddr::V4),
                    AddressFamily::Ipv6 => nh.ipv6.map(IpAddr::V6),
                }?;
                Some((addr, Nexthop { system_id: nh.system_id, iface_idx: nh.iface_idx?, addr, sr_label: None }))
            })
            .collect::<BTreeMap<_, _>>()
            .into_iter()
            .take(max_paths as usize)
            .collect()
    }

    // The instance's RIB brought in line with the tables of the last event, which are dropped: a change that puts nothing on
    // the wire leaves no record (HSPF_DIFF_SILENT) — a route that lost its next hops (route.rs:283-301 replaces it by a fresh
    // route without INSTALLED and sends nothing), a route without next hops whose prefix went away — so a RIB kept from records
    // alone goes stale, and the next event, which compares against the RIB, would withdraw routes that are not installed
    // and skip installs that are due.  (Found in the C++ form by random chains of LSP changes, tests/test_cpp_driver.py of the
    // engine repository; include/holo_spf_isis.hpp `RibPipeline::settle_installed`.)  The metric of a route that lost its next
    // hops is taken from the tables; its CONNECTED flag is not re-derived (the vertex's hop count left with the run).
    fn settle(&mut self, rib: &mut BTreeMap<IpNetwork, Route>, max_paths: u16) {
        let Some(prev) = self.prev.take() else { return };
        let Ok((metric, entry, mask)) = prev.to_host().map_err(|e| e.log()) else { return };
        let w = prev.words as usize;
        for (i, prefix) in self.prefixes.iter().enumerate() {
            if entry[i] == u32::MAX {
                if rib.get(prefix).is_some_and(|r| !r.flags.contains(RouteFlags::INSTALLED)) {
                    rib.remove(prefix);
                }
                continue;
            }
            if self.nexthops_of(&mask[i * w..(i + 1) * w], prefix.address_family(), max_paths).is_empty()
                && let Some(route) = rib.get_mut(prefix)
            {
                route.metric = metric[i];
                route.nexthops.clear();
                route.flags.remove(RouteFlags::INSTALLED);
            }
        }
    }
}

// compute_routes + route::update_rib for `level` from device tables.  `Some(())`: the instance's RIB and the global RIB are
// up to date; `None`: not applicable / engine error (logged) — the caller runs compute_routes + route::update_rib as before.
#[allow(clippy::too_many_arguments)]
pub(crate) fn update_rib(
    level: LevelNumber,
    trigger_lans: &BTreeSet<LanId>,
    instance: &mut InstanceUpView<'_>,
    interfaces: &Interfaces,
    adjacencies: &Arena<Adjacency>,
    lsp_entries: &Arena<LspEntry>,
) -> Option<()> {
    let eng = ENGINE.with(|e| *e)?;
    if instance.config.level_type == LevelType::All || instance.config.sr.enabled || instance.config.is_topology_enabled(MtId::Ipv6Unicast) {
        return None; // L1/L2 merge, summaries, SR labels and a second topology stay with the host path
    }
    let mt_id = MtId::Standard;
    let root_system_id = instance.config.system_id?;
    let (vids, csr) = level_csr(level, Some(mt_id), MetricMode::Normal, instance, lsp_entries);
    GRAPHS.with(|graphs| {
        RIBS.with(|ribs| {
            let mut graphs = graphs.borrow_mut();
            let mut ribs = ribs.borrow_mut();
            let cache = graphs.entry((level, Some(mt_id), false)).or_default();
            let same_vertices = cache.keys == vids;
            let graph = cache.get_or_patch(eng, vids, csr).map_err(|e| e.log()).ok()?;
            let (vids, csr) = (&cache.keys, &cache.csr);
            // (no LSP of the root: the caller's compute_routes + route::update_rib handle the event, and the tables kept here
            // no longer describe the RIB afterwards)
            let Ok(root) = vids.binary_search(&VertexId::from(root_system_id)) else {
                ribs.remove(&(level, mt_id));
                return None;
            };
            let root = root as u32;
            // the prefix table: as long as the vertex set stands and no changed LSP advertises other prefixes than before
            let stale = !same_vertices
                || ribs.get(&(level, mt_id)).is_none_or(|p| {
                    trigger_lans.iter().any(|lan| {
                        let now: Vec<_> = networks_of(level, mt_id, &VertexId::from(*lan), instance, interfaces, adjacencies, lsp_entries)
                            .iter()
                            .map(|n| (n.prefix, n.metric, n.external))
                            .collect();
                        p.signature.get(lan).is_none_or(|was| *was != now)
                    })
                });
            if stale {
                if let Some(old) = ribs.get_mut(&(level, mt_id)) {
                    old.settle(instance.state.rib_mut(instance.config.level_type), instance.config.max_paths);
                }
                ribs.insert((level, mt_id), RibPipeline::build(level, mt_id, vids, instance, interfaces, adjacencies, lsp_entries));
            }
            let pipe = ribs.get_mut(&(level, mt_id))?;
            // SPT and prefix attachment on the device
            let run = eng.run_device(graph, &[root], 0).map_err(|e| e.log()).ok()?;
            let fresh = eng.routes_device(&run, &pipe.table, pipe.resident).map_err(|e| e.log()).ok()?;
            pipe.resident = true;
            // first-hop slots -> next hops, every event (which relaxations the root makes depends on distances elsewhere)
            // (distance / hops / flags only: the replay reads the root's two-hop neighbourhood, the masks stay on the device)
            let t = run.to_host_without_masks().map_err(|e| e.log()).ok()?;
            let exact = if (0..t.n_vertices).any(|v| t.exact(0, v)) { Some(eng.pop_ranks(graph, &[root], 0).map_err(|e| e.log()).ok()?) } else { None };
            let key = |v: u32| rank_key(csr, &t, 0, v, false, exact.as_deref());
            let slot_table = graph.slot_table(root).map_err(|e| e.log()).ok()?;
            let slot_nh = slot_nexthops(csr, vids, &slot_table, &t, 0, &key, true, level, Some(mt_id), interfaces, adjacencies);
            let same_slots = slot_nh.len() == pipe.slot_nh.len()
                && slot_nh.iter().zip(&pipe.slot_nh).all(|((sa, a), (sb, b))| {
                    sa == sb && a.system_id == b.system_id && a.iface_idx == b.iface_idx && a.ipv4 == b.ipv4 && a.ipv6 == b.ipv6
                });
            let max_paths = instance.config.max_paths;
            let rib = instance.state.rib_mut(instance.config.level_type);
            if pipe.prev.is_some() && !same_slots {
                pipe.settle(rib, max_paths); // a slot means another next hop now: the old masks are void (read with the OLD slots)
            }
            pipe.slot_nh = slot_nh;
            // nothing comparable on the device: the stored RIB as "held before" (poisoned metric: every such pair comes back
            // and is decided below, record by record)
            let host_old = pipe.prev.is_none();
            if host_old {
                let (p, w) = (pipe.prefixes.len(), fresh.words as usize);
                let (mut bm, mut be, mut nm) = (vec![u32::MAX; p], vec![u32::MAX; p], vec![0u64; p * w]);
                for (i, prefix) in pipe.prefixes.iter().enumerate() {
                    if let Some(route) = rib.get(prefix) {
                        bm[i] = 0xFFFF_FFFE;
                        be[i] = 0;
                        if !route.nexthops.is_empty() {
                            nm[i * w] = 1;
                        }
                    }
                }
                pipe.prev = Some(eng.routes_upload(1, p as u32, fresh.words, &bm, &be, &nm).map_err(|e| e.log()).ok()?);
            }
            let records = eng.routes_changed(pipe.prev.as_ref()?, &fresh).map_err(|e| e.log()).ok()?;
            // the records: installs in prefix order, then the withdrawals (update_global_rib, route.rs:254-312)
            let mut withdrawn = vec![];
            for rec in &records {
                let prefix = pipe.prefixes[rec.prefix as usize];
                let af = prefix.address_family();
                if rec.new_entry == u32::MAX {
                    // no route any more: uninstall what was installed (:303-310)
                    if let Some(old) = rib.remove(&prefix)
                        && old.flags.contains(RouteFlags::INSTALLED)
                    {
                        withdrawn.push((prefix, old));
                    }
                    continue;
                }
                // the new route, built for THIS prefix only: owner entry = first vertex attaining the metric (Route::new),
                // next hops = union over the tied vertices = the slot mask (merge_nexthops), max-paths applied
                let (owner, network) = &pipe.entries[rec.new_entry as usize];
                let vertex = Vertex::new(vids[*owner as usize], t.dist(0, *owner), t.hops(0, *owner));
                let mut route = Route::new(&vertex, network, level);
                route.metric = rec.new_metric;
                route.nexthops = pipe.nexthops_of(&rec.new_mask, af, max_paths);
                let mut old_sr_label = None;
                if let Some(old) = rib.get(&prefix) {
                    old_sr_label = old.sr_label;
                    // the reference's "unchanged" (:268-277): compared on resolved next hops (two slots may resolve to one
                    // adjacency), against what the route WAS — the old record when the previous tables were comparable
                    let was_same = if host_old {
                        old.metric == route.metric && old.tag == route.tag && old.nexthops == route.nexthops
                    } else {
                        rec.old_entry != u32::MAX
                            && rec.old_metric == route.metric
                            && pipe.nexthops_of(&rec.old_mask, af, max_paths) == route.nexthops
                    };
                    if was_same {
                        if old.flags.contains(RouteFlags::INSTALLED) {
                            route.flags.insert(RouteFlags::INSTALLED);
                        }
                        rib.insert(prefix, route);
                        continue;
                    }
                }
                if !route.flags.contains(RouteFlags::CONNECTED) && !route.nexthops.is_empty() {
                    let distance = route.distance(instance.config);
                    ibus::tx::route_install(&instance.tx.ibus, &prefix, &route, old_sr_label, distance, interfaces);
                    route.flags.insert(RouteFlags::INSTALLED);
                }
                rib.insert(prefix, route);
            }
            if host_old {
                // routes of the RIB whose prefix the (rebuilt) table does not list any more
                let gone: Vec<IpNetwork> = rib.keys().filter(|p| pipe.prefixes.binary_search(p).is_err()).copied().collect();
                for prefix in gone {
                    if let Some(old) = rib.remove(&prefix)
                        && old.flags.contains(RouteFlags::INSTALLED)
                    {
                        withdrawn.push((prefix, old));
                    }
                }
                withdrawn.sort_by(|a, b| a.0.cmp(&b.0)); // one pass over the old RIB in its order (route.rs:303-310)
            }
            for (prefix, route) in withdrawn {
                ibus::tx::route_uninstall(&instance.tx.ibus, &prefix, &route);
            }
            pipe.prev = Some(fresh);
            Some(())
        })
    })
}

// ---- flooding::manet on device tables -------------------------------------------------------------------
//
// reflood_list (flooding/manet.rs:99-173) asks Spt::is_on_path (a DFS over every parent, spf.rs:261-286) once per (second
// hop, LSP originator) and once per (remote neighbour, second hop).  For the hop-count SPTs of ALL Up adjacencies — one
// batched run — hspf_ancestors_device leaves, per neighbour root, the bit sets "level-1 / level-2 routers above vertex v":
// every is_on_path of reflood_list becomes one bit test.  C++ form (tested against the literal restatement on the recorded
// topologies): holo_amd.isis.manet_init_cache_device / reflood_list_device of the engine repository.
#[derive(Debug)]
pub(crate) struct ManetSets {
    pub vids: Vec<VertexId>,
    pub first_hops: AncestorSets,   // level 1: the remote-neighbour list
    pub second_hops: AncestorSets,  // level 2
}

impl ManetSets {
    // is_on_path(ancestor, descendant) in the hop-count SPT of neighbour row `r`; None: not answerable from the sets
    // (ancestor is neither a first nor a second hop of that root, or the root ran on the sequential kernel): the caller
    // keeps Spt::is_on_path.
    pub(crate) fn is_on_path(&self, r: u32, ancestor: SystemId, descendant: SystemId) -> Option<bool> {
        let a = self.vids.binary_search(&VertexId::from(ancestor)).ok()? as u32;
        let d = self.vids.binary_search(&VertexId::from(descendant)).ok()? as u32;
        self.first_hops.is_on_path(r, a, d).or_else(|| self.second_hops.is_on_path(r, a, d))
    }
}

pub(crate) fn manet_sets(
    level: LevelNumber,
    nbrs: &[SystemId],
    instance: &InstanceUpView<'_>,
    lsp_entries: &Arena<LspEntry>,
) -> Option<ManetSets> {
    let eng = ENGINE.with(|e| *e)?;
    let (vids, csr) = level_csr(level, None, MetricMode::HopCount, instance, lsp_entries);
    GRAPHS.with(|graphs| {
        let mut graphs = graphs.borrow_mut();
        let cache = graphs.entry((level, None, true)).or_default();
        let graph = cache.get_or_patch(eng, vids, csr).map_err(|e| e.log()).ok()?;
        let roots: Vec<u32> =
            nbrs.iter().map(|sid| cache.keys.binary_search(&VertexId::from(*sid)).ok().map(|i| i as u32)).collect::<Option<_>>()?;
        let flags = sys::HSPF_RUN_IGNORE_OVERLOAD;
        let run = eng.run_device(graph, &roots, flags).map_err(|e| e.log()).ok()?;
        let first_hops = eng.ancestors_device(graph, &roots, flags, &run, 1).map_err(|e| e.log()).ok()?;
        let second_hops = eng.ancestors_device(graph, &roots, flags, &run, 2).map_err(|e| e.log()).ok()?;
        Some(ManetSets { vids: cache.keys.clone(), first_hops, second_hops })
    })
}
