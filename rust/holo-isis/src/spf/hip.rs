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

use holo_spf_hip::{Csr, CsrCache, Engine, Tables, sys};

use super::*;

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
fn expandable(csr: &Csr, t: &Tables, r: u32, u: u32, ignore_overload: bool) -> bool {
    let f = csr.vflags[u as usize];
    if f & VF_NO_EXPAND != 0 {
        return false;
    }
    !(t.hops(r, u) != 0 && f & VF_NETWORK == 0 && !ignore_overload && f & VF_NO_TRANSIT != 0)
}

// The reference's pop order is (distance, VertexId) = (distance, index) — except on hop-count graphs, where a
// pseudonode (cost 0 from every router) is popped right after the lowest-numbered router of its own distance that
// lists it — and for roots the engine flagged HSPF_RF_EXACT, whose ranks come from `Engine::pop_ranks`.
fn rank_key(csr: &Csr, t: &Tables, r: u32, v: u32, hopcount: bool, exact: Option<&[u32]>) -> (u32, u32, u32, u32) {
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
    t: &Tables,
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
    t: &Tables,
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
        let t = eng.run(graph, &roots, run_flags).map_err(|e| e.log()).ok()?;
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
