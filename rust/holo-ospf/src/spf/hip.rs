//
// SPDX-License-Identifier: MIT
//
// holo-ospf/src/spf/hip.rs — the MI355X path of `run_area<V>` (child module of spf.rs; generic over the OSPF version:
// everything version specific stays behind the unchanged `V::vertex_lsa_find`, `V::vertex_lsa_links`,
// `V::calc_nexthops`).
//
//   here, on the instance thread   area LSDB -> CSR (the vertices the loop could ever reach, found by walking
//                                  V::vertex_lsa_links from the root; one row per Router- / Network-LSA in link order),
//                                  first-hop slots -> next hops through the unchanged V::calc_nexthops (once per slot),
//                                  the SPT map, `area.state.routers` and TransitCapability (spf.rs:627-643);
//   on the GPU (libholo_spf_hip)   the SPT loop: distances, hop counts, ECMP first-hop masks.
//
// Python twin, checked against the reference's recorded RIBs: holo_amd/ospf.py (AreaGraph, spt_from_engine,
// run_area, routers_table) and holo_amd/ospfv3.py of the engine repository.
//
// NOT compiled in the engine repository's image (no cargo / rustc): a mechanical translation of the tested twin.

use std::cell::RefCell;
use std::collections::{BTreeMap, VecDeque};

use holo_spf_hip::{Csr, CsrCache, DeviceTables, Engine, OrderedPrefixTable, RibDevice, sys};

use super::*;

// One engine per instance thread; one resident graph per area of this thread's instance.  VertexId types differ per
// OSPF version, so the cache is keyed by the vertex RANK fingerprint (the ids rendered with Debug) — the cache only
// needs equality of the vertex set.
thread_local! {
    static ENGINE: Option<&'static Engine> = Engine::from_env().map(|e| &*Box::leak(Box::new(e)));
    static GRAPHS: RefCell<BTreeMap<Ipv4Addr, CsrCache<'static, String>>> = RefCell::new(BTreeMap::new());
}

struct AreaGraph<V: Version> {
    vids: Vec<V::VertexId>,      // ascending: index = rank in VertexId order = the candidate list's tie-break
    lsas: Vec<V::VertexLsa>,
    csr: Csr,
}

// The vertices the loop can reach from the root: V::vertex_lsa_links only yields links whose target LSA exists and is
// not MaxAge, so walking it from the root finds every vertex that can enter the candidate list.
fn area_graph<V>(
    root_vid: V::VertexId,
    af: AddressFamily,
    area: &Area<V>,
    extended_lsa: bool,
    lsa_entries: &Arena<LsaEntry<V>>,
) -> Option<AreaGraph<V>>
where
    V: Version,
{
    let mut found: BTreeMap<V::VertexId, V::VertexLsa> = BTreeMap::new();
    let mut queue = VecDeque::new();
    found.insert(root_vid, V::vertex_lsa_find(af, root_vid, area, extended_lsa, lsa_entries)?);
    queue.push_back(root_vid);
    while let Some(vid) = queue.pop_front() {
        let new: Vec<_> = V::vertex_lsa_links(&found[&vid], af, area, extended_lsa, lsa_entries)
            .filter(|link| !found.contains_key(&link.id))
            .map(|link| (link.id, link.lsa))
            .collect();
        for (id, lsa) in new {
            if !found.contains_key(&id) {
                found.insert(id, lsa);
                queue.push_back(id);
            }
        }
    }
    let (vids, lsas): (Vec<_>, Vec<_>) = found.into_iter().unzip();
    let mut csr = Csr { row_ptr: vec![0], max_path_metric: u32::MAX, ..Default::default() }; // saturating add, no prune (:672)
    for lsa in &lsas {
        for link in V::vertex_lsa_links(lsa, af, area, extended_lsa, lsa_entries) {
            if let Ok(j) = vids.binary_search(&link.id) {
                csr.col.push(j as u32);
                csr.metric.push(link.cost.into());
            }
        }
        csr.row_ptr.push(csr.col.len() as u32);
        csr.vflags.push(if lsa.is_router() { 0 } else { sys::HSPF_VF_NETWORK as u8 });
    }
    Some(AreaGraph { vids, lsas, csr })
}

// The area's SPT through the engine, or None (no engine, too small to pay, root LSA missing, engine error — logged):
// the caller runs the existing loop.
pub(crate) fn run_area<V>(
    area: &Area<V>,
    root_vid: V::VertexId,
    instance: &InstanceUpView<'_, V>,
    interfaces: &Arena<Interface<V>>,
    neighbors: &Arena<Neighbor<V>>,
    lsa_entries: &Arena<LsaEntry<V>>,
) -> Option<BTreeMap<V::VertexId, Vertex<V>>>
where
    V: Version,
{
    let eng = ENGINE.with(|e| *e)?;
    let af = instance.state.af;
    let extended_lsa = instance.config.extended_lsa;
    let AreaGraph { vids, lsas, csr } = area_graph::<V>(root_vid, af, area, extended_lsa, lsa_entries)?;
    if Engine::recommend_cpu(csr.n_vertices(), csr.col.len() as u32, 1) {
        return None;
    }
    let root = vids.binary_search(&root_vid).ok()? as u32;
    let keys: Vec<String> = vids.iter().map(|v| format!("{v:?}")).collect();
    let (t, slot_table) = GRAPHS.with(|graphs| {
        let mut graphs = graphs.borrow_mut();
        let cache = graphs.entry(area.area_id).or_default();
        let graph = cache.get_or_patch(eng, keys, csr.clone()).map_err(|e| e.log()).ok()?;
        let t = eng.run(graph, &[root], sys::HSPF_RUN_NET_NEXTHOPS, None).map_err(|e| e.log()).ok()?;
        let slot_table = graph.slot_table(root).map_err(|e| e.log()).ok()?;
        Some((t, slot_table))
    })?;

    // vertices in pop order: (distance, id) — a hops == 0 network is materialised before the routers behind it — unless the
    // engine says the root's order is dynamic (zero-cost links: HSPF_RF_EXACT), then by the ranks it computed in that order
    let mut members: Vec<u32> = (0..t.n_vertices).filter(|&v| t.in_spt(0, v)).collect();
    if members.iter().any(|&v| t.exact(0, v)) {
        let rank = GRAPHS.with(|graphs| {
            let graphs = graphs.borrow();
            let resident = graphs.get(&area.area_id)?;
            let graph = Option::as_ref(&resident.graph)?;
            eng.pop_ranks(graph, &[root], sys::HSPF_RUN_NET_NEXTHOPS).map_err(|e| e.log()).ok()
        })?;
        members.sort_by_key(|&v| rank[v as usize]);
    } else {
        members.sort_by_key(|&v| (t.dist(0, v), v));
    }
    let mut lsas: Vec<Option<V::VertexLsa>> = lsas.into_iter().map(Some).collect();
    let mut spt: BTreeMap<V::VertexId, Vertex<V>> = BTreeMap::new();
    let mut slot_cache: BTreeMap<u32, Option<Nexthops<V::IpAddr>>> = BTreeMap::new();
    for &v in &members {
        let vid = vids[v as usize];
        let mut vertex = Vertex::<V>::new(vid, lsas[v as usize].take()?, t.dist(0, v), t.hops(0, v));
        for s in t.slots(0, v) {
            if !slot_cache.contains_key(&s) {
                // slot s = link j of the row of p, p in {root, networks attached to it}: V::calc_nexthops for a
                // hops == 0 parent (spf.rs:747-760), evaluated ONCE; everybody behind inherits it through the mask
                let i = slot_table.partition_point(|&(_, base)| base <= s) - 1;
                let (p, base) = slot_table[i];
                let parent = spt.get(&vids[p as usize])?;
                let link = V::vertex_lsa_links(&parent.lsa, af, area, extended_lsa, lsa_entries)
                    .filter(|link| vids.binary_search(&link.id).is_ok())
                    .nth((s - base) as usize)?;
                let nexthops = match V::calc_nexthops(
                    area, parent, link.parent, link.id, &link.lsa, interfaces, neighbors, extended_lsa, lsa_entries,
                ) {
                    Ok(nexthops) => Some(nexthops),
                    Err(error) => {
                        error.log(); // as the loop does (:717-718): logged, nothing added
                        None
                    }
                };
                slot_cache.insert(s, nexthops);
            }
            if let Some(Some(nexthops)) = slot_cache.get(&s) {
                vertex.nexthops.extend(nexthops.clone());
            }
        }
        spt.insert(vid, vertex);
    }
    Some(spt)
}

// What the loop does besides the SPT (spf.rs:627-643, 723-728): the "router" routing table, TransitCapability, the
// area's SPT and statistics.
pub(crate) fn finish_area<V>(area: &mut Area<V>, spt: BTreeMap<V::VertexId, Vertex<V>>)
where
    V: Version,
{
    for vertex in spt.values().filter(|vertex| vertex.lsa.is_router()) {
        let route = RouteRtr::new(
            area.area_id,
            PathType::IntraArea,
            vertex.lsa.router_options(),
            vertex.lsa.router_flags(),
            vertex.distance,
            vertex.nexthops.clone(),
        );
        area.state.routers.insert(vertex.lsa.router_id(), route);
        if vertex.lsa.router_flags().is_vlink_endpoint() {
            area.state.transit_capability = true;
        }
    }
    area.state.spt = spt;
    area.state.spf_run_count += 1;
    area.state.discontinuity_time = Utc::now();
}

// ---- update_rib_intra_area of every area on the device (route.rs:343-448; the per-area loop of update_rib_full, :157-160) -----
//
// `area_tables[a]` = the area's ordered prefix table (one entry per item of `V::intra_area_networks(area, ..)` in the order it
// yields them: vertex rank, metric, LS-ID of the vertex LSA; `HSPF_PFX_ENTRY_NETWORK` on Network-LSA vertices) and the map from
// the area's prefixes to the instance-wide prefix list `prefixes` (ascending: BTreeMap<IpNetwork, _> order).  Every area runs
// its root on the engine (`run_device`), its table is folded into ONE instance-wide RIB state on the device
// (`Engine::rib_fold` = hspf_rib_fold_device: better replaces, equal merges, the transit-network rule on the larger LS-ID) and
// only (metric, owner, first-hop mask) of every prefix come back.  `resolve(a, slot)` = the next hops of first-hop slot `slot`
// of area `a` (the slot replay of `run_area`: V::calc_nexthops once per slot).  Returns the intra-area RIB the unchanged
// inter-area / external steps and `update_global_rib` (route.rs:856-916) continue with; None: engine error (logged), the
// caller folds the areas on the host as before.  C++ twin, tested against the recorded RIBs and ibus sequences:
// include/holo_spf_ospf.hpp RibOnEngine / update_global_rib_device_t; Python: holo_amd/routes.py ospf_update_global_rib_device.
pub(crate) struct AreaTable {
    pub graph_root: u32,
    pub table: OrderedPrefixTable,
    pub prefix_map: Vec<u32>,
}

pub(crate) fn update_rib_intra_area_device<V, F>(
    areas: &[(Ipv4Addr, AreaTable)],
    prefixes: &[V::IpNetwork],
    max_paths: u16,
    mut resolve: F,
) -> Option<BTreeMap<V::IpNetwork, (u32, u32, Nexthops<V::IpAddr>)>>
where
    V: Version,
    F: FnMut(usize, u32) -> Option<Nexthops<V::IpAddr>>,
{
    let eng = ENGINE.with(|e| *e)?;
    // one run per area, results left in HBM; the areas' first-hop slots get word-aligned places in one instance-wide numbering
    let mut runs: Vec<DeviceTables<'static>> = Vec::with_capacity(areas.len());
    let mut offsets = Vec::with_capacity(areas.len());
    let mut words = 0u32;
    GRAPHS.with(|graphs| {
        let graphs = graphs.borrow();
        for (area_id, at) in areas {
            let resident = graphs.get(area_id)?;
            let graph = Option::as_ref(&resident.graph)?;
            let run = eng.run_device(graph, &[at.graph_root], sys::HSPF_RUN_NET_NEXTHOPS).map_err(|e| e.log()).ok()?;
            offsets.push(words);
            words += run.words;
            runs.push(run);
        }
        Some(())
    })?;
    let rib: RibDevice<'static> = eng.rib_new(prefixes.len() as u32, words.max(1)).map_err(|e| e.log()).ok()?;
    for (a, ((_, at), run)) in areas.iter().zip(&runs).enumerate() {
        eng.rib_fold(&rib, run, &at.table, &at.prefix_map, a as u32, offsets[a]).map_err(|e| e.log()).ok()?;
    }
    let (metric, owner, mask) = rib.to_host().map_err(|e| e.log()).ok()?;
    let w = words.max(1) as usize;
    let mut out = BTreeMap::new();
    for (i, prefix) in prefixes.iter().enumerate() {
        if owner[i] == u32::MAX {
            continue; // no area reaches the prefix
        }
        let mut nexthops = Nexthops::<V::IpAddr>::new();
        for (a, &off) in offsets.iter().enumerate() {
            let aw = runs[a].words as usize;
            for k in 0..aw {
                let mut m = mask[i * w + off as usize + k];
                while m != 0 {
                    let b = m.trailing_zeros();
                    m &= m - 1;
                    if let Some(nh) = resolve(a, k as u32 * 64 + b) {
                        nexthops.extend(nh);
                    }
                }
            }
        }
        // route_update's max_paths truncation: the first k by key (route.rs:918-965)
        while nexthops.len() > max_paths as usize {
            let last = nexthops.keys().next_back().cloned()?;
            nexthops.remove(&last);
        }
        out.insert(*prefix, (metric[i], owner[i], nexthops));
    }
    Some(out)
}
