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

use holo_spf_hip::{Csr, CsrCache, Engine, sys};

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

    // vertices in (distance, id) order = pop order: a hops == 0 network is materialised before the routers behind it
    let mut members: Vec<u32> = (0..t.n_vertices).filter(|&v| t.in_spt(0, v)).collect();
    members.sort_by_key(|&v| (t.dist(0, v), v));
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
