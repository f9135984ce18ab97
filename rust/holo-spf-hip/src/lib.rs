//! Safe wrapper over the C ABI of libholo_spf_hip.so (`sys`, generated from include/holo_spf_hip.h).
//!
//! One `Engine` per protocol-instance thread (holo-protocol/src/lib.rs:427-430: an instance runs on its own OS thread
//! and processes its messages sequentially): `Send`, not `Sync`.  Every failure is an integer code turned into
//! `Error`; callers log it and keep their existing CPU loop — the convention of
//! `Error::SpfRootNotFound(area).log(); return;` (holo-ospf/src/spf.rs:605-610).  Nothing panics or aborts across the
//! boundary (the library catches C++ exceptions at every entry point).
//!
//! NOT compiled in the engine repository's image (no cargo / rustc there): kept mechanical, one C call per method.

pub mod sys;

use std::ffi::{CStr, c_void};
use std::fmt;
use std::marker::PhantomData;
use std::ptr;

/// An `HSPF_E_*` code with the library's detail text.
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub detail: String,
}

impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        let what = unsafe { CStr::from_ptr(sys::hspf_strerror(self.code)) }.to_string_lossy();
        write!(f, "holo-spf-hip: {} ({}): {}", what, self.code, self.detail)
    }
}

impl std::error::Error for Error {}

impl Error {
    /// Logged, never fatal: the caller falls back to its own loop.
    pub fn log(&self) {
        tracing::warn!(code = self.code, detail = %self.detail, "SPF engine error, falling back to the CPU loop");
    }
}

/// The caller's CSR of one area / level x topology (include/holo_spf_hip.h `hspf_csr`): vertex index = rank in the
/// reference's VertexId order, links in LSA / LSP order, already filtered by the per-protocol rules.
#[derive(Debug, Default, Clone, PartialEq, Eq)]
pub struct Csr {
    pub row_ptr: Vec<u32>,
    pub col: Vec<u32>,
    pub metric: Vec<u32>,
    pub vflags: Vec<u8>,
    pub max_path_metric: u32,
}

impl Csr {
    pub fn n_vertices(&self) -> u32 {
        self.vflags.len() as u32
    }
    pub fn row(&self, v: u32) -> (&[u32], &[u32]) {
        let (a, b) = (self.row_ptr[v as usize] as usize, self.row_ptr[v as usize + 1] as usize);
        (&self.col[a..b], &self.metric[a..b])
    }
    /// Does vertex `t` list `v` (the two-way check of the reference loop, cost not compared)?
    pub fn links_back(&self, t: u32, v: u32) -> bool {
        self.row(t).0.contains(&v)
    }
}

/// The LSDB records of `Engine::upload_keyed` (include/holo_spf_hip.h `hspf_keyed_lsdb`).
#[derive(Debug, Default, Clone, PartialEq, Eq)]
pub struct KeyedLsdb {
    pub vertex_key: Vec<u64>,
    pub row_ptr: Vec<u32>,
    pub target_key: Vec<u64>,
    pub metric: Vec<u32>,
    pub vflags: Vec<u8>,
    pub max_path_metric: u32,
}

/// One replaced row of `Graph::patch`.
#[derive(Debug, Clone)]
pub struct RowPatch {
    pub vertex: u32,
    pub col: Vec<u32>,
    pub metric: Vec<u32>,
    pub vflags: u8,
}

/// Page-locked host memory from `hspf_host_alloc`: packed results cross the bus at full speed into it.
pub struct PinnedBuf<'e> {
    eng: &'e Engine,
    p: *mut c_void,
    bytes: usize,
}

impl PinnedBuf<'_> {
    pub fn len(&self) -> usize {
        self.bytes
    }
    pub fn is_empty(&self) -> bool {
        self.bytes == 0
    }
}

impl Drop for PinnedBuf<'_> {
    fn drop(&mut self) {
        unsafe { sys::hspf_host_free(self.eng.ctx, self.p) }
    }
}

/// Device memory from `hspf_device_alloc` (tables that stay in HBM between `run_device`, `routes_device` and
/// `routes_changed`).
pub struct DeviceBuf<'e> {
    eng: &'e Engine,
    p: *mut c_void,
    bytes: usize,
}

impl DeviceBuf<'_> {
    pub fn to_host<T: Copy + Default>(&self, count: usize) -> Result<Vec<T>, Error> {
        let mut v = vec![T::default(); count];
        let bytes = count * std::mem::size_of::<T>();
        // a real check: the pool hands out blocks by size class, a count beyond what was asked for would read a
        // neighbour's block — or past the allocation — in a release build
        if bytes > self.bytes {
            return Err(self.eng.err(sys::HSPF_E_INVAL));
        }
        let rc = unsafe { sys::hspf_device_to_host(self.eng.ctx, v.as_mut_ptr() as *mut c_void, self.p, bytes) };
        if rc != sys::HSPF_OK {
            return Err(self.eng.err(rc));
        }
        Ok(v)
    }
}

impl Drop for DeviceBuf<'_> {
    fn drop(&mut self) {
        // back to the engine's pool (freed with the engine)
        self.eng.pool.borrow_mut().entry(Engine::size_class(self.bytes)).or_default().push(self.p);
    }
}

/// Where the per-(root, vertex) values of a run live on the host.
enum Repr<'e> {
    /// `hspf_run`: four arrays, 8 + 8 `words` bytes per (root, vertex).
    Full { words: u32, dist: Vec<u32>, hops: Vec<u16>, flags: Vec<u16>, mask: Vec<u64> },
    /// `hspf_run_packed` (ABI 7): ONE 4- or 8-byte word per (root, vertex) in page-locked memory + the run's field
    /// positions; a vertex is decoded where it is looked at (include/holo_spf_hip.h "packed results").
    Packed { buf: PinnedBuf<'e>, layout: sys::hspf_packed_layout, root_status: Vec<u8> },
}

/// Per (root, vertex) results of a run, row-major `[root][vertex]`.
pub struct Tables<'e> {
    pub n_roots: u32,
    pub n_vertices: u32,
    repr: Repr<'e>,
}

impl<'e> Tables<'e> {
    #[inline]
    fn at(&self, root: u32, v: u32) -> usize {
        root as usize * self.n_vertices as usize + v as usize
    }
    #[inline]
    fn word(&self, buf: &PinnedBuf<'e>, layout: &sys::hspf_packed_layout, i: usize) -> u64 {
        // (hspf_packed_word of the header)
        unsafe {
            if layout.word_bytes == 4 {
                *(buf.p as *const u32).add(i) as u64
            } else {
                *(buf.p as *const u64).add(i)
            }
        }
    }
    pub fn in_spt(&self, root: u32, v: u32) -> bool {
        match &self.repr {
            Repr::Full { flags, .. } => flags[self.at(root, v)] & sys::HSPF_RF_IN_SPT as u16 != 0,
            Repr::Packed { buf, layout, .. } => self.word(buf, layout, self.at(root, v)) < layout.not_reached,
        }
    }
    /// The root went through the sequential kernel (its pop order is not the static one): `Engine::pop_ranks`.
    pub fn exact(&self, root: u32, v: u32) -> bool {
        match &self.repr {
            Repr::Full { flags, .. } => flags[self.at(root, v)] & sys::HSPF_RF_EXACT as u16 != 0,
            Repr::Packed { root_status, .. } => root_status[root as usize] & sys::HSPF_ROOT_EXACT as u8 != 0 && self.in_spt(root, v),
        }
    }
    pub fn dist(&self, root: u32, v: u32) -> u32 {
        match &self.repr {
            Repr::Full { dist, .. } => dist[self.at(root, v)],
            Repr::Packed { buf, layout, .. } => {
                let w = self.word(buf, layout, self.at(root, v));
                if w < layout.not_reached { (w >> layout.dist_shift) as u32 } else { sys::HSPF_DIST_INF }
            }
        }
    }
    pub fn hops(&self, root: u32, v: u32) -> u16 {
        match &self.repr {
            Repr::Full { hops, .. } => hops[self.at(root, v)],
            Repr::Packed { buf, layout, .. } => {
                let w = self.word(buf, layout, self.at(root, v));
                if w < layout.not_reached { ((w >> layout.hops_shift) & layout.hops_mask as u64) as u16 } else { 0 }
            }
        }
    }
    /// First-hop slots of (root, v), ascending.
    pub fn slots(&self, root: u32, v: u32) -> impl Iterator<Item = u32> + '_ {
        let (words, base) = match &self.repr {
            Repr::Full { words, .. } => (*words as usize, self.at(root, v) * *words as usize),
            Repr::Packed { .. } => (1usize, 0usize),
        };
        (0..words).flat_map(move |w| {
            let mut m = match &self.repr {
                Repr::Full { mask, .. } => mask[base + w],
                Repr::Packed { buf, layout, .. } => {
                    let x = self.word(buf, layout, self.at(root, v));
                    if x < layout.not_reached { x & ((1u64 << layout.mask_bits) - 1) } else { 0 }
                }
            };
            std::iter::from_fn(move || {
                if m == 0 {
                    return None;
                }
                let b = m.trailing_zeros();
                m &= m - 1;
                Some(w as u32 * 64 + b)
            })
        })
    }
    /// The page-locked buffer of a packed result, for the next run of the same size (`Engine::run_packed`).
    pub fn into_buffer(self) -> Option<PinnedBuf<'e>> {
        match self.repr {
            Repr::Packed { buf, .. } => Some(buf),
            Repr::Full { .. } => None,
        }
    }
}

/// `hspf_run_device` results left in HBM: input of `routes_device` / `ancestors_device`.
pub struct DeviceTables<'e> {
    pub n_roots: u32,
    pub n_vertices: u32,
    pub words: u32,
    dist: DeviceBuf<'e>,
    hops: DeviceBuf<'e>,
    flags: DeviceBuf<'e>,
    mask: DeviceBuf<'e>,
}

impl<'e> DeviceTables<'e> {
    /// Distance, hops and flags only (the first-hop slot replay reads nothing else; half the bytes of `to_host`): the mask
    /// accessors of the returned tables see empty masks.
    pub fn to_host_without_masks(&self) -> Result<Tables<'e>, Error> {
        let cells = self.n_roots as usize * self.n_vertices as usize;
        Ok(Tables {
            n_roots: self.n_roots,
            n_vertices: self.n_vertices,
            repr: Repr::Full {
                words: self.words,
                dist: self.dist.to_host(cells)?,
                hops: self.hops.to_host(cells)?,
                flags: self.flags.to_host(cells)?,
                mask: vec![0u64; cells * self.words as usize],
            },
        })
    }

    /// The copy the host side still needs for next-hop resolution (one root: 1.6 MB at 100 000 vertices).
    pub fn to_host(&self) -> Result<Tables<'e>, Error> {
        let cells = self.n_roots as usize * self.n_vertices as usize;
        Ok(Tables {
            n_roots: self.n_roots,
            n_vertices: self.n_vertices,
            repr: Repr::Full {
                words: self.words,
                dist: self.dist.to_host(cells)?,
                hops: self.hops.to_host(cells)?,
                flags: self.flags.to_host(cells)?,
                mask: self.mask.to_host(cells * self.words as usize)?,
            },
        })
    }
}

/// The (vertex, prefix, metric) advertisements of one level / topology as CSR by prefix (`hspf_prefix_table`): entries of
/// a prefix in ascending vertex index = the reference's `Spt::iter` order.  Root independent: built once per LSDB
/// generation, kept on the device (`HSPF_PFX_RESIDENT`).
#[derive(Debug, Default, Clone)]
pub struct PrefixTable {
    pub pfx_ptr: Vec<u32>,
    pub pfx_vertex: Vec<u32>,
    pub pfx_metric: Vec<u32>,
    pub flags: u32,
}

impl PrefixTable {
    pub fn n_prefixes(&self) -> u32 {
        self.pfx_ptr.len().saturating_sub(1) as u32
    }
}

/// Route tables of one `routes_device` call (or an uploaded set) in HBM: `hspf_routes`.
pub struct DeviceRoutes<'e> {
    pub n_roots: u32,
    pub n_prefixes: u32,
    pub words: u32,
    best_metric: DeviceBuf<'e>,
    best_entry: DeviceBuf<'e>,
    nexthop_mask: DeviceBuf<'e>,
}

impl DeviceRoutes<'_> {
    /// Metric, owner entry (all ones: no route) and next-hop mask words of every (root, prefix), on the host: what a caller
    /// that keeps its own RIB needs when it has to drop these tables (changes that put nothing on the wire leave no record).
    pub fn to_host(&self) -> Result<(Vec<u32>, Vec<u32>, Vec<u64>), Error> {
        let rp = self.n_roots as usize * self.n_prefixes as usize;
        Ok((self.best_metric.to_host(rp)?, self.best_entry.to_host(rp)?, self.nexthop_mask.to_host(rp * self.words as usize)?))
    }
    fn raw(&self) -> sys::hspf_routes {
        sys::hspf_routes {
            best_metric: self.best_metric.p as *mut u32,
            best_entry: self.best_entry.p as *mut u32,
            nexthop_mask: self.nexthop_mask.p as *mut u64,
        }
    }
}

/// One changed (root, prefix) pair of `routes_changed`: what the route IS (`new_*`) and what it WAS (`old_*`).
#[derive(Debug, Clone)]
pub struct RouteRecord {
    pub root: u32,
    pub prefix: u32,
    pub action: u32, // sys::HSPF_DIFF_INSTALL | sys::HSPF_DIFF_WITHDRAW
    pub new_metric: u32,
    pub new_entry: u32, // 0xFFFFFFFF: the prefix has no route any more
    pub new_mask: Vec<u64>,
    pub old_metric: u32,
    pub old_entry: u32, // 0xFFFFFFFF: there was no route
    pub old_mask: Vec<u64>,
}

impl RouteRecord {
    pub fn slots(mask: &[u64]) -> impl Iterator<Item = u32> + '_ {
        mask.iter().enumerate().flat_map(|(w, &m)| {
            let mut m = m;
            std::iter::from_fn(move || {
                if m == 0 {
                    return None;
                }
                let b = m.trailing_zeros();
                m &= m - 1;
                Some(w as u32 * 64 + b)
            })
        })
    }
}

/// A `run_packed_async` in flight: buffer and status array stay here until `wait_packed` turns them into `Tables`.
pub struct PackedTicket<'e> {
    eng: &'e Engine,
    ticket: u64,
    buf: Option<PinnedBuf<'e>>,
    root_status: Option<Box<[u8]>>,
    n_roots: u32,
    n_vertices: u32,
}

/// A ticket that is dropped without `wait_packed` (an early `?`, an unwinding panic) must not free its page-locked buffer
/// and status array under the lane thread and the copy stream that still write them (a use-after-free reachable
/// from safe code): the drop waits for the run first; the fields go afterwards.
impl Drop for PackedTicket<'_> {
    fn drop(&mut self) {
        if self.buf.is_some() {
            unsafe { sys::hspf_wait(self.eng.ctx, self.ticket, ptr::null_mut()) };
        }
    }
}

/// Level-L ancestor bit sets of every (root, vertex) of a hop-count run (`hspf_ancestors_device`).
#[derive(Debug, Default, Clone)]
pub struct AncestorSets {
    pub n_vertices: u32,
    pub words: u32,
    pub level_rank: Vec<u32>,  // [root][vertex]: rank among the root's level-L routers, u32::MAX elsewhere
    pub level_count: Vec<u32>, // [root]; u32::MAX: no sets for this root (sequential kernel)
    pub anc: Vec<u64>,         // [root][vertex][words]
}

impl AncestorSets {
    pub fn available(&self, root: u32) -> bool {
        self.level_count[root as usize] != u32::MAX
    }
    /// `Spt::is_on_path(ancestor, descendant)` for a level-L router `ancestor` of root row `root`.
    pub fn is_on_path(&self, root: u32, ancestor: u32, descendant: u32) -> Option<bool> {
        let rank = self.level_rank[root as usize * self.n_vertices as usize + ancestor as usize];
        if rank == u32::MAX || !self.available(root) {
            return None;
        }
        let base = (root as usize * self.n_vertices as usize + descendant as usize) * self.words as usize;
        Some(self.anc[base + (rank / 64) as usize] >> (rank % 64) & 1 != 0)
    }
}

pub struct Engine {
    ctx: *mut sys::hspf_ctx,
    /// Device blocks of dropped `DeviceBuf`s, by size class (64 KB steps), handed out again by `device_alloc`: a running
    /// instance makes the same few allocations every SPF event, and `hipMalloc` / `hipFree` cost 10-50 us each and
    /// synchronise the device (the compiled C++ twin's `HipPool`: LSP change -> messages 0.94 -> 0.81 ms).  Every engine call
    /// that touches a block is synchronous, so a returned block is idle.
    pool: std::cell::RefCell<std::collections::BTreeMap<usize, Vec<*mut c_void>>>,
}

// One OS thread per protocol instance; a context is used by exactly one thread at a time.
unsafe impl Send for Engine {}

pub struct Graph<'e> {
    eng: &'e Engine,
    g: *mut sys::hspf_graph,
    pub n: u32,
    _not_sync: PhantomData<*mut ()>,
}

impl Engine {
    fn err(&self, code: i32) -> Error {
        let detail = unsafe { CStr::from_ptr(sys::hspf_last_error(self.ctx)) }.to_string_lossy().into_owned();
        Error { code, detail }
    }

    pub fn new(device: i32) -> Result<Self, Error> {
        if unsafe { sys::hspf_abi_version() } != sys::HSPF_ABI_VERSION {
            return Err(Error { code: sys::HSPF_E_INTERNAL, detail: "libholo_spf_hip.so: ABI version mismatch".into() });
        }
        let mut ctx = ptr::null_mut();
        let rc = unsafe { sys::hspf_init(device, &mut ctx) };
        if rc != sys::HSPF_OK {
            return Err(Error { code: rc, detail: "hspf_init".into() });
        }
        Ok(Engine { ctx, pool: Default::default() })
    }

    /// `HOLO_SPF_HIP_DEVICE=<ordinal>`; unset or unusable => `None` => today's code path, byte for byte.
    pub fn from_env() -> Option<Self> {
        let dev = std::env::var("HOLO_SPF_HIP_DEVICE").ok()?.parse::<i32>().ok()?;
        match Engine::new(dev) {
            Ok(e) => Some(e),
            Err(e) => {
                e.log();
                None
            }
        }
    }

    /// `true`: these runs are too small to pay for a launch — keep the CPU loop (hspf_recommend_cpu).
    pub fn recommend_cpu(n_vertices: u32, n_edges: u32, n_roots: u32) -> bool {
        unsafe { sys::hspf_recommend_cpu(n_vertices, n_edges, n_roots) != 0 }
    }

    pub fn upload(&self, csr: &Csr) -> Result<Graph<'_>, Error> {
        let c = sys::hspf_csr {
            n_vertices: csr.n_vertices(),
            n_edges: csr.col.len() as u32,
            row_ptr: csr.row_ptr.as_ptr(),
            col: csr.col.as_ptr(),
            metric: csr.metric.as_ptr(),
            vflags: csr.vflags.as_ptr(),
            max_path_metric: csr.max_path_metric,
        };
        let mut g = ptr::null_mut();
        let rc = unsafe { sys::hspf_graph_upload(self.ctx, &c, &mut g) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(Graph { eng: self, g, n: csr.n_vertices(), _not_sync: PhantomData })
    }

    /// `hspf_graph_upload_keyed` (ABI 8): the LSDB's records as they sit there — vertices by 64-bit key in any order, links
    /// by target key in LSP / LSA order, targets unresolved.  The device ranks the keys (ascending key order = the reference's
    /// `VertexId` order, holo-isis/src/spf.rs:96-100), resolves the targets and drops links to vertices the LSDB does not
    /// hold (`vertex_edges` does not yield them, holo-isis/src/spf.rs:1013-1128).  Returns the graph and, per input
    /// vertex, its index in it; `Graph::export_csr` reads the CSR it built back for the host mirrors.
    pub fn upload_keyed(&self, lsdb: &KeyedLsdb) -> Result<(Graph<'_>, Vec<u32>), Error> {
        let n = lsdb.vertex_key.len();
        if lsdb.row_ptr.len() != n + 1 || lsdb.vflags.len() != n || lsdb.metric.len() != lsdb.target_key.len()
            || lsdb.row_ptr.last().copied().unwrap_or(0) as usize != lsdb.target_key.len()
        {
            return Err(Error { code: sys::HSPF_E_INVAL, detail: "upload_keyed: array lengths disagree".into() });
        }
        let c = sys::hspf_keyed_lsdb {
            n_vertices: n as u32,
            n_links: lsdb.target_key.len() as u32,
            vertex_key: lsdb.vertex_key.as_ptr(),
            row_ptr: lsdb.row_ptr.as_ptr(),
            target_key: lsdb.target_key.as_ptr(),
            metric: lsdb.metric.as_ptr(),
            vflags: lsdb.vflags.as_ptr(),
            max_path_metric: lsdb.max_path_metric,
        };
        let mut g = ptr::null_mut();
        let mut rank = vec![0u32; n];
        let rc = unsafe { sys::hspf_graph_upload_keyed(self.ctx, &c, &mut g, rank.as_mut_ptr()) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok((Graph { eng: self, g, n: n as u32, _not_sync: PhantomData }, rank))
    }

    pub fn host_alloc(&self, bytes: usize) -> Result<PinnedBuf<'_>, Error> {
        let mut p = ptr::null_mut();
        let rc = unsafe { sys::hspf_host_alloc(self.ctx, bytes, &mut p) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(PinnedBuf { eng: self, p, bytes })
    }

    fn size_class(bytes: usize) -> usize {
        (bytes.max(1) + 65535) & !65535
    }

    pub fn device_alloc(&self, bytes: usize) -> Result<DeviceBuf<'_>, Error> {
        let class = Self::size_class(bytes);
        if let Some(p) = self.pool.borrow_mut().get_mut(&class).and_then(|free| free.pop()) {
            return Ok(DeviceBuf { eng: self, p, bytes });
        }
        let mut p = ptr::null_mut();
        let rc = unsafe { sys::hspf_device_alloc(self.ctx, class, &mut p) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(DeviceBuf { eng: self, p, bytes })
    }

    fn device_from<T: Copy>(&self, v: &[T]) -> Result<DeviceBuf<'_>, Error> {
        let bytes = std::mem::size_of_val(v);
        let d = self.device_alloc(bytes)?;
        let rc = unsafe { sys::hspf_host_to_device(self.ctx, d.p, v.as_ptr() as *const c_void, bytes) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(d)
    }

    /// Results on the host.  First the PACKED hand-off (`hspf_run_packed`: one word per (root, vertex) into page-locked
    /// memory, a quarter of `hspf_run`'s bytes over the bus; `reuse` = the buffer of an earlier result); runs whose
    /// results do not fit packed words (`HSPF_E_NO_PACKED`: more than 24 first-hop slots) take `hspf_run`.  `roots` may
    /// hold `HSPF_NO_ROOT` padding.
    pub fn run<'e>(&'e self, g: &Graph<'_>, roots: &[u32], run_flags: u32, reuse: Option<PinnedBuf<'e>>) -> Result<Tables<'e>, Error> {
        match self.run_packed(g, roots, run_flags, reuse) {
            Err(e) if e.code == sys::HSPF_E_NO_PACKED => self.run_full(g, roots, run_flags),
            other => other,
        }
    }

    /// `hspf_run_packed` (ABI 7).
    pub fn run_packed<'e>(&'e self, g: &Graph<'_>, roots: &[u32], run_flags: u32, reuse: Option<PinnedBuf<'e>>) -> Result<Tables<'e>, Error> {
        let need = 8 * roots.len() * g.n as usize;
        let buf = match reuse {
            Some(b) if b.len() >= need => b,
            _ => self.host_alloc(need)?,
        };
        let mut layout = std::mem::MaybeUninit::<sys::hspf_packed_layout>::zeroed();
        let mut root_status = vec![0u8; roots.len()];
        let rc = unsafe {
            sys::hspf_run_packed(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, run_flags, buf.p, buf.bytes, layout.as_mut_ptr(), root_status.as_mut_ptr())
        };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(Tables { n_roots: roots.len() as u32, n_vertices: g.n, repr: Repr::Packed { buf, layout: unsafe { layout.assume_init() }, root_status } })
    }

    /// `hspf_run_packed_async`: the run AND its copy to the host belong to the ticket (the copy of one batch crosses the
    /// bus while the next computes).  `wait_packed` gives the tables.
    pub fn run_packed_async<'e>(&'e self, g: &Graph<'_>, roots: &[u32], run_flags: u32, reuse: Option<PinnedBuf<'e>>) -> Result<PackedTicket<'e>, Error> {
        let need = 8 * roots.len() * g.n as usize;
        let buf = match reuse {
            Some(b) if b.len() >= need => b,
            _ => self.host_alloc(need)?,
        };
        let mut root_status = vec![0u8; roots.len()].into_boxed_slice();
        let mut ticket = 0u64;
        let rc = unsafe {
            sys::hspf_run_packed_async(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, run_flags, buf.p, buf.bytes, root_status.as_mut_ptr(), &mut ticket)
        };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(PackedTicket { eng: self, ticket, buf: Some(buf), root_status: Some(root_status), n_roots: roots.len() as u32, n_vertices: g.n })
    }

    pub fn wait_packed<'e>(&'e self, mut t: PackedTicket<'e>) -> Result<Tables<'e>, Error> {
        let mut layout = std::mem::MaybeUninit::<sys::hspf_packed_layout>::zeroed();
        let rc = unsafe { sys::hspf_wait_packed(self.ctx, t.ticket, layout.as_mut_ptr(), ptr::null_mut()) };
        // (the run is over either way: the ticket's drop has nothing left to wait for)
        let (buf, root_status) = (t.buf.take().expect("a ticket is waited for once"), t.root_status.take().expect("a ticket is waited for once"));
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(Tables {
            n_roots: t.n_roots,
            n_vertices: t.n_vertices,
            repr: Repr::Packed { buf, layout: unsafe { layout.assume_init() }, root_status: root_status.into_vec() },
        })
    }

    /// `hspf_mask_words` + `hspf_run`: four arrays in host vectors.
    pub fn run_full(&self, g: &Graph<'_>, roots: &[u32], run_flags: u32) -> Result<Tables<'_>, Error> {
        let mut words = 0u32;
        let rc = unsafe { sys::hspf_mask_words(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, &mut words) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        let cells = roots.len() * g.n as usize;
        let (mut dist, mut hops, mut flags, mut mask) = (vec![0u32; cells], vec![0u16; cells], vec![0u16; cells], vec![0u64; cells * words as usize]);
        let mut out = sys::hspf_result {
            dist: dist.as_mut_ptr(),
            hops: hops.as_mut_ptr(),
            vflags_out: flags.as_mut_ptr(),
            first_hop_mask: mask.as_mut_ptr(),
            n_mask_words: words,
            pop_rank: ptr::null_mut(),
        };
        let rc = unsafe { sys::hspf_run(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, run_flags, &mut out) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(Tables { n_roots: roots.len() as u32, n_vertices: g.n, repr: Repr::Full { words, dist, hops, flags, mask } })
    }

    /// `hspf_run_device`: the tables stay in HBM (input of `routes_device` / `ancestors_device`).
    pub fn run_device(&self, g: &Graph<'_>, roots: &[u32], run_flags: u32) -> Result<DeviceTables<'_>, Error> {
        let mut words = 0u32;
        let rc = unsafe { sys::hspf_mask_words(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, &mut words) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        let cells = roots.len() * g.n as usize;
        let t = DeviceTables {
            n_roots: roots.len() as u32,
            n_vertices: g.n,
            words,
            dist: self.device_alloc(cells * 4)?,
            hops: self.device_alloc(cells * 2)?,
            flags: self.device_alloc(cells * 2)?,
            mask: self.device_alloc(cells * 8 * words as usize)?,
        };
        let mut out = sys::hspf_result {
            dist: t.dist.p as *mut u32,
            hops: t.hops.p as *mut u16,
            vflags_out: t.flags.p as *mut u16,
            first_hop_mask: t.mask.p as *mut u64,
            n_mask_words: words,
            pop_rank: ptr::null_mut(),
        };
        let rc = unsafe { sys::hspf_run_device(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, run_flags, &mut out) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(t)
    }

    /// `hspf_routes_device`: prefix attachment for every root of `run`, tables left in HBM.  `resident`: the table is the
    /// one the previous call on this engine passed (same vectors, untouched): nothing is uploaded again.
    pub fn routes_device(&self, run: &DeviceTables<'_>, table: &PrefixTable, resident: bool) -> Result<DeviceRoutes<'_>, Error> {
        let (r, p, w) = (run.n_roots as usize, table.n_prefixes() as usize, run.words as usize);
        let out = DeviceRoutes {
            n_roots: run.n_roots,
            n_prefixes: table.n_prefixes(),
            words: run.words,
            best_metric: self.device_alloc(r * p * 4)?,
            best_entry: self.device_alloc(r * p * 4)?,
            nexthop_mask: self.device_alloc(r * p * 8 * w)?,
        };
        if p == 0 {
            return Ok(out);
        }
        let t = sys::hspf_prefix_table {
            n_prefixes: table.n_prefixes(),
            n_entries: table.pfx_vertex.len() as u32,
            pfx_ptr: table.pfx_ptr.as_ptr(),
            pfx_vertex: table.pfx_vertex.as_ptr(),
            pfx_metric: table.pfx_metric.as_ptr(),
            flags: table.flags | if resident { sys::HSPF_PFX_RESIDENT } else { 0 },
            pfx_origin: ptr::null(),
            init_exists: ptr::null(),
            init_metric: ptr::null(),
            init_origin: ptr::null(),
        };
        let mut ro = out.raw();
        let rc = unsafe {
            sys::hspf_routes_device(self.ctx, run.n_vertices, run.n_roots, run.words, run.dist.p as *const u32, run.flags.p as *const u16, run.mask.p as *const u64, &t, &mut ro)
        };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(out)
    }

    /// A host table set brought to the device: the RIB held before, when no previous device set is comparable.
    pub fn routes_upload(&self, n_roots: u32, n_prefixes: u32, words: u32, best_metric: &[u32], best_entry: &[u32], nexthop_mask: &[u64]) -> Result<DeviceRoutes<'_>, Error> {
        Ok(DeviceRoutes {
            n_roots,
            n_prefixes,
            words,
            best_metric: self.device_from(best_metric)?,
            best_entry: self.device_from(best_entry)?,
            nexthop_mask: self.device_from(nexthop_mask)?,
        })
    }

    /// `hspf_routes_diff_device` + `hspf_routes_pack` (twice: the changed list packed from the new set and from the old
    /// one): the (root, prefix) pairs that need a RouteIpAdd / RouteIpDel, in the reference's emission order, with what the
    /// route is and what it was.  ONE comparison on the device, two small copies to the host.
    pub fn routes_changed(&self, old: &DeviceRoutes<'_>, new: &DeviceRoutes<'_>) -> Result<Vec<RouteRecord>, Error> {
        if old.n_roots != new.n_roots || old.n_prefixes != new.n_prefixes || old.words != new.words {
            return Err(Error { code: sys::HSPF_E_INVAL, detail: "routes_changed: the two sets differ in shape".into() });
        }
        let (r, p, w) = (new.n_roots as usize, new.n_prefixes as usize, new.words as usize);
        if r * p == 0 {
            return Ok(Vec::new());
        }
        let action = self.device_alloc(r * p)?;
        let changed = self.device_alloc(r * p * 4)?;
        let changed_ptr = self.device_alloc((r + 1) * 4)?;
        let (ro, rn) = (old.raw(), new.raw());
        let rc = unsafe {
            sys::hspf_routes_diff_device(self.ctx, new.n_roots, new.n_prefixes, new.words, &ro, &rn, action.p as *mut u8, changed.p as *mut u32, changed_ptr.p as *mut u32)
        };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        let k = unsafe { sys::hspf_routes_diff_count(self.ctx) } as usize;
        let stride = sys::HSPF_ROUTE_REC_WORDS as usize + 2 * w;
        let (mut rec_new, mut rec_old) = (vec![0u32; k * stride], vec![0u32; k * stride]);
        if k != 0 {
            for (set, rec) in [(&rn, &mut rec_new), (&ro, &mut rec_old)] {
                let rc = unsafe {
                    sys::hspf_routes_pack(self.ctx, new.n_roots, new.n_prefixes, new.words, set, action.p as *const u8, changed.p as *const u32, changed_ptr.p as *const u32, k as u32, rec.as_mut_ptr())
                };
                if rc != sys::HSPF_OK {
                    return Err(self.err(rc));
                }
            }
        }
        let words_of = |rec: &[u32]| -> Vec<u64> { (0..w).map(|i| rec[6 + 2 * i] as u64 | (rec[7 + 2 * i] as u64) << 32).collect() };
        Ok((0..k)
            .map(|i| {
                let (n, o) = (&rec_new[i * stride..(i + 1) * stride], &rec_old[i * stride..(i + 1) * stride]);
                RouteRecord {
                    root: n[0],
                    prefix: n[1],
                    action: n[2],
                    new_metric: n[3],
                    new_entry: n[4],
                    new_mask: words_of(n),
                    old_metric: o[3],
                    old_entry: o[4],
                    old_mask: words_of(o),
                }
            })
            .collect())
    }

    /// `hspf_ancestors_device`: for every root of a hop-count `run_device` and level L (1 = first hops, 2 = second hops) the
    /// bit sets that answer `Spt::is_on_path(a, d)` (holo-isis/src/spf.rs:261-286) for a level-L router `a` with one bit
    /// test: `AncestorSets::is_on_path`.  A root that needed the sequential kernel has no sets (`level_count` all ones):
    /// the caller keeps its own walk for it.
    pub fn ancestors_device(&self, g: &Graph<'_>, roots: &[u32], run_flags: u32, run: &DeviceTables<'_>, level: u32) -> Result<AncestorSets, Error> {
        let (r, n) = (roots.len(), g.n as usize);
        let level_rank = self.device_alloc(r * n * 4)?;
        let level_count = self.device_alloc(r * 4)?;
        let mut words = 1u32;
        loop {
            let anc = self.device_alloc(r * n * 8 * words as usize)?;
            let rc = unsafe {
                sys::hspf_ancestors_device(
                    self.ctx, g.g, roots.as_ptr(), r as u32, run_flags, run.dist.p as *const u32, run.hops.p as *const u16, run.flags.p as *const u16,
                    level, words, level_rank.p as *mut u32, level_count.p as *mut u32, anc.p as *mut u64,
                )
            };
            let count: Vec<u32> = level_count.to_host(r)?;
            if rc == sys::HSPF_E_TOO_MANY_SLOTS {
                let most = count.iter().copied().filter(|&c| c != u32::MAX).max().unwrap_or(0);
                words = most.div_ceil(64).max(words + 1);
                continue;
            }
            if rc != sys::HSPF_OK {
                return Err(self.err(rc));
            }
            return Ok(AncestorSets { n_vertices: g.n, words, level_rank: level_rank.to_host(r * n)?, level_count: count, anc: anc.to_host(r * n * words as usize)? });
        }
    }

    /// Pop ranks of roots whose pop order is dynamic (`HSPF_RF_EXACT`: zero-cost plateaus): `[root][vertex]`.
    pub fn pop_ranks(&self, g: &Graph<'_>, roots: &[u32], run_flags: u32) -> Result<Vec<u32>, Error> {
        let cells = roots.len() * g.n as usize;
        let (mut dist, mut rank) = (vec![0u32; cells], vec![0u32; cells]);
        let mut out = sys::hspf_result {
            dist: dist.as_mut_ptr(),
            hops: ptr::null_mut(),
            vflags_out: ptr::null_mut(),
            first_hop_mask: ptr::null_mut(),
            n_mask_words: 0,
            pop_rank: rank.as_mut_ptr(),
        };
        let flags = run_flags | sys::HSPF_RUN_POP_RANK;
        let rc = unsafe { sys::hspf_run(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, flags, &mut out) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(rank)
    }

    /// Several runs in flight from this one thread (one per area / level / neighbour set): device buffers, a ticket.
    ///
    /// # Safety
    /// `out_device` holds DEVICE pointers sized for `roots.len()` rows; they must stay valid, and unshared with other
    /// runs in flight, until `wait(ticket)` has returned.
    pub unsafe fn run_device_async(
        &self,
        g: &Graph<'_>,
        roots: &[u32],
        run_flags: u32,
        out_device: &sys::hspf_result,
    ) -> Result<u64, Error> {
        let mut ticket = 0u64;
        let rc = sys::hspf_run_device_async(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, run_flags, out_device, &mut ticket);
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(ticket)
    }

    pub fn wait(&self, ticket: u64) -> Result<sys::hspf_stats, Error> {
        let mut st = std::mem::MaybeUninit::<sys::hspf_stats>::zeroed();
        let rc = unsafe { sys::hspf_wait(self.ctx, ticket, st.as_mut_ptr()) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(unsafe { st.assume_init() })
    }

    // ---- several areas, ONE RIB, on the device (ABI 7: hspf_rib_clear_device / hspf_rib_fold_device) ---------------------

    /// The empty instance-wide RIB state in front of the first area (`hspf_rib_clear_device`).
    pub fn rib_new(&self, n_prefixes: u32, words: u32) -> Result<RibDevice<'_>, Error> {
        let (p, w) = (n_prefixes as usize, words as usize);
        let rib = RibDevice {
            n_prefixes,
            words,
            best_metric: self.device_alloc(p.max(1) * 4)?,
            best_entry: self.device_alloc(p.max(1) * 4)?,
            nexthop_mask: self.device_alloc((p * w).max(1) * 8)?,
            origin: self.device_alloc(p.max(1) * 4)?,
        };
        let rc = unsafe { sys::hspf_rib_clear_device(self.ctx, &rib.raw()) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(rib)
    }

    /// `hspf_rib_fold_device`: the literal ordered fold of ONE area's table (`HSPF_PFX_ORDERED`: entries in
    /// `V::intra_area_networks` order with their owners' LS-IDs) into the state the earlier areas left — `update_rib_intra_area`
    /// (holo-ospf/src/route.rs:343-448) for the whole area in one call.  `run` = the area's one-root `run_device`;
    /// `prefix_map[i]` = instance-wide index of the area's prefix i; `word_offset` = where the area's first-hop slots start
    /// in the instance-wide numbering.
    #[allow(clippy::too_many_arguments)]
    pub fn rib_fold(&self, rib: &RibDevice<'_>, run: &DeviceTables<'_>, table: &OrderedPrefixTable, prefix_map: &[u32], area_index: u32, word_offset: u32) -> Result<(), Error> {
        if run.n_roots != 1 || prefix_map.len() != table.n_prefixes() as usize || word_offset + run.words > rib.words {
            return Err(Error { code: sys::HSPF_E_INVAL, detail: "rib_fold: one root per area; one map entry per prefix; the area's mask words must fit".into() });
        }
        let t = sys::hspf_prefix_table {
            n_prefixes: table.n_prefixes(),
            n_entries: table.pfx_vertex.len() as u32,
            pfx_ptr: table.pfx_ptr.as_ptr(),
            pfx_vertex: table.pfx_vertex.as_ptr(),
            pfx_metric: table.pfx_metric.as_ptr(),
            flags: table.flags | sys::HSPF_PFX_ORDERED,
            pfx_origin: table.pfx_origin.as_ptr(),
            init_exists: ptr::null(),
            init_metric: ptr::null(),
            init_origin: ptr::null(),
        };
        let rc = unsafe {
            sys::hspf_rib_fold_device(
                self.ctx, run.n_vertices, run.words, run.dist.p as *const u32, run.flags.p as *const u16, run.mask.p as *const u64, &t, prefix_map.as_ptr(), area_index, word_offset, &rib.raw(),
            )
        };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(())
    }

    /// `hspf_run_packed_device`: the packed words of every (root, vertex) left in HBM (a quarter of `run_device`'s bytes, for
    /// consumers that stay on the GPU or send the rows on with `Multi::allgather_rows`); returns the buffer, the layout of the
    /// words and the per-root status (`sys::HSPF_ROOT_EXACT`: the root's pop order is dynamic).
    pub fn run_packed_device(&self, g: &Graph<'_>, roots: &[u32], run_flags: u32) -> Result<(DeviceBuf<'_>, sys::hspf_packed_layout, Vec<u8>), Error> {
        let cap = roots.len() * g.n as usize * 8;
        let buf = self.device_alloc(cap.max(8))?;
        let mut layout = std::mem::MaybeUninit::<sys::hspf_packed_layout>::zeroed();
        let mut status = vec![0u8; roots.len()];
        let rc = unsafe { sys::hspf_run_packed_device(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, run_flags, buf.p, cap, layout.as_mut_ptr(), status.as_mut_ptr()) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok((buf, unsafe { layout.assume_init() }, status))
    }

    /// Statistics of the last run on this engine (`hspf_get_stats`): which kernels ran, `n_exact_roots`, `n_repaired_roots`.
    pub fn stats(&self) -> sys::hspf_stats {
        let mut st = std::mem::MaybeUninit::<sys::hspf_stats>::zeroed();
        unsafe {
            sys::hspf_get_stats(self.ctx, st.as_mut_ptr());
            st.assume_init()
        }
    }

    /// Every run handed to a lane is over (`hspf_wait_all`); the number of lanes (`hspf_async_lanes`).
    pub fn wait_all(&self) -> Result<(), Error> {
        let rc = unsafe { sys::hspf_wait_all(self.ctx) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(())
    }
    pub fn async_lanes(&self) -> u32 {
        unsafe { sys::hspf_async_lanes(self.ctx) }
    }
    pub fn device_count() -> i32 {
        unsafe { sys::hspf_device_count() }
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        for (_, free) in std::mem::take(&mut *self.pool.borrow_mut()) {
            for p in free {
                unsafe { sys::hspf_device_free(self.ctx, p) }
            }
        }
        unsafe { sys::hspf_shutdown(self.ctx) }
    }
}

impl Graph<'_> {
    /// The caller's CSR as it is resident on the device (`hspf_graph_export`: ROW_PTR / COL / METRIC / VFLAGS) — what
    /// `Engine::upload_keyed` built from the LSDB records, for `CsrCache`'s host mirror.
    pub fn export_csr(&self, max_path_metric: u32) -> Result<Csr, Error> {
        fn fetch<T: Clone + Default>(g: &Graph<'_>, which: u32) -> Result<Vec<T>, Error> {
            let mut bytes = 0usize;
            let rc = unsafe { sys::hspf_graph_export(g.eng.ctx, g.g, which, ptr::null_mut(), 0, &mut bytes) };
            if rc != sys::HSPF_OK && bytes == 0 {
                return Err(g.eng.err(rc));
            }
            let mut v = vec![T::default(); bytes / std::mem::size_of::<T>()];
            let rc = unsafe { sys::hspf_graph_export(g.eng.ctx, g.g, which, v.as_mut_ptr() as *mut _, bytes, &mut bytes) };
            if rc != sys::HSPF_OK {
                return Err(g.eng.err(rc));
            }
            Ok(v)
        }
        Ok(Csr {
            row_ptr: fetch::<u32>(self, sys::HSPF_GX_ROW_PTR)?,
            col: fetch::<u32>(self, sys::HSPF_GX_COL)?,
            metric: fetch::<u32>(self, sys::HSPF_GX_METRIC)?,
            vflags: fetch::<u8>(self, sys::HSPF_GX_VFLAGS)?,
            max_path_metric,
        })
    }

    /// Whole rows replaced (the rows of the LSPs / LSAs that triggered the run: `trigger_lsps`, `SpfTriggerLsa`).
    pub fn patch(&mut self, rows: &[RowPatch]) -> Result<(), Error> {
        let vertex: Vec<u32> = rows.iter().map(|r| r.vertex).collect();
        let vflags: Vec<u8> = rows.iter().map(|r| r.vflags).collect();
        let mut row_ptr = Vec::with_capacity(rows.len() + 1);
        let (mut col, mut metric) = (Vec::new(), Vec::new());
        row_ptr.push(0u32);
        for r in rows {
            col.extend_from_slice(&r.col);
            metric.extend_from_slice(&r.metric);
            row_ptr.push(col.len() as u32);
        }
        let p = sys::hspf_rows {
            n_changed: rows.len() as u32,
            vertex: vertex.as_ptr(),
            row_ptr: row_ptr.as_ptr(),
            col: col.as_ptr(),
            metric: metric.as_ptr(),
            vflags: vflags.as_ptr(),
        };
        let rc = unsafe { sys::hspf_graph_patch(self.eng.ctx, self.g, &p) };
        if rc != sys::HSPF_OK {
            return Err(self.eng.err(rc));
        }
        Ok(())
    }

    /// `(vertex, slot base)` of every vertex of H = {root, networks attached to it}: slot `base + j` is the j-th link
    /// of that vertex's row (include/holo_spf_hip.h "first-hop slots").
    pub fn slot_table(&self, root: u32) -> Result<Vec<(u32, u32)>, Error> {
        let mut total = 0u32;
        let cnt = unsafe { sys::hspf_slot_table(self.eng.ctx, self.g, root, ptr::null_mut(), ptr::null_mut(), 0, &mut total) };
        if cnt < 0 {
            return Err(self.eng.err(cnt));
        }
        let (mut hv, mut hb) = (vec![0u32; cnt as usize], vec![0u32; cnt as usize]);
        let rc = unsafe { sys::hspf_slot_table(self.eng.ctx, self.g, root, hv.as_mut_ptr(), hb.as_mut_ptr(), cnt as u32, &mut total) };
        if rc < 0 {
            return Err(self.eng.err(rc));
        }
        Ok(hv.into_iter().zip(hb).collect())
    }
}

impl Drop for Graph<'_> {
    fn drop(&mut self) {
        unsafe { sys::hspf_graph_free(self.eng.ctx, self.g) }
    }
}

/// The resident CSR of one (area | level x topology x metric mode) and what it was built from: `get_or_patch` turns
/// the rows derived from the CURRENT LSDB into either nothing (unchanged), a row patch (`hspf_graph_patch`: same
/// vertex set, some rows differ — what `trigger_lsps` / `SpfTriggerLsa` describe) or a fresh upload (a vertex appeared
/// or vanished).  `K` = the protocol's vertex id (`VertexId`), in the reference's order.
pub struct CsrCache<'e, K: Ord + Clone> {
    pub keys: Vec<K>,
    pub csr: Csr,
    pub graph: Option<Graph<'e>>,
}

impl<'e, K: Ord + Clone> Default for CsrCache<'e, K> {
    fn default() -> Self {
        CsrCache { keys: Vec::new(), csr: Csr::default(), graph: None }
    }
}

impl<'e, K: Ord + Clone> CsrCache<'e, K> {
    pub fn index_of(&self, key: &K) -> Option<u32> {
        self.keys.binary_search(key).ok().map(|i| i as u32)
    }

    /// `keys` ascending, `fresh` = the CSR derived from the LSDB as it is now.
    pub fn get_or_patch(&mut self, eng: &'e Engine, keys: Vec<K>, fresh: Csr) -> Result<&Graph<'e>, Error> {
        let same_shape = self.graph.is_some() && self.keys == keys && self.csr.max_path_metric == fresh.max_path_metric;
        if same_shape {
            let mut rows = Vec::new();
            for v in 0..fresh.n_vertices() {
                let (nc, nm) = fresh.row(v);
                let (oc, om) = self.csr.row(v);
                if nc != oc || nm != om || fresh.vflags[v as usize] != self.csr.vflags[v as usize] {
                    rows.push(RowPatch { vertex: v, col: nc.to_vec(), metric: nm.to_vec(), vflags: fresh.vflags[v as usize] });
                }
            }
            if !rows.is_empty() {
                if let Err(e) = self.graph.as_mut().unwrap().patch(&rows) {
                    // After a failed patch the library's graph is invalid (device arrays and host mirrors may be out of
                    // step; every later call on it returns HSPF_E_INVAL) and `self.csr` no longer describes it: drop the
                    // replica, so that the next call uploads afresh instead of diffing against a stale CSR.  The error
                    // still goes up: the caller falls back to its CPU loop for this run.
                    self.graph = None;
                    self.keys.clear();
                    self.csr = Csr::default();
                    return Err(e);
                }
                self.csr = fresh;
            }
        } else {
            self.graph = None; // frees the old replica first
            self.graph = Some(eng.upload(&fresh)?);
            self.keys = keys;
            self.csr = fresh;
        }
        Ok(self.graph.as_ref().unwrap())
    }
}

/// `hspf_prefix_table` of one OSPF area with `HSPF_PFX_ORDERED`: the entries of a prefix in the order
/// `V::intra_area_networks` yields them, `pfx_origin` = Link State ID of the entry's vertex LSA (the transit-network rule of
/// holo-ospf/src/route.rs:388-400 compares it); network-LSA entries carry `HSPF_PFX_ENTRY_NETWORK` in `pfx_vertex`.
#[derive(Debug, Default, Clone)]
pub struct OrderedPrefixTable {
    pub pfx_ptr: Vec<u32>,
    pub pfx_vertex: Vec<u32>,
    pub pfx_metric: Vec<u32>,
    pub pfx_origin: Vec<u32>,
    pub flags: u32, // HSPF_PFX_SATURATING for OSPF (u32 saturating add, spf.rs:672)
}

impl OrderedPrefixTable {
    pub fn n_prefixes(&self) -> u32 {
        self.pfx_ptr.len().saturating_sub(1) as u32
    }
}

/// The instance-wide RIB state of one OSPF Full computation on the device (`hspf_rib_device`): after the last area's fold
/// `(best_metric, best_entry, nexthop_mask)` IS an `hspf_routes` of one root — `into_routes` hands it to `routes_changed`.
pub struct RibDevice<'e> {
    pub n_prefixes: u32,
    pub words: u32,
    best_metric: DeviceBuf<'e>,
    best_entry: DeviceBuf<'e>,
    nexthop_mask: DeviceBuf<'e>,
    origin: DeviceBuf<'e>,
}

impl<'e> RibDevice<'e> {
    fn raw(&self) -> sys::hspf_rib_device {
        sys::hspf_rib_device {
            n_prefixes: self.n_prefixes,
            n_mask_words: self.words,
            best_metric: self.best_metric.p as *mut u32,
            best_entry: self.best_entry.p as *mut u32,
            nexthop_mask: self.nexthop_mask.p as *mut u64,
            origin: self.origin.p as *mut u32,
        }
    }
    /// Metric, owner (`area_index << 24 | entry`; all ones: no route) and next-hop mask of every prefix, on the host.
    pub fn to_host(&self) -> Result<(Vec<u32>, Vec<u32>, Vec<u64>), Error> {
        let (p, w) = (self.n_prefixes as usize, self.words as usize);
        Ok((self.best_metric.to_host(p)?, self.best_entry.to_host(p)?, self.nexthop_mask.to_host(p * w)?))
    }
    pub fn into_routes(self) -> DeviceRoutes<'e> {
        DeviceRoutes { n_roots: 1, n_prefixes: self.n_prefixes, words: self.words, best_metric: self.best_metric, best_entry: self.best_entry, nexthop_mask: self.nexthop_mask }
    }
}

// ---- several GPUs (include/holo_spf_hip.h "several GPUs") -----------------------------------------------------
// Roots are independent units over a replicated graph: a `Multi` drives the devices of this process (one engine context
// each) or is ONE rank of a job of processes (RCCL: `unique_id` from rank 0, carried by the host's own transport — holo's
// ibus).  Tables are DEVICE buffers the caller owns, one `hspf_result` per local device, sized for ALL roots.

pub struct Multi {
    m: *mut sys::hspf_multi,
}

pub struct MultiGraph<'m> {
    multi: &'m Multi,
    g: *mut sys::hspf_multi_graph,
    pub n: u32,
}

impl Multi {
    fn err(&self, code: i32) -> Error {
        let detail = unsafe { CStr::from_ptr(sys::hspf_multi_last_error(self.m)) }.to_string_lossy().into_owned();
        Error { code, detail }
    }

    /// Rank 0 of a job of processes: the communicator id every rank passes to `new` (`hspf_multi_unique_id`; needs librccl.so).
    pub fn unique_id() -> Result<[u8; sys::HSPF_COMM_ID_BYTES as usize], Error> {
        let mut id = [0u8; sys::HSPF_COMM_ID_BYTES as usize];
        let rc = unsafe { sys::hspf_multi_unique_id(id.as_mut_ptr()) };
        if rc != sys::HSPF_OK {
            return Err(Error { code: rc, detail: "hspf_multi_unique_id (librccl.so)".into() });
        }
        Ok(id)
    }

    /// `unique_id == None`: one process drives every device (`world == device_ordinals.len()`, peer copies over xGMI);
    /// `Some(id)`: this process is the ranks `first_rank ..` of a job of `world` ranks (RCCL).
    pub fn new(device_ordinals: &[i32], world: u32, first_rank: u32, unique_id: Option<&[u8; sys::HSPF_COMM_ID_BYTES as usize]>) -> Result<Self, Error> {
        let cfg = sys::hspf_multi_config {
            n_local: device_ordinals.len() as u32,
            device_ordinals: device_ordinals.as_ptr(),
            world,
            first_rank,
            unique_id: unique_id.map_or(ptr::null(), |id| id.as_ptr()),
        };
        let mut m = ptr::null_mut();
        let rc = unsafe { sys::hspf_multi_init(&cfg, &mut m) };
        if rc != sys::HSPF_OK {
            let detail = unsafe { CStr::from_ptr(sys::hspf_multi_init_error()) }.to_string_lossy().into_owned();
            return Err(Error { code: rc, detail });
        }
        Ok(Multi { m })
    }

    pub fn n_local(&self) -> u32 {
        unsafe { sys::hspf_multi_n_local(self.m) }
    }

    /// A replica of the graph on every local device.
    pub fn upload(&self, csr: &Csr) -> Result<MultiGraph<'_>, Error> {
        let c = sys::hspf_csr {
            n_vertices: csr.n_vertices(),
            n_edges: csr.col.len() as u32,
            row_ptr: csr.row_ptr.as_ptr(),
            col: csr.col.as_ptr(),
            metric: csr.metric.as_ptr(),
            vflags: csr.vflags.as_ptr(),
            max_path_metric: csr.max_path_metric,
        };
        let mut g = ptr::null_mut();
        let rc = unsafe { sys::hspf_multi_graph_upload(self.m, &c, &mut g) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(MultiGraph { multi: self, g, n: csr.n_vertices() })
    }

    /// `[begin, end)` of `rank` in a list of `n_roots` roots: whole 64-root batches (`hspf_shard_bounds`).
    pub fn shard_bounds(n_roots: u32, world: u32, rank: u32) -> (u32, u32) {
        let (mut b, mut e) = (0u32, 0u32);
        unsafe { sys::hspf_shard_bounds(n_roots, world, rank, &mut b, &mut e) };
        (b, e)
    }

    /// Areas first, then roots (multi-area OSPF, BASELINE configs[3]): the (rank, area, root range) slices of `hspf_plan_areas`.
    pub fn plan_areas(roots_per_area: &[u32], world: u32) -> Vec<sys::hspf_area_slice> {
        let cap = roots_per_area.len() as u32 + world;
        let mut out = vec![sys::hspf_area_slice { rank: 0, area: 0, root_begin: 0, root_end: 0 }; cap as usize];
        let k = unsafe { sys::hspf_plan_areas(roots_per_area.len() as u32, roots_per_area.as_ptr(), world, out.as_mut_ptr(), cap) };
        out.truncate(k.min(cap) as usize);
        out
    }

    pub fn mask_words(&self, g: &MultiGraph<'_>, roots: &[u32]) -> Result<u32, Error> {
        let mut w = 0u32;
        let rc = unsafe { sys::hspf_multi_mask_words(self.m, g.g, roots.as_ptr(), roots.len() as u32, &mut w) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(w)
    }

    /// Sharded run + the all-gather of the tables selected in `gather` (the `HSPF_GATHER_*` bits of `sys`).
    ///
    /// # Safety
    /// `all[i]` holds DEVICE pointers on local device i sized for ALL `roots.len()` rows; valid until the call (or, with
    /// `HSPF_GATHER_ASYNC`, `wait`) has returned.
    pub unsafe fn run(&self, g: &MultiGraph<'_>, roots: &[u32], run_flags: u32, all: &mut [sys::hspf_result], gather: u32) -> Result<(), Error> {
        if all.len() != self.n_local() as usize {
            return Err(Error { code: sys::HSPF_E_INVAL, detail: "Multi::run: one hspf_result per local device".into() });
        }
        let rc = sys::hspf_multi_run(self.m, g.g, roots.as_ptr(), roots.len() as u32, run_flags, all.as_mut_ptr(), gather);
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(())
    }

    /// The two halves of `run` (several steps in flight, into DIFFERENT tables; wait in ticket order on every rank).
    ///
    /// # Safety
    /// As `run`; the tables of a ticket must not be shared with another ticket in flight.
    pub unsafe fn run_async(&self, g: &MultiGraph<'_>, roots: &[u32], run_flags: u32, all: &[sys::hspf_result]) -> Result<u64, Error> {
        let mut ticket = 0u64;
        let rc = sys::hspf_multi_run_async(self.m, g.g, roots.as_ptr(), roots.len() as u32, run_flags, all.as_ptr(), &mut ticket);
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(ticket)
    }
    /// # Safety
    /// `all` = the tables the ticket was started with.
    pub unsafe fn run_wait(&self, ticket: u64, all: &mut [sys::hspf_result], gather: u32) -> Result<(), Error> {
        let rc = sys::hspf_multi_run_wait(self.m, ticket, all.as_mut_ptr(), gather);
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(())
    }

    /// Pending asynchronous gathers are over (`hspf_multi_wait`).
    pub fn wait(&self) -> Result<(), Error> {
        let rc = unsafe { sys::hspf_multi_wait(self.m) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(())
    }

    /// The collective on its own, for any per-root table (route tables after `routes_device`: "a single all-gather of
    /// per-root route tables").
    ///
    /// # Safety
    /// `tables[i]`: a device buffer on local device i of `n_roots` rows of `row_bytes` bytes whose own rows are filled.
    pub unsafe fn allgather_rows(&self, tables: &[*mut c_void], row_bytes: usize, n_roots: u32) -> Result<(), Error> {
        let rc = sys::hspf_multi_allgather_rows(self.m, tables.as_ptr(), row_bytes, n_roots);
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(())
    }

    pub fn stats(&self, local_index: u32) -> Result<sys::hspf_stats, Error> {
        let mut st = std::mem::MaybeUninit::<sys::hspf_stats>::zeroed();
        let rc = unsafe { sys::hspf_multi_get_stats(self.m, local_index, st.as_mut_ptr()) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(unsafe { st.assume_init() })
    }

    /// The context of local device `i`, for the one-device calls (`hspf_device_alloc`, `hspf_routes_device` ...) on that rank.
    pub fn ctx_raw(&self, local_index: u32) -> *mut sys::hspf_ctx {
        unsafe { sys::hspf_multi_ctx(self.m, local_index) }
    }
}

impl Drop for Multi {
    fn drop(&mut self) {
        unsafe { sys::hspf_multi_shutdown(self.m) }
    }
}

impl MultiGraph<'_> {
    /// Whole rows replaced on every replica (`hspf_multi_graph_patch`).
    pub fn patch(&mut self, rows: &[RowPatch]) -> Result<(), Error> {
        let vertex: Vec<u32> = rows.iter().map(|r| r.vertex).collect();
        let vflags: Vec<u8> = rows.iter().map(|r| r.vflags).collect();
        let mut row_ptr = vec![0u32];
        let (mut col, mut metric) = (Vec::new(), Vec::new());
        for r in rows {
            col.extend_from_slice(&r.col);
            metric.extend_from_slice(&r.metric);
            row_ptr.push(col.len() as u32);
        }
        let p = sys::hspf_rows { n_changed: rows.len() as u32, vertex: vertex.as_ptr(), row_ptr: row_ptr.as_ptr(), col: col.as_ptr(), metric: metric.as_ptr(), vflags: vflags.as_ptr() };
        let rc = unsafe { sys::hspf_multi_graph_patch(self.multi.m, self.g, &p) };
        if rc != sys::HSPF_OK {
            return Err(self.multi.err(rc));
        }
        Ok(())
    }
    /// The replica on local device `i` (`hspf_multi_graph_local`), for the one-device calls on that rank's context.
    pub fn local_raw(&self, local_index: u32) -> *mut sys::hspf_graph {
        unsafe { sys::hspf_multi_graph_local(self.g, local_index) }
    }
}

impl Drop for MultiGraph<'_> {
    fn drop(&mut self) {
        unsafe { sys::hspf_multi_graph_free(self.multi.m, self.g) }
    }
}
