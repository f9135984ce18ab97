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

use std::ffi::CStr;
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

/// One replaced row of `Graph::patch`.
#[derive(Debug, Clone)]
pub struct RowPatch {
    pub vertex: u32,
    pub col: Vec<u32>,
    pub metric: Vec<u32>,
    pub vflags: u8,
}

/// Per (root, vertex) results of a run, row-major `[root][vertex]` (`hspf_result`).
#[derive(Debug, Default)]
pub struct Tables {
    pub n_roots: u32,
    pub n_vertices: u32,
    pub words: u32,
    pub dist: Vec<u32>,
    pub hops: Vec<u16>,
    pub flags: Vec<u16>,
    pub mask: Vec<u64>,
}

impl Tables {
    #[inline]
    fn at(&self, root: u32, v: u32) -> usize {
        root as usize * self.n_vertices as usize + v as usize
    }
    pub fn in_spt(&self, root: u32, v: u32) -> bool {
        self.flags[self.at(root, v)] & sys::HSPF_RF_IN_SPT as u16 != 0
    }
    pub fn exact(&self, root: u32, v: u32) -> bool {
        self.flags[self.at(root, v)] & sys::HSPF_RF_EXACT as u16 != 0
    }
    pub fn dist(&self, root: u32, v: u32) -> u32 {
        self.dist[self.at(root, v)]
    }
    pub fn hops(&self, root: u32, v: u32) -> u16 {
        self.hops[self.at(root, v)]
    }
    /// First-hop slots of (root, v), ascending.
    pub fn slots(&self, root: u32, v: u32) -> impl Iterator<Item = u32> + '_ {
        let base = self.at(root, v) * self.words as usize;
        (0..self.words as usize).flat_map(move |w| {
            let mut m = self.mask[base + w];
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

pub struct Engine {
    ctx: *mut sys::hspf_ctx,
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
        Ok(Engine { ctx })
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

    /// `hspf_mask_words` + `hspf_run`: results in host vectors.  `roots` may hold `HSPF_NO_ROOT` padding.
    pub fn run(&self, g: &Graph<'_>, roots: &[u32], run_flags: u32) -> Result<Tables, Error> {
        let mut words = 0u32;
        let rc = unsafe { sys::hspf_mask_words(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, &mut words) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        let cells = roots.len() * g.n as usize;
        let mut t = Tables {
            n_roots: roots.len() as u32,
            n_vertices: g.n,
            words,
            dist: vec![0; cells],
            hops: vec![0; cells],
            flags: vec![0; cells],
            mask: vec![0; cells * words as usize],
        };
        let mut out = sys::hspf_result {
            dist: t.dist.as_mut_ptr(),
            hops: t.hops.as_mut_ptr(),
            vflags_out: t.flags.as_mut_ptr(),
            first_hop_mask: t.mask.as_mut_ptr(),
            n_mask_words: words,
            pop_rank: ptr::null_mut(),
        };
        let rc = unsafe { sys::hspf_run(self.ctx, g.g, roots.as_ptr(), roots.len() as u32, run_flags, &mut out) };
        if rc != sys::HSPF_OK {
            return Err(self.err(rc));
        }
        Ok(t)
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
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { sys::hspf_shutdown(self.ctx) }
    }
}

impl Graph<'_> {
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
                self.graph.as_mut().unwrap().patch(&rows)?;
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
