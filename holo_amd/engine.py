"""Thin Python wrapper over the C ABI of libholo_spf_hip.so (include/holo_spf_hip.h).

This is the host-side handle layer a caller (the Python mirrors of holo-ospf `run_area` /
holo-isis `compute_spt` in holo_amd.isis / holo_amd.ospf, bench.py, tests) uses.  All compute is
in the HIP library; there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes
import time
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib as L

# run flags (mirror of HSPF_RUN_*)
RUN_NET_NEXTHOPS = 0x01
RUN_IGNORE_OVERLOAD = 0x02
RUN_FORCE_EXACT = 0x04
RUN_POP_RANK = 0x08
RUN_COUNT_ROWS = 0x10
E_TOO_MANY_SLOTS = -5
E_NO_PACKED = -7
ROOT_EXACT = 0x01

PFX_SATURATING = 0x1
PFX_LAST_MIN = 0x2
PFX_ORDERED = 0x4
PFX_RESIDENT = 0x8        # the three main arrays are what the previous routes_device call on this context passed
PFX_ENTRY_NETWORK = 0x80000000
PFX_KEPT_INIT = 0xFFFFFFFE
DIFF_SAME, DIFF_INSTALL, DIFF_WITHDRAW, DIFF_SILENT = 0, 1, 2, 3

RF_IN_SPT = 0x0001
RF_EXACT = 0x0002
DIST_INF = 0xFFFFFFFF
NO_ROOT = 0xFFFFFFFF


class HspfError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        super().__init__(f"{what}: {code} ({L.load().hspf_strerror(code).decode()})"
                         + (f": {detail}" if detail else ""))


@dataclass
class SpfResult:
    """Row-major per-root results on the host (numpy)."""
    dist: np.ndarray                 # [R, N] u32, DIST_INF when not in SPT
    hops: np.ndarray                 # [R, N] u16
    flags: np.ndarray                # [R, N] u16 (RF_*)
    first_hop_mask: np.ndarray       # [R, N, W] u64
    pop_rank: Optional[np.ndarray]   # [R, N] u32 or None
    stats: dict


@dataclass
class PackedResult:
    """hspf_run_packed(): ONE machine word per (root, vertex), row-major, plus the field positions of the run
    (include/holo_spf_hip.h "packed results").  The accessors decode whole tables with numpy; a caller that looks at a
    vertex once decodes only what it touches."""
    words: np.ndarray                # [R, N] u32 or u64 (a view of the caller's / the wrapper's buffer)
    word_bytes: int
    dist_shift: int
    hops_shift: int
    hops_mask: int
    mask_bits: int
    not_reached: int
    root_status: np.ndarray          # [R] u8, ROOT_EXACT
    stats: dict

    @property
    def in_spt(self) -> np.ndarray:
        return self.words < self.words.dtype.type(self.not_reached)     # (4-byte words: not_reached fits 32 bits)

    @property
    def dist(self) -> np.ndarray:
        d = (self.words >> self.words.dtype.type(self.dist_shift)).astype(np.uint32)
        return np.where(self.in_spt, d, np.uint32(DIST_INF))

    @property
    def hops(self) -> np.ndarray:
        h = ((self.words >> self.words.dtype.type(self.hops_shift)) & self.words.dtype.type(self.hops_mask)).astype(np.uint16)
        return np.where(self.in_spt, h, np.uint16(0))

    @property
    def first_hop_mask(self) -> np.ndarray:
        m = (self.words & self.words.dtype.type((1 << self.mask_bits) - 1)).astype(np.uint64)
        return np.where(self.in_spt, m, np.uint64(0))[..., None]


class PinnedBuffer:
    """Page-locked host memory from hspf_host_alloc (freed with the object)."""

    def __init__(self, ctx: "SpfContext", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = ctypes.c_void_p()
        rc = ctx.lib.hspf_host_alloc(ctx.handle, self.nbytes, ctypes.byref(p))
        if rc != 0:
            raise HspfError(rc, "hspf_host_alloc", ctx.last_error())
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * self.nbytes).from_address(self.ptr))

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.ctx.lib.hspf_host_free(self.ctx.handle, ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            if getattr(self.ctx, "handle", None):
                self.free()
        except Exception:
            pass


def _u32(a):
    return a.ctypes.data_as(L.u32p)


def splice_rows(row_ptr, col, metric, vflags, vertices, cols, mets, new_flags):
    """CSR with the rows of `vertices` (ascending) replaced — the host-side twin of hspf_graph_patch."""
    n = len(row_ptr) - 1
    lens = np.diff(row_ptr.astype(np.int64))
    for v, c in zip(vertices, cols):
        lens[int(v)] = len(c)
    nrp = np.zeros(n + 1, np.int64)
    nrp[1:] = np.cumsum(lens)
    ncol = np.empty(int(nrp[-1]), np.uint32)
    nmet = np.empty(int(nrp[-1]), np.uint32)
    prev = 0
    for v, c, m in list(zip(vertices, cols, mets)) + [(n, None, None)]:
        v = int(v)
        a, b = int(row_ptr[prev]), int(row_ptr[v])                  # unchanged rows [prev, v)
        ncol[nrp[prev]:nrp[prev] + (b - a)] = col[a:b]
        nmet[nrp[prev]:nrp[prev] + (b - a)] = metric[a:b]
        if c is not None:
            ncol[nrp[v]:nrp[v + 1]] = c
            nmet[nrp[v]:nrp[v + 1]] = m
        prev = v + 1
    nvf = vflags.copy()
    nvf[np.asarray(vertices, dtype=np.int64)] = new_flags
    return nrp.astype(np.uint32), ncol, nmet, nvf


class SpfGraph:
    """Device-resident graph of one LSDB generation (hspf_graph)."""

    def __init__(self, ctx: "SpfContext", row_ptr, col, metric, vflags, max_path_metric: int):
        self.ctx = ctx
        # numpy mirrors of the caller's CSR (tests and tools read them back; the library keeps its own): patches are
        # recorded and spliced in when a mirror is next READ (row_ptr / col / metric / vflags are properties)
        self._rp = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        self._col = np.ascontiguousarray(col, dtype=np.uint32)
        self._met = np.ascontiguousarray(metric, dtype=np.uint32)
        self._vf = np.ascontiguousarray(vflags, dtype=np.uint8)
        self._pending = []
        self._own_mirrors = False
        self.n = len(self._rp) - 1
        csr = L.HspfCsr(self.n, len(self._col), _u32(self._rp), _u32(self._col), _u32(self._met),
                        self._vf.ctypes.data_as(L.u8p), ctypes.c_uint32(max_path_metric))
        h = ctypes.c_void_p()
        rc = ctx.lib.hspf_graph_upload(ctx.handle, ctypes.byref(csr), ctypes.byref(h))
        if rc != 0:
            raise HspfError(rc, "hspf_graph_upload", ctx.last_error())
        self.handle = h

    @classmethod
    def from_keys(cls, ctx: "SpfContext", vertex_key, row_ptr, target_key, metric, vflags, max_path_metric: int):
        """hspf_graph_upload_keyed(): vertices by 64-bit key in any order, links as (target key, cost), targets unresolved — the
        device ranks the keys, resolves the targets, drops links to absent vertices and builds the graph.  Returns
        (graph, rank): rank[i] = index of input vertex i.  The numpy mirrors are what the device built (read back)."""
        vk = np.ascontiguousarray(vertex_key, dtype=np.uint64)
        rp = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        tk = np.ascontiguousarray(target_key, dtype=np.uint64)
        mt = np.ascontiguousarray(metric, dtype=np.uint32)
        vf = np.ascontiguousarray(vflags, dtype=np.uint8)
        rank = np.empty(len(vk), np.uint32)
        k = L.HspfKeyedLsdb(len(vk), len(tk), vk.ctypes.data_as(L.u64p), _u32(rp), tk.ctypes.data_as(L.u64p), _u32(mt),
                            vf.ctypes.data_as(L.u8p), ctypes.c_uint32(max_path_metric))
        h = ctypes.c_void_p()
        rc = ctx.lib.hspf_graph_upload_keyed(ctx.handle, ctypes.byref(k), ctypes.byref(h), _u32(rank))
        if rc != 0:
            raise HspfError(rc, "hspf_graph_upload_keyed", ctx.last_error())
        self = cls.__new__(cls)
        self.ctx, self.handle, self.n = ctx, h, len(vk)
        self._pending, self._own_mirrors = [], True
        self._rp, self._col, self._met, self._vf = (self.export(x) for x in ("row_ptr", "col", "metric", "vflags"))
        return self, rank

    def _flush(self) -> None:
        for vs, cols, mets, nf in self._pending:
            lens = self._rp[vs.astype(np.int64) + 1] - self._rp[vs]
            if np.array_equal(lens, [len(c) for c in cols]):       # same row lengths: the mirrors change in place
                if not self._own_mirrors:                           # the caller's arrays until now: never written through
                    self._col, self._met, self._vf = self._col.copy(), self._met.copy(), self._vf.copy()
                    self._own_mirrors = True
                for v, c, m in zip(vs.tolist(), cols, mets):
                    a = int(self._rp[v])
                    self._col[a:a + len(c)] = c
                    self._met[a:a + len(m)] = m
                self._vf[vs] = nf
            else:
                self._rp, self._col, self._met, self._vf = splice_rows(self._rp, self._col, self._met, self._vf, vs, cols, mets, nf)
                self._own_mirrors = True
        self._pending = []

    @property
    def row_ptr(self) -> np.ndarray:
        if self._pending:
            self._flush()
        return self._rp

    @property
    def col(self) -> np.ndarray:
        if self._pending:
            self._flush()
        return self._col

    @property
    def metric(self) -> np.ndarray:
        if self._pending:
            self._flush()
        return self._met

    @property
    def vflags(self) -> np.ndarray:
        if self._pending:
            self._flush()
        return self._vf

    @property
    def n_edges_kept(self) -> int:
        return int(self.ctx.lib.hspf_graph_n_edges_kept(self.handle))

    # arrays of the device-resident graph (HSPF_GX_*)
    GX = {"row_ptr": (0, np.uint32), "col": (1, np.uint32), "metric": (2, np.uint32), "vflags": (3, np.uint8),
          "in_ptr": (4, np.uint32), "in_src": (5, np.uint32), "in_cost": (6, np.uint32), "in_pos": (7, np.uint32),
          "out_ptr": (8, np.uint32), "out_dst": (9, np.uint32), "out_cost": (10, np.uint32),
          "out_pos": (11, np.uint32), "rowflags": (12, np.uint8), "twoway": (13, np.uint8), "units": (14, np.uint32),
          "build_mode": (15, np.uint32), "ell_src": (16, np.uint32), "ell_cost": (17, np.uint32),
          "ell_out": (18, np.uint32), "summary": (19, np.uint32), "leaf": (20, np.uint8),
          "host_row_ptr": (21, np.uint32), "host_col": (22, np.uint32), "zcyc": (23, np.uint8)}

    def export(self, name: str) -> np.ndarray:
        """One array of the graph as it sits on the device (hspf_graph_export)."""
        which, dt = self.GX[name]
        nbytes = ctypes.c_size_t()
        rc = self.ctx.lib.hspf_graph_export(self.ctx.handle, self.handle, which, None, 0, ctypes.byref(nbytes))
        if rc != 0:
            raise HspfError(rc, "hspf_graph_export", self.ctx.last_error())
        out = np.empty(nbytes.value // np.dtype(dt).itemsize, dt)
        rc = self.ctx.lib.hspf_graph_export(self.ctx.handle, self.handle, which, out.ctypes.data_as(ctypes.c_void_p),
                                            out.nbytes, ctypes.byref(nbytes))
        if rc != 0:
            raise HspfError(rc, "hspf_graph_export", self.ctx.last_error())
        return out

    def patch(self, vertices, rows, vflags) -> None:
        """Replace whole rows (hspf_graph_patch): `rows[i]` = (col array, metric array) of `vertices[i]`,
        `vflags[i]` its new flags.  The numpy mirrors of the CSR are spliced the same way."""
        order = np.argsort(np.asarray(vertices, dtype=np.int64), kind="stable")
        vs = np.ascontiguousarray(np.asarray(vertices, dtype=np.uint32)[order])
        cols = [np.asarray(rows[i][0], dtype=np.uint32) for i in order]
        mets = [np.asarray(rows[i][1], dtype=np.uint32) for i in order]
        nf = np.ascontiguousarray(np.asarray(vflags, dtype=np.uint8)[order])
        rp = np.zeros(len(vs) + 1, np.uint32)
        rp[1:] = np.cumsum([len(c) for c in cols], dtype=np.uint64)
        dcol = np.ascontiguousarray(np.concatenate(cols)) if cols else np.zeros(0, np.uint32)
        dmet = np.ascontiguousarray(np.concatenate(mets)) if mets else np.zeros(0, np.uint32)
        if len(dcol) == 0:
            dcol = np.zeros(1, np.uint32); dmet = np.zeros(1, np.uint32)     # non-NULL pointers
        r = L.HspfRows(len(vs), _u32(vs), _u32(rp), _u32(dcol), _u32(dmet), nf.ctypes.data_as(L.u8p))
        t0 = time.perf_counter()
        rc = self.ctx.lib.hspf_graph_patch(self.ctx.handle, self.handle, ctypes.byref(r))
        self.last_patch_call_ms = (time.perf_counter() - t0) * 1e3     # the C call alone (the numpy mirrors below are this twin's own)
        if rc != 0:
            raise HspfError(rc, "hspf_graph_patch", self.ctx.last_error())
        # the numpy mirrors follow when they are next read (copies: the caller may reuse its arrays)
        self._pending.append((vs, [c.copy() for c in cols], [m_.copy() for m_ in mets], nf.copy()))

    def mask_words(self, roots) -> int:
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        w = ctypes.c_uint32()
        rc = self.ctx.lib.hspf_mask_words(self.ctx.handle, self.handle, _u32(roots), len(roots), ctypes.byref(w))
        if rc != 0:
            raise HspfError(rc, "hspf_mask_words", self.ctx.last_error())
        return int(w.value)

    def slot_table(self, root: int):
        """(H vertices, slot bases, total slots) of one root — see include/holo_spf_hip.h."""
        total = ctypes.c_uint32()
        cnt = self.ctx.lib.hspf_slot_table(self.ctx.handle, self.handle, root, None, None, 0, ctypes.byref(total))
        if cnt < 0:
            raise HspfError(cnt, "hspf_slot_table", self.ctx.last_error())
        hv = np.empty(cnt, np.uint32)
        hb = np.empty(cnt, np.uint32)
        self.ctx.lib.hspf_slot_table(self.ctx.handle, self.handle, root, _u32(hv), _u32(hb), cnt, ctypes.byref(total))
        return hv, hb, int(total.value)

    def slot_to_link(self, root: int, slot: int):
        """Map a first-hop slot of `root` back to (parent vertex, link position in its row,
        global link index)."""
        hv, hb, total = self.slot_table(root)
        if slot >= total:
            raise IndexError(slot)
        i = int(np.searchsorted(hb, slot, side="right")) - 1
        p = int(hv[i])
        j = slot - int(hb[i])
        return p, j, int(self.row_ptr[p]) + j

    def free(self):
        if self.handle:
            self.ctx.lib.hspf_graph_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
                self.free()
        except Exception:
            pass


class SpfContext:
    """One engine context: device, HIP stream, scratch (hspf_ctx).  One thread at a time."""

    def __init__(self, device: int = 0):
        self.lib = L.load()
        h = ctypes.c_void_p()
        rc = self.lib.hspf_init(device, ctypes.byref(h))
        if rc != 0:
            raise HspfError(rc, "hspf_init")
        self.handle = h
        self.device = device

    def last_error(self) -> str:
        return self.lib.hspf_last_error(self.handle).decode()

    def set_stream(self, hip_stream_ptr: int):
        rc = self.lib.hspf_set_stream(self.handle, ctypes.c_void_p(hip_stream_ptr))
        if rc != 0:
            raise HspfError(rc, "hspf_set_stream")

    def upload(self, row_ptr, col, metric, vflags, max_path_metric: int) -> SpfGraph:
        return SpfGraph(self, row_ptr, col, metric, vflags, max_path_metric)

    def stats(self) -> dict:
        s = L.HspfStats()
        self.lib.hspf_get_stats(self.handle, ctypes.byref(s))
        out = {name: getattr(s, name) for name, _ in L.HspfStats._fields_}
        out["dbg"] = list(out["dbg"])
        return out

    def run(self, graph: SpfGraph, roots: Sequence[int], run_flags: int = 0, *, want_mask: bool = True,
            mask_words: Optional[int] = None) -> SpfResult:
        """hspf_run(): results land in host numpy arrays."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        R, n = len(roots), graph.n
        W = mask_words or graph.mask_words(roots)
        dist = np.empty((R, n), np.uint32)
        hops = np.empty((R, n), np.uint16)
        flags = np.empty((R, n), np.uint16)
        mask = np.empty((R, n, W), np.uint64) if want_mask else None
        rank = np.empty((R, n), np.uint32) if (run_flags & RUN_POP_RANK) else None
        res = L.HspfResult(dist.ctypes.data, hops.ctypes.data, flags.ctypes.data,
                           mask.ctypes.data if want_mask else None, W,
                           rank.ctypes.data if rank is not None else None)
        rc = self.lib.hspf_run(self.handle, graph.handle, _u32(roots), R, run_flags, ctypes.byref(res))
        if rc != 0:
            raise HspfError(rc, "hspf_run", self.last_error())
        return SpfResult(dist, hops, flags, mask, rank, self.stats())

    def host_alloc(self, nbytes: int) -> PinnedBuffer:
        """hspf_host_alloc(): page-locked host memory for result buffers."""
        return PinnedBuffer(self, nbytes)

    def _packed_result(self, buf_u8: np.ndarray, R: int, n: int, ly, status, stats) -> PackedResult:
        dt = np.uint32 if ly.word_bytes == 4 else np.uint64
        words = buf_u8[: R * n * ly.word_bytes].view(dt).reshape(R, n)
        return PackedResult(words, int(ly.word_bytes), int(ly.dist_shift), int(ly.hops_shift), int(ly.hops_mask), int(ly.mask_bits),
                            int(ly.not_reached), status, stats)

    def run_packed(self, graph: SpfGraph, roots: Sequence[int], run_flags: int = 0, *, buffer=None) -> PackedResult:
        """hspf_run_packed(): packed words in host memory.  `buffer`: a PinnedBuffer (bus speed) or a writable numpy uint8
        array of at least 8 * R * N bytes (staged by the library); default: a fresh numpy array.  Raises HspfError with
        code E_NO_PACKED when the run's results do not fit packed words (the caller then uses run())."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        R, n = len(roots), graph.n
        if buffer is None:
            buffer = np.empty(8 * R * n, np.uint8)
        arr = buffer.array if isinstance(buffer, PinnedBuffer) else buffer
        status = np.zeros(R, np.uint8)
        ly = L.HspfPackedLayout()
        rc = self.lib.hspf_run_packed(self.handle, graph.handle, _u32(roots), R, run_flags, ctypes.c_void_p(arr.ctypes.data), arr.nbytes,
                                      ctypes.byref(ly), status.ctypes.data_as(L.u8p))
        if rc != 0:
            raise HspfError(rc, "hspf_run_packed", self.last_error())
        return self._packed_result(arr, R, n, ly, status, self.stats())

    def run_packed_device(self, graph: SpfGraph, roots: Sequence[int], run_flags: int, *, words_ptr: int, cap_bytes: int):
        """hspf_run_packed_device(): packed words at a device pointer; returns (layout struct, root status, stats)."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        status = np.zeros(len(roots), np.uint8)
        ly = L.HspfPackedLayout()
        rc = self.lib.hspf_run_packed_device(self.handle, graph.handle, _u32(roots), len(roots), run_flags, ctypes.c_void_p(words_ptr), cap_bytes,
                                             ctypes.byref(ly), status.ctypes.data_as(L.u8p))
        if rc != 0:
            raise HspfError(rc, "hspf_run_packed_device", self.last_error())
        return ly, status, self.stats()

    def run_packed_async(self, graph: SpfGraph, roots: Sequence[int], run_flags: int, buffer) -> tuple:
        """hspf_run_packed_async(): the run AND its copy to the host on a lane; returns a handle for wait_packed()."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        arr = buffer.array if isinstance(buffer, PinnedBuffer) else buffer
        status = np.zeros(len(roots), np.uint8)
        t = ctypes.c_uint64()
        rc = self.lib.hspf_run_packed_async(self.handle, graph.handle, _u32(roots), len(roots), run_flags, ctypes.c_void_p(arr.ctypes.data), arr.nbytes,
                                            status.ctypes.data_as(L.u8p), ctypes.byref(t))
        if rc != 0:
            raise HspfError(rc, "hspf_run_packed_async", self.last_error())
        return (int(t.value), arr, len(roots), graph.n, status)

    def wait_packed(self, handle: tuple) -> PackedResult:
        ticket, arr, R, n, status = handle
        ly, s = L.HspfPackedLayout(), L.HspfStats()
        rc = self.lib.hspf_wait_packed(self.handle, ticket, ctypes.byref(ly), ctypes.byref(s))
        if rc != 0:
            raise HspfError(rc, "hspf_wait_packed", self.last_error())
        st = {name: getattr(s, name) for name, _ in L.HspfStats._fields_}
        st["dbg"] = list(st["dbg"])
        return self._packed_result(arr, R, n, ly, status, st)

    def run_device(self, graph: SpfGraph, roots: Sequence[int], run_flags: int, *, dist_ptr: int,
                   hops_ptr: int = 0, flags_ptr: int = 0, mask_ptr: int = 0, mask_words: int = 1,
                   pop_rank_ptr: int = 0) -> dict:
        """hspf_run_device(): results stay in HBM at the given device pointers (e.g. torch
        tensors' data_ptr()); returns the stats of the run."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        res = L.HspfResult(dist_ptr or None, hops_ptr or None, flags_ptr or None, mask_ptr or None,
                           mask_words, pop_rank_ptr or None)
        rc = self.lib.hspf_run_device(self.handle, graph.handle, _u32(roots), len(roots), run_flags,
                                      ctypes.byref(res))
        if rc != 0:
            raise HspfError(rc, "hspf_run_device", self.last_error())
        return self.stats()

    def run_device_async(self, graph: SpfGraph, roots: Sequence[int], run_flags: int, *, dist_ptr: int,
                         hops_ptr: int = 0, flags_ptr: int = 0, mask_ptr: int = 0, mask_words: int = 1) -> int:
        """hspf_run_device_async(): hands the run to a lane of this context and returns its ticket at once; the device
        buffers must stay valid (and unshared with other runs in flight) until wait(ticket)."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        res = L.HspfResult(dist_ptr or None, hops_ptr or None, flags_ptr or None, mask_ptr or None, mask_words, None)
        t = ctypes.c_uint64()
        rc = self.lib.hspf_run_device_async(self.handle, graph.handle, _u32(roots), len(roots), run_flags,
                                            ctypes.byref(res), ctypes.byref(t))
        if rc != 0:
            raise HspfError(rc, "hspf_run_device_async", self.last_error())
        return int(t.value)

    def wait(self, ticket: int) -> dict:
        """hspf_wait(): blocks until the run behind `ticket` is over; returns its stats."""
        s = L.HspfStats()
        rc = self.lib.hspf_wait(self.handle, ticket, ctypes.byref(s))
        if rc != 0:
            raise HspfError(rc, "hspf_wait", self.last_error())
        out = {name: getattr(s, name) for name, _ in L.HspfStats._fields_}
        out["dbg"] = list(out["dbg"])
        return out

    def wait_all(self) -> None:
        self.lib.hspf_wait_all(self.handle)

    def async_lanes(self) -> int:
        return int(self.lib.hspf_async_lanes(self.handle))

    @staticmethod
    def async_lanes_of(ctx_handle) -> int:
        """hspf_async_lanes of a raw hspf_ctx handle (e.g. MultiEngine.ctx_handle(i))."""
        return int(L.load().hspf_async_lanes(ctx_handle))

    def ancestors_device(self, graph: SpfGraph, roots: Sequence[int], run_flags: int, *, dist_ptr: int, hops_ptr: int,
                         flags_ptr: int, level: int, n_words: int, level_rank_ptr: int, level_count_ptr: int, anc_ptr: int) -> int:
        """hspf_ancestors_device(): level-L ancestor bit sets of every (root, vertex) of a previous run_device(); all
        `*_ptr` are device pointers.  Returns 0, or E_TOO_MANY_SLOTS (-5) when n_words is too small (level counts are
        filled in either way)."""
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        rc = self.lib.hspf_ancestors_device(self.handle, graph.handle, _u32(roots), len(roots), run_flags, dist_ptr, hops_ptr,
                                            flags_ptr, level, n_words, level_rank_ptr or None, level_count_ptr, anc_ptr)
        if rc not in (0, E_TOO_MANY_SLOTS):
            raise HspfError(rc, "hspf_ancestors_device", self.last_error())
        return rc

    def routes_device(self, n_vertices: int, n_roots: int, mask_words: int, dist_ptr: int, flags_ptr: int,
                      mask_ptr: int, pfx_ptr, pfx_vertex, pfx_metric, *, best_metric_ptr: int,
                      best_entry_ptr: int, nexthop_mask_ptr: int, flags: int = 0, pfx_origin=None,
                      init_exists=None, init_metric=None, init_origin=None) -> None:
        """hspf_routes_device(): prefix attachment for every root of a previous run_device(); all
        `*_ptr` arguments are device pointers, the prefix table is host numpy.  PFX_ORDERED tables also pass
        pfx_origin (and, optionally, the per-prefix route an earlier area left: init_exists / init_metric / init_origin)."""
        src = (pfx_ptr, pfx_vertex, pfx_metric)
        pfx_ptr = np.ascontiguousarray(pfx_ptr, np.uint32)
        pfx_vertex = np.ascontiguousarray(pfx_vertex, np.uint32)
        pfx_metric = np.ascontiguousarray(pfx_metric, np.uint32)
        if flags & PFX_RESIDENT and any(a is not b for a, b in zip(src, (pfx_ptr, pfx_vertex, pfx_metric))):
            # HSPF_PFX_RESIDENT tells the library "the table at THESE host addresses is the one you hold": a converted
            # copy is a temporary whose address the next temporary may reuse, and a stale device table would be used silently
            raise ValueError("routes_device: PFX_RESIDENT needs the caller's own contiguous uint32 arrays (a conversion made a copy)")
        keep = []

        def opt(a, dt, ptr_t):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data_as(ptr_t)
        t = L.HspfPrefixTable(len(pfx_ptr) - 1, len(pfx_vertex), _u32(pfx_ptr), _u32(pfx_vertex), _u32(pfx_metric), flags,
                              opt(pfx_origin, np.uint32, L.u32p), opt(init_exists, np.uint8, L.u8p),
                              opt(init_metric, np.uint32, L.u32p), opt(init_origin, np.uint32, L.u32p))
        o = L.HspfRoutes(best_metric_ptr, best_entry_ptr, nexthop_mask_ptr)
        rc = self.lib.hspf_routes_device(self.handle, n_vertices, n_roots, mask_words, dist_ptr, flags_ptr, mask_ptr,
                                         ctypes.byref(t), ctypes.byref(o))
        if rc != 0:
            raise HspfError(rc, "hspf_routes_device", self.last_error())

    def rib_clear_device(self, n_prefixes: int, mask_words: int, *, best_metric_ptr: int, best_entry_ptr: int, nexthop_mask_ptr: int,
                         origin_ptr: int) -> None:
        """hspf_rib_clear_device(): the empty instance-wide RIB state in front of the first area (device pointers)."""
        rib = L.HspfRibDevice(n_prefixes, mask_words, best_metric_ptr, best_entry_ptr, nexthop_mask_ptr, origin_ptr)
        rc = self.lib.hspf_rib_clear_device(self.handle, ctypes.byref(rib))
        if rc != 0:
            raise HspfError(rc, "hspf_rib_clear_device", self.last_error())

    def rib_fold_device(self, n_vertices: int, area_mask_words: int, dist_ptr: int, flags_ptr: int, mask_ptr: int, pfx_ptr, pfx_vertex,
                        pfx_metric, pfx_origin, prefix_map, area_index: int, word_offset: int, *, n_prefixes: int, mask_words: int,
                        best_metric_ptr: int, best_entry_ptr: int, nexthop_mask_ptr: int, origin_ptr: int) -> None:
        """hspf_rib_fold_device(): the ordered fold of ONE area's table into the instance-wide RIB state on the device
        (several areas, one RIB: holo-ospf/src/route.rs:343-448 on what the earlier areas left)."""
        pfx_ptr = np.ascontiguousarray(pfx_ptr, np.uint32); pfx_vertex = np.ascontiguousarray(pfx_vertex, np.uint32)
        pfx_metric = np.ascontiguousarray(pfx_metric, np.uint32); pfx_origin = np.ascontiguousarray(pfx_origin, np.uint32)
        prefix_map = np.ascontiguousarray(prefix_map, np.uint32)
        t = L.HspfPrefixTable(len(pfx_ptr) - 1, len(pfx_vertex), _u32(pfx_ptr), _u32(pfx_vertex), _u32(pfx_metric), PFX_SATURATING | PFX_ORDERED,
                              _u32(pfx_origin), None, None, None)
        rib = L.HspfRibDevice(n_prefixes, mask_words, best_metric_ptr, best_entry_ptr, nexthop_mask_ptr, origin_ptr)
        rc = self.lib.hspf_rib_fold_device(self.handle, n_vertices, area_mask_words, dist_ptr, flags_ptr, mask_ptr, ctypes.byref(t), _u32(prefix_map),
                                           area_index, word_offset, ctypes.byref(rib))
        if rc != 0:
            raise HspfError(rc, "hspf_rib_fold_device", self.last_error())

    def routes_diff_device(self, n_roots: int, n_prefixes: int, mask_words: int, old: tuple, new: tuple, *,
                           action_ptr: int, changed_ptr: int, changed_ptr_ptr: int) -> None:
        """hspf_routes_diff_device(): old / new = (best_metric_ptr, best_entry_ptr, nexthop_mask_ptr) of two
        hspf_routes_device() result sets over the same prefix list; all device pointers."""
        o, n = L.HspfRoutes(*old), L.HspfRoutes(*new)
        rc = self.lib.hspf_routes_diff_device(self.handle, n_roots, n_prefixes, mask_words, ctypes.byref(o), ctypes.byref(n),
                                              action_ptr, changed_ptr, changed_ptr_ptr)
        if rc != 0:
            raise HspfError(rc, "hspf_routes_diff_device", self.last_error())

    def routes_pack(self, n_roots: int, n_prefixes: int, mask_words: int, new: tuple, *, action_ptr: int, changed_ptr: int,
                    changed_ptr_ptr: int) -> np.ndarray:
        """hspf_routes_pack(): the changed (root, prefix) pairs of the last routes_diff_device() as ONE record stream,
        one device-to-host copy.  Returns [n_records, 6 + 2 W] u32: root, prefix, action, metric, entry, 0, mask words."""
        k = int(self.lib.hspf_routes_diff_count(self.handle))
        rec = np.zeros((k, 6 + 2 * mask_words), np.uint32)
        if k:
            n = L.HspfRoutes(*new)
            rc = self.lib.hspf_routes_pack(self.handle, n_roots, n_prefixes, mask_words, ctypes.byref(n), action_ptr, changed_ptr,
                                           changed_ptr_ptr, k, _u32(rec))
            if rc != 0:
                raise HspfError(rc, "hspf_routes_pack", self.last_error())
        return rec

    def close(self):
        if self.handle:
            self.lib.hspf_shutdown(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- several GPUs: hspf_multi_* (include/holo_spf_hip.h "several GPUs") -------------------------------------------------

GATHER_DIST, GATHER_HOPS, GATHER_FLAGS, GATHER_MASK = 1, 2, 4, 8
GATHER_ASYNC = 0x100


def shard_bounds(n_roots: int, world: int, rank: int):
    """hspf_shard_bounds: [begin, end) of `rank` — whole 64-root batches (pure host arithmetic, no GPU needed)."""
    lib = L.load()
    b, e = ctypes.c_uint32(), ctypes.c_uint32()
    lib.hspf_shard_bounds(n_roots, world, rank, ctypes.byref(b), ctypes.byref(e))
    return int(b.value), int(e.value)


def plan_areas(roots_per_area: Sequence[int], world: int):
    """hspf_plan_areas: areas first, then roots — list of (rank, area, root_begin, root_end)."""
    lib = L.load()
    rpa = np.ascontiguousarray(roots_per_area, np.uint32)
    n = lib.hspf_plan_areas(len(rpa), _u32(rpa), world, None, 0)
    buf = (L.HspfAreaSlice * max(n, 1))()
    lib.hspf_plan_areas(len(rpa), _u32(rpa), world, buf, n)
    return [(int(s.rank), int(s.area), int(s.root_begin), int(s.root_end)) for s in buf[:n]]


def multi_unique_id() -> bytes:
    lib = L.load()
    buf = (ctypes.c_uint8 * L.COMM_ID_BYTES)()
    rc = lib.hspf_multi_unique_id(buf)
    if rc != 0:
        raise HspfError(rc, "hspf_multi_unique_id")
    return bytes(buf)


class MultiEngine:
    """hspf_multi: one engine context per rank.  `devices` = the ordinals this process drives (an ordinal may repeat:
    several contexts on one GPU); single process when unique_id is None (world = len(devices)), else one member of a
    job of `world` ranks whose RCCL communicator is created from `unique_id`."""

    def __init__(self, devices: Sequence[int], world: Optional[int] = None, first_rank: int = 0, unique_id: Optional[bytes] = None):
        self.lib = L.load()
        self.devices = list(devices)
        self.world = world if world is not None else len(self.devices)
        self.first_rank = first_rank
        ords = (ctypes.c_int * len(self.devices))(*self.devices)
        idbuf = (ctypes.c_uint8 * L.COMM_ID_BYTES)(*unique_id) if unique_id is not None else None
        cfg = L.HspfMultiConfig(len(self.devices), ords, self.world, first_rank, idbuf)
        h = ctypes.c_void_p()
        rc = self.lib.hspf_multi_init(ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            raise HspfError(rc, "hspf_multi_init", (self.lib.hspf_multi_init_error() or b"").decode())
        self.handle = h
        self.graph = None

    def last_error(self) -> str:
        return (self.lib.hspf_multi_last_error(self.handle) or b"").decode()

    def ctx_handle(self, i: int):
        return self.lib.hspf_multi_ctx(self.handle, i)

    def upload(self, row_ptr, col, metric, vflags, max_path_metric: int):
        row_ptr = np.ascontiguousarray(row_ptr, np.uint32); col = np.ascontiguousarray(col, np.uint32)
        metric = np.ascontiguousarray(metric, np.uint32); vflags = np.ascontiguousarray(vflags, np.uint8)
        csr = L.HspfCsr(len(row_ptr) - 1, len(col), _u32(row_ptr), _u32(col), _u32(metric),
                        vflags.ctypes.data_as(L.u8p), max_path_metric)
        g = ctypes.c_void_p()
        rc = self.lib.hspf_multi_graph_upload(self.handle, ctypes.byref(csr), ctypes.byref(g))
        if rc != 0:
            raise HspfError(rc, "hspf_multi_graph_upload", self.last_error())
        return g

    def free_graph(self, g):
        self.lib.hspf_multi_graph_free(self.handle, g)

    def mask_words(self, g, roots) -> int:
        roots = np.ascontiguousarray(roots, np.uint32)
        w = ctypes.c_uint32()
        rc = self.lib.hspf_multi_mask_words(self.handle, g, _u32(roots), len(roots), ctypes.byref(w))
        if rc != 0:
            raise HspfError(rc, "hspf_multi_mask_words", self.last_error())
        return int(w.value)

    def run(self, g, roots, run_flags: int, results: Sequence[dict], gather: int) -> None:
        """hspf_multi_run: results[i] = dict(dist=ptr, hops=ptr, flags=ptr, mask=ptr, mask_words=W) of device pointers on
        local device i, each table sized for ALL roots."""
        roots = np.ascontiguousarray(roots, np.uint32)
        arr = (L.HspfResult * len(results))()
        for i, r in enumerate(results):
            arr[i] = L.HspfResult(r["dist"], r.get("hops") or None, r.get("flags") or None, r.get("mask") or None,
                                  r.get("mask_words", 1), None)
        rc = self.lib.hspf_multi_run(self.handle, g, _u32(roots), len(roots), run_flags, arr, gather)
        if rc != 0:
            raise HspfError(rc, "hspf_multi_run", self.last_error())

    def _results(self, results: Sequence[dict]):
        arr = (L.HspfResult * len(results))()
        for i, r in enumerate(results):
            arr[i] = L.HspfResult(r["dist"], r.get("hops") or None, r.get("flags") or None, r.get("mask") or None,
                                  r.get("mask_words", 1), None)
        return arr

    def run_async(self, g, roots, run_flags: int, results: Sequence[dict]) -> int:
        """hspf_multi_run_async: every local device's slice goes to a lane of its context; returns the ticket."""
        roots = np.ascontiguousarray(roots, np.uint32)
        t = ctypes.c_uint64()
        rc = self.lib.hspf_multi_run_async(self.handle, g, _u32(roots), len(roots), run_flags, self._results(results), ctypes.byref(t))
        if rc != 0:
            raise HspfError(rc, "hspf_multi_run_async", self.last_error())
        return int(t.value)

    def run_wait(self, ticket: int, results: Sequence[dict], gather: int) -> None:
        """hspf_multi_run_wait: the slices of `ticket` are complete; then the exchange selected in `gather`."""
        rc = self.lib.hspf_multi_run_wait(self.handle, ticket, self._results(results), gather)
        if rc != 0:
            raise HspfError(rc, "hspf_multi_run_wait", self.last_error())

    def wait(self) -> None:
        rc = self.lib.hspf_multi_wait(self.handle)
        if rc != 0:
            raise HspfError(rc, "hspf_multi_wait", self.last_error())

    def allgather_rows(self, table_ptrs: Sequence[int], row_bytes: int, n_roots: int) -> None:
        arr = (ctypes.c_void_p * len(table_ptrs))(*table_ptrs)
        rc = self.lib.hspf_multi_allgather_rows(self.handle, arr, row_bytes, n_roots)
        if rc != 0:
            raise HspfError(rc, "hspf_multi_allgather_rows", self.last_error())

    def stats(self, i: int = 0) -> dict:
        s = L.HspfStats()
        self.lib.hspf_multi_get_stats(self.handle, i, ctypes.byref(s))
        out = {name: getattr(s, name) for name, _ in L.HspfStats._fields_}
        out["dbg"] = list(out["dbg"])
        return out

    def close(self):
        if self.handle:
            self.lib.hspf_multi_shutdown(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
