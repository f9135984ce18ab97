"""ctypes front-end of oracle/liboracle_spf.so (CPU restatement of the SPF hot loop).

TEST INFRASTRUCTURE ONLY — see oracle/spf_oracle.cpp for the reference file:line map.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REF, MAP, HEAP = 0, 1, 2          # variants, see spf_oracle.cpp
RUN_NET_NEXTHOPS = 0x01
RUN_IGNORE_OVERLOAD = 0x02
INF = 0xFFFFFFFF


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle_spf.so")
    src = os.path.join(_HERE, "spf_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_spf.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        u32p = ctypes.POINTER(ctypes.c_uint32)
        _LIB.oracle_spf_run.restype = ctypes.c_int
        _LIB.oracle_spf_run_mt.restype = ctypes.c_int
        _LIB.oracle_mask_words.restype = ctypes.c_uint32
    return _LIB


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ty))


@dataclass
class OracleResult:
    dist: np.ndarray        # [R, N] u32
    hops: np.ndarray        # [R, N] u16
    flags: np.ndarray       # [R, N] u16 (bit0 = in SPT)
    pop_rank: np.ndarray    # [R, N] u32
    mask: np.ndarray        # [R, N, W] u64
    n_nexthops: np.ndarray  # [R, N] u32  (IS-IS Vec length incl. duplicates)
    n_parents: np.ndarray   # [R, N] u32
    work: np.ndarray        # [R] u64 (variant REF only: scan steps)


def mask_words(row_ptr, col, metric, vflags, roots) -> int:
    lib = _lib()
    n = len(row_ptr) - 1
    e = len(col)
    w = 1
    for r in roots:
        if r == INF:
            continue
        w = max(w, lib.oracle_mask_words(
            ctypes.c_uint32(n), ctypes.c_uint32(e), _p(row_ptr, ctypes.c_uint32),
            _p(col, ctypes.c_uint32), _p(metric, ctypes.c_uint32), _p(vflags, ctypes.c_uint8),
            ctypes.c_uint32(int(r))))
    return int(w)


def run(row_ptr, col, metric, vflags, max_path_metric, roots, run_flags=0, variant=MAP,
        mask_words_=None, threads: int = 1) -> OracleResult:
    """Run the oracle for every root: sequentially (threads = 1) or with the roots dealt to `threads` workers."""
    lib = _lib()
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    metric = np.ascontiguousarray(metric, dtype=np.uint32)
    vflags = np.ascontiguousarray(vflags, dtype=np.uint8)
    roots = np.ascontiguousarray(roots, dtype=np.uint32)
    n = len(row_ptr) - 1
    e = len(col)
    R = len(roots)
    W = mask_words_ or mask_words(row_ptr, col, metric, vflags, roots)
    dist = np.empty((R, n), np.uint32)
    hops = np.empty((R, n), np.uint16)
    flags = np.empty((R, n), np.uint16)
    rank = np.empty((R, n), np.uint32)
    mask = np.empty((R, n, W), np.uint64)
    nnh = np.empty((R, n), np.uint32)
    npar = np.empty((R, n), np.uint32)
    work = np.zeros((R,), np.uint64)
    rc = lib.oracle_spf_run_mt(
        ctypes.c_uint32(n), ctypes.c_uint32(e), _p(row_ptr, ctypes.c_uint32), _p(col, ctypes.c_uint32),
        _p(metric, ctypes.c_uint32), _p(vflags, ctypes.c_uint8), ctypes.c_uint32(max_path_metric),
        _p(roots, ctypes.c_uint32), ctypes.c_uint32(R), ctypes.c_uint32(run_flags), ctypes.c_int(variant),
        _p(dist, ctypes.c_uint32), _p(hops, ctypes.c_uint16), _p(flags, ctypes.c_uint16),
        _p(rank, ctypes.c_uint32), _p(mask, ctypes.c_uint64), ctypes.c_uint32(W),
        _p(nnh, ctypes.c_uint32), _p(npar, ctypes.c_uint32), _p(work, ctypes.c_uint64), ctypes.c_uint32(max(1, int(threads))))
    if rc != 0:
        raise RuntimeError(f"oracle_spf_run failed: {rc}")
    return OracleResult(dist, hops, flags, rank, mask, nnh, npar, work)


class Runner:
    """Repeated runs into preallocated dist / hops / mask arrays (the diagnostic outputs are skipped): what bench.py's
    all-cores CPU baseline times, so that the baseline is the SPF loop and not the page faults of fresh result arrays."""

    def __init__(self, row_ptr, col, metric, vflags, max_path_metric, max_roots: int, mask_words_: int = 1):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.metric = np.ascontiguousarray(metric, dtype=np.uint32)
        self.vflags = np.ascontiguousarray(vflags, dtype=np.uint8)
        self.maxp = int(max_path_metric)
        self.n = len(self.row_ptr) - 1
        self.W = mask_words_
        self.dist = np.zeros((max_roots, self.n), np.uint32)
        self.hops = np.zeros((max_roots, self.n), np.uint16)
        self.mask = np.zeros((max_roots, self.n, self.W), np.uint64)

    def run(self, roots, run_flags=0, variant=HEAP, threads: int = 1):
        lib = _lib()
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        R = len(roots)
        assert R <= self.dist.shape[0]
        rc = lib.oracle_spf_run_mt(
            ctypes.c_uint32(self.n), ctypes.c_uint32(len(self.col)), _p(self.row_ptr, ctypes.c_uint32), _p(self.col, ctypes.c_uint32),
            _p(self.metric, ctypes.c_uint32), _p(self.vflags, ctypes.c_uint8), ctypes.c_uint32(self.maxp),
            _p(roots, ctypes.c_uint32), ctypes.c_uint32(R), ctypes.c_uint32(run_flags), ctypes.c_int(variant),
            _p(self.dist, ctypes.c_uint32), _p(self.hops, ctypes.c_uint16), None, None, _p(self.mask, ctypes.c_uint64),
            ctypes.c_uint32(self.W), None, None, None, ctypes.c_uint32(max(1, int(threads))))
        if rc != 0:
            raise RuntimeError(f"oracle_spf_run_mt failed: {rc}")
        return self.dist[:R], self.hops[:R], self.mask[:R]
