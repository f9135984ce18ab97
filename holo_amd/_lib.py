"""ctypes binding of libholo_spf_hip.so (include/holo_spf_hip.h).

Fails loudly: there is no Python / CPU fallback for the SPF engine.  If the shared library is
missing or a symbol is absent, importing callers get an exception, not a slow path.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HSPF_LIB") or os.path.join(HERE, "libholo_spf_hip.so")     # HSPF_LIB: another build of the library (A/B measurements)

u8p = ctypes.POINTER(ctypes.c_uint8)
u16p = ctypes.POINTER(ctypes.c_uint16)
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)


class HspfCsr(ctypes.Structure):
    _fields_ = [("n_vertices", ctypes.c_uint32), ("n_edges", ctypes.c_uint32),
                ("row_ptr", u32p), ("col", u32p), ("metric", u32p), ("vflags", u8p),
                ("max_path_metric", ctypes.c_uint32)]


class HspfResult(ctypes.Structure):
    _fields_ = [("dist", ctypes.c_void_p), ("hops", ctypes.c_void_p), ("vflags_out", ctypes.c_void_p),
                ("first_hop_mask", ctypes.c_void_p), ("n_mask_words", ctypes.c_uint32),
                ("pop_rank", ctypes.c_void_p)]


class HspfStats(ctypes.Structure):
    _fields_ = [("n_roots", ctypes.c_uint32), ("n_batches", ctypes.c_uint32),
                ("n_relax_launches", ctypes.c_uint32), ("n_dag_launches", ctypes.c_uint32),
                ("n_exact_roots", ctypes.c_uint32), ("n_mask_words", ctypes.c_uint32),
                ("ms_total", ctypes.c_float), ("ms_relax", ctypes.c_float), ("ms_dag", ctypes.c_float),
                ("ms_finish", ctypes.c_float), ("ms_d2h", ctypes.c_float),
                ("state_bytes", ctypes.c_uint32), ("narrow_overflow", ctypes.c_uint32),
                ("rows_recomputed", ctypes.c_uint64), ("single_wg", ctypes.c_uint32), ("lane_vertex", ctypes.c_uint32),
                ("dbg", ctypes.c_uint32 * 4),
                ("n_repaired_roots", ctypes.c_uint32), ("repair_sweeps", ctypes.c_uint32), ("repair_evals", ctypes.c_uint32),
                ("repair_groups", ctypes.c_uint32), ("ms_repair", ctypes.c_float)]


class HspfPackedLayout(ctypes.Structure):
    _fields_ = [("word_bytes", ctypes.c_uint32), ("dist_shift", ctypes.c_uint32), ("hops_shift", ctypes.c_uint32),
                ("hops_mask", ctypes.c_uint32), ("mask_bits", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("not_reached", ctypes.c_uint64)]


class HspfPrefixTable(ctypes.Structure):
    _fields_ = [("n_prefixes", ctypes.c_uint32), ("n_entries", ctypes.c_uint32),
                ("pfx_ptr", u32p), ("pfx_vertex", u32p), ("pfx_metric", u32p), ("flags", ctypes.c_uint32),
                ("pfx_origin", u32p), ("init_exists", u8p), ("init_metric", u32p), ("init_origin", u32p)]


class HspfRows(ctypes.Structure):
    _fields_ = [("n_changed", ctypes.c_uint32), ("vertex", u32p), ("row_ptr", u32p), ("col", u32p),
                ("metric", u32p), ("vflags", u8p)]


class HspfKeyedLsdb(ctypes.Structure):
    _fields_ = [("n_vertices", ctypes.c_uint32), ("n_links", ctypes.c_uint32), ("vertex_key", u64p), ("row_ptr", u32p),
                ("target_key", u64p), ("metric", u32p), ("vflags", u8p), ("max_path_metric", ctypes.c_uint32)]


class HspfRoutes(ctypes.Structure):
    _fields_ = [("best_metric", ctypes.c_void_p), ("best_entry", ctypes.c_void_p), ("nexthop_mask", ctypes.c_void_p)]


class HspfRibDevice(ctypes.Structure):
    _fields_ = [("n_prefixes", ctypes.c_uint32), ("n_mask_words", ctypes.c_uint32), ("best_metric", ctypes.c_void_p),
                ("best_entry", ctypes.c_void_p), ("nexthop_mask", ctypes.c_void_p), ("origin", ctypes.c_void_p)]


class HspfMultiConfig(ctypes.Structure):
    _fields_ = [("n_local", ctypes.c_uint32), ("device_ordinals", ctypes.POINTER(ctypes.c_int)),
                ("world", ctypes.c_uint32), ("first_rank", ctypes.c_uint32), ("unique_id", u8p)]


class HspfAreaSlice(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_uint32), ("area", ctypes.c_uint32), ("root_begin", ctypes.c_uint32), ("root_end", ctypes.c_uint32)]


COMM_ID_BYTES = 128

# every symbol include/holo_spf_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("hspf_abi_version", ctypes.c_uint32, []),
    ("hspf_device_count", ctypes.c_int, []),
    ("hspf_init", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    ("hspf_shutdown", None, [ctypes.c_void_p]),
    ("hspf_strerror", ctypes.c_char_p, [ctypes.c_int]),
    ("hspf_last_error", ctypes.c_char_p, [ctypes.c_void_p]),
    ("hspf_set_stream", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("hspf_get_stream", ctypes.c_void_p, [ctypes.c_void_p]),
    ("hspf_graph_upload", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(HspfCsr), ctypes.POINTER(ctypes.c_void_p)]),
    ("hspf_graph_upload_keyed", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(HspfKeyedLsdb), ctypes.POINTER(ctypes.c_void_p), u32p]),
    ("hspf_graph_free", None, [ctypes.c_void_p, ctypes.c_void_p]),
    ("hspf_graph_n_vertices", ctypes.c_uint32, [ctypes.c_void_p]),
    ("hspf_graph_n_edges", ctypes.c_uint32, [ctypes.c_void_p]),
    ("hspf_graph_n_edges_kept", ctypes.c_uint32, [ctypes.c_void_p]),
    ("hspf_graph_patch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(HspfRows)]),
    ("hspf_graph_export", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    ("hspf_mask_words", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, u32p]),
    ("hspf_slot_table", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, u32p, u32p, ctypes.c_uint32, u32p]),
    ("hspf_run", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(HspfResult)]),
    ("hspf_run_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(HspfResult)]),
    ("hspf_get_stats", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(HspfStats)]),
    ("hspf_run_device_async", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.POINTER(HspfResult), u64p]),
    ("hspf_wait", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(HspfStats)]),
    ("hspf_wait_all", ctypes.c_int, [ctypes.c_void_p]),
    ("hspf_async_lanes", ctypes.c_uint32, [ctypes.c_void_p]),
    ("hspf_recommend_cpu", ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    # packed results (ABI 7)
    ("hspf_host_alloc", ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    ("hspf_host_free", None, [ctypes.c_void_p, ctypes.c_void_p]),
    ("hspf_device_alloc", ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    ("hspf_device_free", None, [ctypes.c_void_p, ctypes.c_void_p]),
    ("hspf_device_to_host", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    ("hspf_host_to_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    ("hspf_run_packed", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(HspfPackedLayout), u8p]),
    ("hspf_run_packed_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32,
                                              ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(HspfPackedLayout), u8p]),
    ("hspf_run_packed_async", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_void_p, ctypes.c_size_t, u8p, u64p]),
    ("hspf_wait_packed", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(HspfPackedLayout), ctypes.POINTER(HspfStats)]),
    ("hspf_routes_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.POINTER(HspfPrefixTable), ctypes.POINTER(HspfRoutes)]),
    ("hspf_rib_clear_device", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(HspfRibDevice)]),
    ("hspf_rib_fold_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.POINTER(HspfPrefixTable), u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(HspfRibDevice)]),
    ("hspf_routes_diff_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                               ctypes.POINTER(HspfRoutes), ctypes.POINTER(HspfRoutes),
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("hspf_routes_diff_count", ctypes.c_uint32, [ctypes.c_void_p]),
    ("hspf_routes_pack", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(HspfRoutes),
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, u32p]),
    ("hspf_ancestors_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    # several GPUs
    ("hspf_multi_unique_id", ctypes.c_int, [u8p]),
    ("hspf_multi_init", ctypes.c_int, [ctypes.POINTER(HspfMultiConfig), ctypes.POINTER(ctypes.c_void_p)]),
    ("hspf_multi_init_error", ctypes.c_char_p, []),
    ("hspf_multi_shutdown", None, [ctypes.c_void_p]),
    ("hspf_multi_last_error", ctypes.c_char_p, [ctypes.c_void_p]),
    ("hspf_multi_ctx", ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_uint32]),
    ("hspf_multi_n_local", ctypes.c_uint32, [ctypes.c_void_p]),
    ("hspf_multi_graph_upload", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(HspfCsr), ctypes.POINTER(ctypes.c_void_p)]),
    ("hspf_multi_graph_patch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(HspfRows)]),
    ("hspf_multi_graph_free", None, [ctypes.c_void_p, ctypes.c_void_p]),
    ("hspf_multi_graph_local", ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_uint32]),
    ("hspf_shard_bounds", None, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p]),
    ("hspf_plan_areas", ctypes.c_uint32, [ctypes.c_uint32, u32p, ctypes.c_uint32, ctypes.POINTER(HspfAreaSlice), ctypes.c_uint32]),
    ("hspf_multi_mask_words", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, u32p]),
    ("hspf_multi_run", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.POINTER(HspfResult), ctypes.c_uint32]),
    ("hspf_multi_wait", ctypes.c_int, [ctypes.c_void_p]),
    ("hspf_multi_run_async", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, u32p, ctypes.c_uint32, ctypes.c_uint32,
                                            ctypes.POINTER(HspfResult), u64p]),
    ("hspf_multi_run_wait", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(HspfResult), ctypes.c_uint32]),
    ("hspf_multi_allgather_rows", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint32]),
    ("hspf_multi_get_stats", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(HspfStats)]),
]

_lib = None


def load():
    """Load the shared library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m holo_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the SPF engine.")
    # PyTorch-ROCm wheels ship their own libamdhip64; if it is loaded AFTER this library has pulled
    # in /opt/rocm's copy, torch finds "no HIP GPUs".  Whoever wants torch device buffers next to
    # the engine (bench.py, holo_amd.routes) gets one HIP runtime per process this way.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001  (torch is optional for the engine itself)
        pass
    # Runs in flight on the lanes of a context are HIP streams of their own; the runtime maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and streams sharing one serialise.  It is a property of the PROCESS:
    # the host sets it (here: this Python host, before HIP initialises), the library does not.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
