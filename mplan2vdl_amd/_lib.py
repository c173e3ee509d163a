"""ctypes binding of libvdl.so (C ABI: include/vdl.h).  Fails loudly when the HIP library
has not been built: there is no CPU fallback in the product path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VDL_LIB") or os.path.join(_HERE, "lib", "libvdl.so")      # VDL_LIB: kernel experiments (tools/build_variant.sh)

VDL_OK, VDL_ERR_PARSE, VDL_ERR_COLUMN, VDL_ERR_UNSUPPORTED, VDL_ERR_DEVICE, VDL_ERR_ARG, VDL_ERR_SHAPE, VDL_ERR_NOMEM = range(8)
REDUCE_NONE, REDUCE_SUM, REDUCE_MIN, REDUCE_MAX, REDUCE_FIRST = range(5)

_lib = None
COMM_ID_BYTES = 128
ALL_GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
ALL_TO_ALL_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p,
                                 ctypes.POINTER(ctypes.c_size_t))


class CommHost(ctypes.Structure):          # vdl_comm_host (include/vdl.h)
    _fields_ = [("user", ctypes.c_void_p), ("all_gather", ALL_GATHER_FN), ("all_to_all", ALL_TO_ALL_FN)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mplan2vdl_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mplan2vdl_amd/csrc`); the engine has no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; importing it first makes the dynamic linker hand that
    # same runtime to libvdl.so (one HIP runtime per process, so torch device pointers are usable).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, cp, i64, i32 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int
    P = ctypes.POINTER
    sig = {
        "vdl_open": (i32, [P(vp), i32]),
        "vdl_close": (None, [vp]),
        "vdl_last_error": (cp, [vp]),
        "vdl_version": (cp, []),
        "vdl_set_stream": (i32, [vp, vp]),
        "vdl_use_own_stream": (i32, [vp]),
        "vdl_register_column": (i32, [vp, cp, vp, i32, i64]),
        "vdl_upload_column": (i32, [vp, cp, vp, i32, i64]),
        "vdl_generate_column": (i32, [vp, cp, i32, i64, i64, ctypes.c_uint64, i64, i64, i64, i64]),
        "vdl_drop_column": (i32, [vp, cp]),
        "vdl_column_info": (i32, [vp, cp, P(i32), P(i64), P(vp)]),
        "vdl_download_column": (i32, [vp, cp, vp, ctypes.c_size_t]),
        "vdl_parse": (i32, [vp, cp, ctypes.c_size_t, P(vp)]),
        "vdl_plan_free": (None, [vp]),
        "vdl_plan_describe": (cp, [vp]),
        "vdl_plan_is_fused": (i32, [vp]),
        "vdl_plan_set_fusion": (i32, [vp, i32]),
        "vdl_plan_set_profiling": (i32, [vp, i32]),
        "vdl_plan_set_trace": (i32, [vp, i32]),
        "vdl_plan_set_jit": (i32, [vp, i32]),
        "vdl_plan_jit_note": (ctypes.c_char_p, [vp]),
        "vdl_plan_jit_check": (i32, [vp, vp]),
        "vdl_n_traced": (i32, [vp]),
        "vdl_traced": (i32, [vp, i32, P(i32), P(cp), P(i64), P(P(i64)), P(P(ctypes.c_uint8))]),
        "vdl_run": (i32, [vp, vp]),
        "vdl_n_outputs": (i32, [vp]),
        "vdl_output": (i32, [vp, i32, P(cp), P(cp), P(P(i64)), P(ctypes.c_size_t)]),
        "vdl_plan_set_device_outputs": (i32, [vp, i32]),
        "vdl_output_device": (i32, [vp, i32, P(P(i64)), P(ctypes.c_size_t)]),
        "vdl_n_timings": (i32, [vp]),
        "vdl_timing": (i32, [vp, i32, P(cp), P(ctypes.c_double)]),
        "vdl_plan_scan_stats": (i32, [vp, P(i64), P(i64), P(ctypes.c_double)]),
        "vdl_plan_scan_traffic": (i32, [vp, vp, P(i64), P(ctypes.c_char_p)]),
        "vdl_plan_partial_spec": (i32, [vp, P(i64), P(P(ctypes.c_int32))]),
        "vdl_plan_sharded_route": (i32, [vp, vp, P(ctypes.c_char_p), P(i32)]),
        "vdl_run_local": (i32, [vp, vp, vp]),
        "vdl_finalize": (i32, [vp, vp, vp]),
        "vdl_plan_set_row_offset": (i32, [vp, i64]),
        "vdl_plan_set_sharded_table": (i32, [vp, ctypes.c_char_p]),
        "vdl_resolve_first": (i32, [vp, vp, vp]),
        "vdl_exchange_spec": (i32, [vp, ctypes.c_char_p, P(i32)]),
        "vdl_exchange_begin": (i32, [vp, vp, i32, P(i64)]),
        "vdl_exchange_pack": (i32, [vp, vp, vp]),
        "vdl_exchange_finish": (i32, [vp, vp, vp, i64]),
        "vdl_finalize_begin": (i32, [vp, vp, vp, i32]),
        "vdl_finalize_end": (i32, [vp, vp, i32]),
    }
    sig.update({
        "vdl_comm_unique_id": (i32, [vp]),
        "vdl_comm_init": (i32, [vp, i32, i32, vp]),
        "vdl_comm_init_host": (i32, [vp, i32, i32, P(CommHost)]),
        "vdl_comm_info": (i32, [vp, P(i32), P(i32), P(cp)]),
        "vdl_comm_free": (None, [vp]),
        "vdl_run_sharded": (i32, [vp, vp]),
        "vdl_run_sharded_begin": (i32, [vp, vp, i32]),
        "vdl_run_sharded_end": (i32, [vp, vp, i32]),
        "vdl_comm_merge_host": (i32, [i32, i64, P(ctypes.c_int32), P(i64), i64, P(i64), P(i64)]),
    })
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here = the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


ABI_SYMBOLS = [
    "vdl_open", "vdl_close", "vdl_last_error", "vdl_version", "vdl_set_stream", "vdl_use_own_stream", "vdl_register_column",
    "vdl_upload_column", "vdl_generate_column", "vdl_drop_column", "vdl_column_info", "vdl_download_column",
    "vdl_parse", "vdl_plan_free", "vdl_plan_describe", "vdl_plan_is_fused", "vdl_plan_set_fusion",
    "vdl_plan_set_profiling", "vdl_plan_set_jit", "vdl_plan_jit_note", "vdl_plan_jit_check", "vdl_plan_set_trace", "vdl_n_traced", "vdl_traced", "vdl_run", "vdl_n_outputs", "vdl_output", "vdl_plan_set_device_outputs", "vdl_output_device", "vdl_n_timings", "vdl_timing",
    "vdl_plan_scan_stats", "vdl_plan_scan_traffic", "vdl_plan_partial_spec", "vdl_plan_sharded_route", "vdl_run_local", "vdl_finalize", "vdl_finalize_begin", "vdl_finalize_end", "vdl_plan_set_row_offset", "vdl_plan_set_sharded_table", "vdl_resolve_first", "vdl_exchange_spec", "vdl_exchange_begin", "vdl_exchange_pack",
    "vdl_exchange_finish", "vdl_comm_unique_id", "vdl_comm_init", "vdl_comm_init_host", "vdl_comm_info", "vdl_comm_free", "vdl_run_sharded",
    "vdl_run_sharded_begin", "vdl_run_sharded_end", "vdl_comm_merge_host",
]
