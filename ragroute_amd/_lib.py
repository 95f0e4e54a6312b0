"""ctypes binding of libragroute_hip.so (C ABI in include/ragroute_hip.h).

There is no CPU fallback: if the HIP library is missing or fails, every operator raises.
"""
import ctypes
import os

from ._build import LIB_PATH

RR_DTYPE_F16, RR_DTYPE_BF16 = 0, 1
RR_MAX_K = 1024
RR_QUERY_BLOCK = 256
RR_MAX_SEGMENTS = 32
RR_SEGMENT_ALIGN = 256

EXPORTS = ("rr_version", "rr_last_error", "rr_device_cus", "rr_padded_dim", "rr_l2_normalize_f32", "rr_rows_to_half",
           "rr_flat_search_workspace_bytes", "rr_flat_search", "rr_merge_topk", "rr_router_mlp", "rr_profile_begin",
           "rr_profile_end", "rr_centroid", "rr_flat_search_l2", "rr_half_sqnorms", "rr_screen_dim", "rr_screen_build",
           "rr_flat_search_screened_workspace_bytes", "rr_flat_search_screened", "rr_router_workspace_bytes", "rr_router_mlp_ws",
           "rr_build_flags", "rr_merge_topk_gathered", "rr_flat_scan_kernel_name", "rr_flat_search_segments", "rr_flat_search_workspace_bytes_for")


class RouterWeightsStruct(ctypes.Structure):
    """struct rr_router_weights"""
    _fields_ = [
        ("n_sources", ctypes.c_int32), ("d_max", ctypes.c_int32), ("n_models", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("model_of_source", ctypes.c_void_p), ("w1q", ctypes.c_void_p), ("c1", ctypes.c_void_p),
        ("ln1_g", ctypes.c_void_p), ("ln1_b", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p),
        ("ln2_g", ctypes.c_void_p), ("ln2_b", ctypes.c_void_p), ("w3", ctypes.c_void_p),
        ("b3", ctypes.c_float), ("prob_threshold", ctypes.c_float), ("ln_eps", ctypes.c_float), ("reserved2", ctypes.c_float),
    ]


class SegmentStruct(ctypes.Structure):
    """struct rr_segment"""
    _fields_ = [("row_begin", ctypes.c_int64), ("n_rows", ctypes.c_int64), ("id_offset", ctypes.c_int64),
                ("mask_col", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class RagrouteHipError(RuntimeError):
    pass


_LIB = None


def lib():
    """Load the HIP library; raise loudly if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RagrouteHipError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        L = ctypes.CDLL(os.environ.get("RR_LIB_OVERRIDE", LIB_PATH))  # RR_LIB_OVERRIDE: A/B builds during development
        vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
        L.rr_version.restype = i32
        L.rr_last_error.restype = ctypes.c_char_p
        L.rr_device_cus.restype = i32
        L.rr_padded_dim.argtypes = [i32]
        L.rr_l2_normalize_f32.argtypes = [vp, i64, i64, vp]
        L.rr_rows_to_half.argtypes = [vp, i64, i64, i64, vp, i32, i64, i32, vp]
        L.rr_flat_search_workspace_bytes.argtypes = [i32]
        L.rr_flat_search_workspace_bytes.restype = sz
        L.rr_flat_search_workspace_bytes_for.argtypes = [i32, i32]
        L.rr_flat_search_workspace_bytes_for.restype = sz
        L.rr_flat_search.argtypes = [vp, i32, i64, i32, vp, i32, i32, vp, vp, i64, vp, sz, vp, i64, vp]
        L.rr_merge_topk.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
        L.rr_flat_search_segments.argtypes = [vp, i32, i64, i32, ctypes.POINTER(SegmentStruct), i32, vp, i32, i32, vp, vp, vp, sz, vp, i64, vp]
        L.rr_merge_topk_gathered.argtypes = [vp, i32, sz, sz, i32, i32, i32, i32, i32, vp, vp, vp]
        L.rr_build_flags.restype = ctypes.c_char_p
        L.rr_flat_scan_kernel_name.argtypes = [i32, i32]
        L.rr_flat_scan_kernel_name.restype = ctypes.c_char_p
        L.rr_router_mlp.argtypes = [ctypes.POINTER(RouterWeightsStruct), vp, i32, vp, vp, vp]
        L.rr_router_workspace_bytes.argtypes = [ctypes.POINTER(RouterWeightsStruct), i32]
        L.rr_router_workspace_bytes.restype = sz
        L.rr_router_mlp_ws.argtypes = [ctypes.POINTER(RouterWeightsStruct), vp, i32, vp, vp, vp, sz, vp]
        L.rr_flat_search_l2.argtypes = [vp, vp, i32, i64, i32, vp, i32, i32, vp, vp, i64, vp, sz, vp, i64, vp]
        L.rr_half_sqnorms.argtypes = [vp, i32, i64, i32, vp, vp]
        L.rr_centroid.argtypes = [vp, i32, i64, i32, i32, vp, vp]
        L.rr_screen_dim.argtypes = [i32]
        L.rr_screen_build.argtypes = [vp, i32, i64, i32, vp, vp, vp]
        L.rr_flat_search_screened_workspace_bytes.argtypes = [i32, i32, i32, i32]
        L.rr_flat_search_screened_workspace_bytes.restype = sz
        L.rr_flat_search_screened.argtypes = [vp, i32, vp, vp, i64, i32, vp, i32, i32, i32, vp, vp, i64, vp, vp, sz, vp, i64, vp]
        L.rr_profile_begin.argtypes = [i32]
        L.rr_profile_end.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
        _LIB = L
    return _LIB


def check(status, what):
    if status != 0:
        msg = lib().rr_last_error().decode("utf-8", "replace")
        raise RagrouteHipError(f"{what} failed (rr_status {status}): {msg}")
