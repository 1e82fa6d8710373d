"""ctypes binding of the C ABI in ``include/hgnn_hip.h`` (libhgnn_hip.so).

The library is built in-tree by ``hierarchicalgnn_amd.build`` (hipcc,
``--offload-arch=gfx950``).  There is deliberately NO fallback: if the shared
object is missing, or a call returns a non-zero status, a ``RuntimeError`` is
raised.  PyTorch is only used for device memory and the stream handle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# HGNN_LIB: A/B measurements of two builds of the SAME sources with different tuning constants (tools/); never a fallback
LIB_PATH = os.environ.get("HGNN_LIB") or os.path.join(_HERE, "csrc", "libhgnn_hip.so")

HGNN_OK = 0
CNT_WORK, CNT_SPLIT, CNT_PARTIAL, CNT_ERR, CNT_VALID, CNT_UNSORTED = 0, 1, 2, 3, 4, 5
ABI_VERSION = 23
MLP_BWD_BLOCKS = 512   # HGNN_MLP_BWD_BLOCKS
GMM_STATE, GMM_BLOCKS = 16, 1024   # HGNN_GMM_STATE, HGNN_GMM_BLOCKS
LN_ACT_BLOCKS = 1024   # HGNN_LN_ACT_BLOCKS


class HgnnPlan(Structure):
    """mirror of ``struct hgnn_plan``"""
    _fields_ = [
        ("n_rows", c_int64), ("n_dst", c_int64), ("n_src", c_int64),
        ("chunk", c_int32), ("has_gather", c_int32),
        ("max_work", c_int64), ("max_split", c_int64), ("max_partial", c_int64),
        ("perm", c_void_p), ("src_row", c_void_p), ("dst32", c_void_p), ("rowptr", c_void_p),
        ("wi_begin", c_void_p), ("wi_end", c_void_p), ("wi_target", c_void_p), ("wi_dst", c_void_p),
        ("split_dst", c_void_p), ("split_pbegin", c_void_p), ("counts", c_void_p),
    ]


class HgnnMlpDesc(Structure):
    """mirror of ``struct hgnn_mlp_desc``"""
    _fields_ = [
        ("n_seg", c_int32),
        ("seg_table", c_void_p * 3), ("seg_index", c_void_p * 3), ("seg_width", c_int32 * 3),
        ("n_layers", c_int32),
        ("W", c_void_p * 3), ("b", c_void_p * 3), ("ln_w", c_void_p * 3), ("ln_b", c_void_p * 3),
        ("width", c_int32 * 4), ("act", c_int32 * 3),
        ("ln_eps", c_float),
        ("skip", c_void_p),
        ("M", c_int64),
        ("w0_cols", c_int32),
        ("w_last_rows", c_int32),
        ("save_pre", c_void_p * 3),
        ("n_pre", c_int32),
        ("pre_table", c_void_p * 2), ("pre_index", c_void_p * 2),
    ]


_SIGNATURES = {
    "hgnn_abi_version": (c_int, []),
    "hgnn_last_error": (c_char_p, []),
    "hgnn_set_option": (c_int, [c_char_p, c_int]),
    "hgnn_sizeof_plan": (c_int, []),
    "hgnn_sizeof_mlp_desc": (c_int, []),
    "hgnn_plan_dims": (c_int, [c_int64, c_int64, c_int64, c_int32, POINTER(HgnnPlan)]),
    "hgnn_plan_workspace_bytes": (c_int, [c_int64, c_int64, POINTER(c_size_t)]),
    "hgnn_plan_build": (c_int, [c_void_p, c_void_p, POINTER(HgnnPlan), c_void_p, c_size_t, c_void_p]),
    "hgnn_segment_reduce_f32": (c_int, [POINTER(HgnnPlan), c_void_p, c_int32, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "hgnn_gather_rows_f32": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "hgnn_spread_rows_f32": (c_int, [POINTER(HgnnPlan), c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "hgnn_segment_reduce_bf16": (c_int, [POINTER(HgnnPlan), c_void_p, c_int32, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "hgnn_spread_rows_bf16": (c_int, [POINTER(HgnnPlan), c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "hgnn_gather_rows_bf16": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p]),
    "hgnn_edge_dot_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                  c_int64, c_void_p, c_void_p]),
    "hgnn_index_to_i32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "hgnn_knn_radius_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p,
                                    c_void_p, c_void_p]),
    "hgnn_knn_workspace_bytes": (c_int, [c_int64, c_int64, c_int32, POINTER(c_size_t)]),
    "hgnn_knn_radius_ws_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hgnn_gmm2_fit_f32": (c_int, [c_void_p, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "hgnn_gmm2_cut_f32": (c_int, [c_void_p, c_float, c_int32, c_float, c_void_p, c_void_p]),
    "hgnn_cc_labels": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "hgnn_ln_act_forward_bf16": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_float,
                                         c_void_p, c_void_p]),
    "hgnn_ln_act_backward_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32,
                                          c_float, c_void_p, c_void_p, c_void_p]),
    "hgnn_wgrad_workspace_bytes": (c_int, [c_int64, c_int32, c_int32, POINTER(c_size_t)]),
    "hgnn_wgrad_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_int64,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "hgnn_wgrad_f32_split3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_int64,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "hgnn_mlp_backward_layer_supported_bf16": (c_int, [c_int32, c_int32]),
    "hgnn_mlp_backward_layer_bf16": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hgnn_mlp_supported": (c_int, [POINTER(HgnnMlpDesc)]),
    "hgnn_mlp_forward_f32": (c_int, [POINTER(HgnnMlpDesc), c_void_p, c_void_p]),
    "hgnn_mlp_supported_bf16": (c_int, [POINTER(HgnnMlpDesc)]),
    "hgnn_mlp_forward_bf16": (c_int, [POINTER(HgnnMlpDesc), c_void_p, c_void_p]),
    "hgnn_mlp_supported_bf16_split": (c_int, [POINTER(HgnnMlpDesc)]),
    "hgnn_mlp_forward_bf16_split": (c_int, [POINTER(HgnnMlpDesc), c_void_p, c_void_p]),
    "hgnn_mlp_supported_f32_split3": (c_int, [POINTER(HgnnMlpDesc)]),
    "hgnn_mlp_forward_f32_split3": (c_int, [POINTER(HgnnMlpDesc), c_void_p, c_void_p]),
    "hgnn_linear_f32_split3": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "hgnn_project_f32_split3": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                        c_void_p]),
    "hgnn_ln_act_forward_f32": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_float,
                                        c_void_p, c_void_p]),
    "hgnn_ln_act_backward_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32,
                                         c_float, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libhgnn_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"hierarchicalgnn_amd: {LIB_PATH} is missing. Build it with "
            "`python -m hierarchicalgnn_amd.build` (needs hipcc; gfx950). "
            "There is no CPU / eager fallback for the HIP kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def declared_symbols():
    """names the binding expects (kept in sync with include/hgnn_hip.h by a test)"""
    return sorted(_SIGNATURES)


def register(name, restype, argtypes):
    _SIGNATURES[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype = restype
        fn.argtypes = argtypes


def check(rc: int, what: str = "") -> None:
    if rc != HGNN_OK:
        msg = load().hgnn_last_error()
        raise RuntimeError(f"libhgnn_hip {what} failed (status {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a tensor (or NULL)"""
    return c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else c_void_p(0)


def current_stream(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
