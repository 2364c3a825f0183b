"""ctypes binding of libanemoi_hip.so (the C ABI declared in include/anemoi_hip.h).

The library is mandatory: there is no CPU or eager-PyTorch fallback anywhere in the product path.  A missing
or unloadable library raises ``HipLibraryError`` on first use.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

ABI_VERSION = 13
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libanemoi_hip.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_GELU = 0, 1
E_INVALID, E_UNSUPPORTED, E_LAUNCH = -1, -2, -3

_p, _i64, _i32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_float

# name -> argtypes, exactly as in include/anemoi_hip.h
SIGNATURES = {
    "anemoi_hip_abi_version": ([], C.c_int),
    "anemoi_hip_last_error": ([], C.c_char_p),
    "anemoi_gt_attention_fwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gt_attention_bwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p,
                                 _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gt_attention_dropout_fwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _i32, _f, C.c_uint64,
                                         C.c_int, _p], C.c_int),
    "anemoi_gt_attention_dropout_bwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p,
                                         _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, _f, C.c_uint64, C.c_int, _p], C.c_int),
    "anemoi_attention_dropout_mask": ([_p, _i32, _i32, _f, C.c_uint64, _p], C.c_int),
    "anemoi_gt_attention_fused_edge_fwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i32, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _i32, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_pack_edge_weights": ([_p, _p, _p, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_pack_edge_features": ([_p, _i64, _p, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_reduce_workspace_bytes": ([_i32], _i64),
    "anemoi_layernorm_bwd": ([_p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _p, _i32, _i32, _f, C.c_int, _p], C.c_int),
    "anemoi_colsum": ([_p, _i64, _p, _p, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gelu_fwd": ([_p, _i64, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gelu_bwd": ([_p, _i64, _p, _i64, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_cond_layernorm_bwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _f, C.c_int, _p], C.c_int),
    "anemoi_cond_layernorm_fwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _f, C.c_int, _p], C.c_int),
    "anemoi_assemble_input": ([_p, _i64, _i64, _i32, _i32, _p, _i64, _i32, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_assemble_output": ([_p, _i64, _p, _i64, _p, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_assemble_input_norm": ([_p, C.c_int, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _i32, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_assemble_output_norm": ([_p, _i64, C.c_int, _p, _i64, _p, _p, _p, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_affine_columns": ([_p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_bound_columns": ([_p, _i64, _i32, _i32, _p, _p, _i32, C.c_int, _p], C.c_int),
    "anemoi_layernorm_fwd": ([_p, _i64, _p, _p, _p, _i64, _p, _i64, _i32, _i32, _f, C.c_int, _p], C.c_int),
    "anemoi_gt_attention_fused_edge_bwd_partial_floats": ([_i32, _i32, _i32], _i64),
    "anemoi_gt_attention_fused_edge_bwd": ([_p, _i64, _p, _i64, _p, _i64, _p, _i32, _p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p,
                                            _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _i64,
                                            _i32, _i32, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_linear_wgrad_workspace_bytes": ([_i32, _i32, _i32], _i64),
    "anemoi_linear_wgrad": ([_p, _i64, _p, _i64, _p, _i64, _p, _p, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_linear_splitk_f32": ([_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_linear_stats_fwd": ([_p, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_linear_lnfold_fwd": ([_p, _i64, _i32, _p, _i64, _p, _p, _p, _i32, _f, C.c_int, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_linear_fwd": ([_p, _i64, _i32, _p, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i64, _i32, _i32, C.c_int, C.c_int, _p], C.c_int),
    "anemoi_linear_fwd_pre": ([_p, _i64, _i32, _p, _i64, _i32, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _i32, _i32,
                               C.c_int, C.c_int, _p], C.c_int),
    "anemoi_edge_ln_residual_segment_sum_fwd": ([_p, _i64, _p, _i64, _p, _p, _f, _p, _p, _i64, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_segment_sum_rows": ([_p, _i64, _p, _p, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gather_add_rows": ([_p, _i64, _p, _i64, _p, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_transpose_pad": ([_p, _i64, _p, _i64, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_glu_fwd": ([_p, _i64, _p, _i64, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_glu_bwd": ([_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gather_rows": ([_p, _i64, _p, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_peer_alloc": ([C.POINTER(_p), _i64, _i32], C.c_int),
    "anemoi_peer_free": ([_p], C.c_int),
    "anemoi_peer_export": ([_p, _p], C.c_int),
    "anemoi_peer_open": ([_p, C.POINTER(_p)], C.c_int),
    "anemoi_peer_close": ([_p], C.c_int),
    "anemoi_gt_chain_rows_per_tile": ([_i32], C.c_int),
    "anemoi_gt_chain2_fwd": ([_p, C.c_int, _p], C.c_int),
    "anemoi_gt_rowchain_fwd": ([_p, C.c_int, _p], C.c_int),
    "anemoi_gt_cluster_chain_fwd": ([_p, C.c_int, _p], C.c_int),
    "anemoi_gt_cluster_chain_workspace_bytes": ([], _i64),
    "anemoi_gnn_edge_chain_fwd": ([_p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gnn_mlp_chain_fwd": ([_p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_gnn_node_chain_segsum_fwd": ([_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _p, _p, _i32, _p, _i64, _i32, _i32,
                                          C.c_int, _p], C.c_int),
    "anemoi_gnn_node_chain_fwd": ([_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _p, _p, _i32, _p, _i64, _i32, _i32, C.c_int, _p], C.c_int),
    "anemoi_peer_exchange_rows": ([_p, _i64, _p, _p, _i32, _i32, _i32, _p, _p, _i64, _p], C.c_int),
}


# entry points of the EXPERIMENTS build only (csrc/experiments/anemoi_hip_experiments.h; lib/libanemoi_hip_exp.so): bound when the loaded
# library has them, used by tools/ only
EXPERIMENT_SIGNATURES = {
    "anemoi_gt_chain_fwd": ([_p, C.c_int, _p], C.c_int),
    "anemoi_gnn_edge_chain_timeline": ([_p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _i32, _p, _p], C.c_int),
}


class HipLibraryError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


def load(path: str | None = None):
    """Load (once) and return the ctypes handle; raises HipLibraryError if the library is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = path or os.environ.get("ANEMOI_HIP_LIB", LIB_PATH)
        if not os.path.exists(path):
            raise HipLibraryError(
                f"{path} not found: build it with `python -m anemoi_core_amd.build` (hipcc, gfx950). "
                "anemoi_core_amd has no CPU / eager fallback."
            )
        try:
            lib = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise HipLibraryError(f"cannot load {path}: {e}") from e
        for name, (argtypes, restype) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise HipLibraryError(f"{path} does not export {name}; rebuild the library") from e
            fn.argtypes = argtypes
            fn.restype = restype
        for name, (argtypes, restype) in EXPERIMENT_SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.argtypes = argtypes
                fn.restype = restype
        got = lib.anemoi_hip_abi_version()
        if got != ABI_VERSION:
            raise HipLibraryError(f"{path}: ABI version {got}, expected {ABI_VERSION}; rebuild the library")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().anemoi_hip_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: {msg} (code {rc})")
