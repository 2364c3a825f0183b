"""The TORCH_LIBRARY layer over the C ABI (csrc/torch_binding.cpp -> lib/libanemoi_torch.so): the hot forward ops as
``torch.ops.anemoi_hip.*``, taking tensors and the current HIP stream in C++ instead of ~10 us of ctypes marshalling per launch.

Same kernels, same C ABI underneath, bit-identical results (tests/test_torch_ext_gpu.py).  ``ANEMOI_TORCH_EXT=0`` keeps every
call on the ctypes binding (for A/B runs).  Built by ``python -m anemoi_core_amd.build`` together with libanemoi_hip.so; a
missing library is an error, like a missing libanemoi_hip.so - there is no silent fallback."""
from __future__ import annotations

import os
import threading

import torch

from . import _lib

EXT_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libanemoi_torch.so")
# an alternative build of the kernel library (ANEMOI_HIP_LIB, same-box A/Bs) is only reachable through ctypes: the extension is
# linked against THE library next to it
ENABLED = os.environ.get("ANEMOI_TORCH_EXT", "1") != "0" and not os.environ.get("ANEMOI_HIP_LIB")

_lock = threading.Lock()
_ops = None


def ops():
    """``torch.ops.anemoi_hip`` (loading the library on first use), or None when switched off."""
    global _ops
    if not ENABLED:
        return None
    if _ops is not None:
        return _ops
    with _lock:
        if _ops is None:
            _lib.load()  # libanemoi_hip.so first: the extension links against it
            if not os.path.exists(EXT_PATH):
                raise _lib.HipLibraryError(f"{EXT_PATH} not found: build it with `python -m anemoi_core_amd.build` "
                                           "(or set ANEMOI_TORCH_EXT=0 to stay on the ctypes binding). anemoi_core_amd has no CPU / eager fallback.")
            torch.ops.load_library(EXT_PATH)
            _register_fakes()
            _ops = torch.ops.anemoi_hip
    return _ops


def _register_fakes() -> None:
    """Meta / FakeTensor kernels of ``torch.ops.anemoi_hip.*`` (shapes and dtypes only): the module path traces under
    ``torch.compile`` / ``torch.export`` / FakeTensorMode with the ops as opaque leaves - what the reference does for its Triton op
    with ``register_fake`` (models/src/anemoi/models/triton/gt.py:431-447, 553-556).  Output shapes are exactly those of
    csrc/torch_binding.cpp; the two LayerNorm-fold ops report the 'shape not eligible' sentinel (1-D empty) by the same shape rule."""
    fake = torch.library.register_fake

    @fake("anemoi_hip::linear")
    def _(x, weight, bias, act, residual, x2, g1, idx1, g2, idx2):
        return x.new_empty((x.shape[0], weight.shape[0]))

    @fake("anemoi_hip::linear_out")
    def _(x, weight, bias, act, residual, x2, g1, idx1, g2, idx2, out):
        return None

    @fake("anemoi_hip::layer_norm")
    def _(x, weight, bias, eps, residual):
        return x.new_empty(x.shape)

    @fake("anemoi_hip::layer_norm_out")
    def _(x, weight, bias, eps, residual, out):
        return None

    @fake("anemoi_hip::gt_attention_fused_edge")
    def _(q, k, v, edge_feat, w_packed, row, colptr, order, n_src, num_heads, addend, return_lse):
        n_dst = q.shape[0]
        return q.new_empty((n_dst, q.shape[1])), q.new_empty((n_dst if return_lse else 0, num_heads), dtype=torch.float32)

    @fake("anemoi_hip::linear_with_row_stats")
    def _(x, weight, bias, residual):
        N, K, O = x.shape[0], x.shape[1], weight.shape[0]
        if x.dtype == torch.float32 or O % 64 or K % 64:
            return x.new_empty((0,)), x.new_empty((0,), dtype=torch.float32)
        return x.new_empty((N, O)), x.new_empty((N, O // 64, 2), dtype=torch.float32)

    @fake("anemoi_hip::linear_ln_folded")
    def _(x, w_scaled, c, d, stats, eps, act):
        return x.new_empty((x.shape[0], w_scaled.shape[0]))
