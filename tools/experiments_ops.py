"""Host side of the EXPERIMENTS build's extra entry points (csrc/experiments/anemoi_hip_experiments.h): the round-4 layer chain
``anemoi_gt_chain_fwd`` - superseded by the role-split kernel (``ops.gt_layer_chain2``) and not part of the product library.
Use with ``python -m anemoi_core_amd.build --experiments`` and ``ANEMOI_HIP_LIB=anemoi_core_amd/lib/libanemoi_hip_exp.so``."""
from typing import Optional

import torch
from torch import Tensor

from anemoi_core_amd import _lib
from anemoi_core_amd.ops import CHAIN_CHANNELS, _dev, _dt, _rows, _stream, _vec


class _ChainArgs(_lib.C.Structure):
    _p, _i64, _i32, _f = _lib.C.c_void_p, _lib.C.c_int64, _lib.C.c_int32, _lib.C.c_float
    _fields_ = [("attn", _p), ("ld_attn", _i64), ("x_res", _p), ("ld_x", _i64), ("wp", _p), ("bp", _p),
                ("ln1_w", _p), ("ln1_b", _p), ("ln1_eps", _f), ("w1", _p), ("b1", _p), ("hidden", _i32), ("w2", _p), ("b2", _p),
                ("extra", _p), ("ld_extra", _i64), ("x_out", _p), ("ld_out", _i64),
                ("lnq_w", _p), ("lnq_b", _p), ("lnq_eps", _f), ("wq", _p), ("bq", _p), ("q_out_features", _i32),
                ("q_out", _p), ("ld_q", _i64), ("n_rows", _i32), ("channels", _i32), ("rows_per_tile", _i32), ("timeline", _p)]



def gt_layer_chain(attn: Tensor, x_res: Tensor, wp: Tensor, bp: Tensor, ln1_w: Tensor, ln1_b: Optional[Tensor], ln1_eps: float,
                   w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, *, extra: Optional[Tensor] = None,
                   lnq_w: Optional[Tensor] = None, lnq_b: Optional[Tensor] = None, lnq_eps: float = 1e-5,
                   wq: Optional[Tensor] = None, bq: Optional[Tensor] = None, rows_per_tile: int = 0, timeline: Optional[Tensor] = None):
    """The row-local part of a GraphTransformer block in ONE launch (anemoi_gt_chain_fwd, csrc/gt_chain.hip):

        x1 = attn Wp^T + bp + x_res;  h = GELU(LN(x1; ln1) W1^T + b1);  x_out = h W2^T + b2 + x1 [+ extra]
        q_out = LN(x_out; lnq) Wq^T + bq        (optional: the NEXT block's LayerNorm + fused q|k|v|self projection)

    ``wp, w1, w2, wq`` are fragment-major images (``pack_weight_frag``), biases / LayerNorm vectors in the model dtype.
    Returns ``x_out`` or ``(x_out, q_out)``.  Inference only (no autograd)."""
    _dev(attn, x_res, wp, bp, ln1_w, ln1_b, w1, b1, w2, b2, extra, lnq_w, lnq_b, wq, bq)
    N, D = attn.shape
    dt = attn.dtype
    hidden = b1.shape[0]
    if D != CHAIN_CHANNELS or dt not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"gt_layer_chain: {D} channels / {dt} (built for {CHAIN_CHANNELS} channels, 16-bit dtypes)")
    if tuple(x_res.shape) != (N, D) or (extra is not None and tuple(extra.shape) != (N, D)):
        raise ValueError("gt_layer_chain: attn, x_res and extra must have the same [N, channels] shape")
    for name, w, numel in (("wp", wp, D * D), ("w1", w1, hidden * D), ("w2", w2, D * hidden)):
        if w.dim() != 1 or w.numel() != numel or w.dtype != dt or not w.is_contiguous():
            raise ValueError(f"gt_layer_chain: {name} must be the contiguous fragment-major image ({numel} x {dt}) made by pack_weight_frag")
    q_out_f = 0
    if wq is not None:
        if bq is None or lnq_w is None:
            raise ValueError("gt_layer_chain: the trailing projection needs wq, bq and lnq_w")
        q_out_f = bq.shape[0]
        if wq.dim() != 1 or wq.numel() != q_out_f * D or wq.dtype != dt or not wq.is_contiguous():
            raise ValueError("gt_layer_chain: wq must be the contiguous fragment-major image made by pack_weight_frag")
    x_out = torch.empty((N, D), dtype=dt, device=attn.device)
    q_out = torch.empty((N, q_out_f), dtype=dt, device=attn.device) if q_out_f else None
    (ap, lda), (xp, ldx), (ep, lde) = _rows(attn, "attn", dt), _rows(x_res, "x_res", dt), _rows(extra, "extra", dt)
    a = _ChainArgs(ap, lda, xp, ldx, wp.data_ptr(), _vec(bp, "bp", D, dt), _vec(ln1_w, "ln1_w", D, dt), _vec(ln1_b, "ln1_b", D, dt), float(ln1_eps),
                   w1.data_ptr(), _vec(b1, "b1", hidden, dt), hidden, w2.data_ptr(), _vec(b2, "b2", D, dt), ep, lde, x_out.data_ptr(), D,
                   _vec(lnq_w, "lnq_w", D, dt), _vec(lnq_b, "lnq_b", D, dt), float(lnq_eps), 0 if wq is None else wq.data_ptr(),
                   _vec(bq, "bq", q_out_f, dt) if q_out_f else 0, q_out_f, 0 if q_out is None else q_out.data_ptr(), q_out_f, N, D, int(rows_per_tile),
                   0 if timeline is None else timeline.data_ptr())
    _lib.check(_lib.load().anemoi_gt_chain_fwd(_lib.C.byref(a), _dt(attn), _stream()), "gt_chain_fwd")
    return x_out if q_out is None else (x_out, q_out)
