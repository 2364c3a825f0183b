"""Autograd glue for the training path (scope row f1): torch.autograd.Functions whose forward AND backward are the HIP
kernels of this package.  ``ops.linear`` / ``ops.layer_norm`` route here whenever autograd is recording; the attention op
carries its own registration (``ops.graph_transformer_attention``).

Linear backward (reference: autograd of torch.nn.Linear as instantiated by layer_kernels, layers/utils.py:107-121):
    dX = dZ W          -> ops.linear(dZ, W^T)              (same MFMA kernels; the weight transpose is a small copy)
    dW = dZ^T X        -> ops.linear(dZ^T, X^T)            (the reduction runs over the rows: both operands are transposed
                                                             into K-contiguous, zero-padded [*, N_pad] buffers first)
    db = column sums of dZ (deterministic two-stage reduction), dZ = dY * gelu'(pre) when GELU was fused (the
    pre-activation is recomputed by one extra GEMM instead of being stored by the forward).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops


def _t_padded(a: Tensor, mult: int = 64) -> Tensor:
    """[N, C] -> contiguous [C, N_pad] (zero-padded so that the reduction length suits the MFMA path)."""
    n, c = a.shape
    n_pad = (n + mult - 1) // mult * mult
    out = a.new_zeros((c, n_pad))
    out[:, :n] = a.t()
    return out


class LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b) + residual."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], act: Optional[str], residual: Optional[Tensor]):
        ctx.act = act
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(x, weight, bias)
        return ops._linear_fwd(x, weight, bias, act=act, residual=residual)

    @staticmethod
    def backward(ctx, d_y: Tensor):
        x, weight, bias = ctx.saved_tensors
        d_y = d_y.contiguous()
        dz = d_y
        if ctx.act == "gelu":
            pre = ops._linear_fwd(x, weight, bias)
            dz = ops.gelu_backward(pre, d_y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops._linear_fwd(dz, weight.t().contiguous())
        if ctx.needs_input_grad[1]:
            dw = ops._linear_fwd(_t_padded(dz), _t_padded(x.reshape(-1, x.shape[-1]))).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dz).to(bias.dtype)
        return dx, dw, db, None, (d_y if ctx.has_res and ctx.needs_input_grad[4] else None)


class LayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], eps: float):
        ctx.eps, ctx.has_bias = eps, bias is not None
        ctx.save_for_backward(x, weight)
        return ops._layer_norm_fwd(x, weight, bias, eps)

    @staticmethod
    def backward(ctx, d_y: Tensor):
        x, weight = ctx.saved_tensors
        need_p = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx, dg, db = ops.layer_norm_backward(d_y.contiguous(), x, weight, ctx.eps, need_param_grads=need_p)
        return (dx if ctx.needs_input_grad[0] else None, dg.to(weight.dtype) if ctx.needs_input_grad[1] else None,
                db.to(weight.dtype) if ctx.has_bias and ctx.needs_input_grad[2] else None, None)


def attention(q: Tensor, k: Tensor, v: Tensor, e: Tensor, csc: "ops.CSC", num_heads: int, reverse) -> Tensor:
    """Differentiable edge attention on [rows, H*C] tensors through the op mirror (materialised E, CSC order)."""
    H = num_heads
    C = q.shape[1] // H
    rowptr, edge_ids, edge_dst = reverse
    out, _saved, _m = ops.graph_transformer_attention(q.reshape(-1, H, C), k.reshape(-1, H, C), v.reshape(-1, H, C), e.reshape(-1, H, C),
                                                      csc.row, csc.colptr, rowptr, edge_ids, edge_dst)
    return out.reshape(-1, H * C)
