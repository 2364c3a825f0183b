"""Autograd glue for the training path (scope row f1): torch.autograd.Functions whose forward AND backward are the HIP
kernels of this package.  ``ops.linear`` / ``ops.layer_norm`` route here whenever autograd is recording; the attention op
carries its own registration (``ops.graph_transformer_attention``).

Linear backward (reference: autograd of torch.nn.Linear as instantiated by layer_kernels, layers/utils.py:107-121):
    dX = dZ W          -> ops.linear(dZ, W^T)              (same MFMA kernels; the weight transpose is a small copy)
    dW = dZ^T X        -> ops.linear_wgrad(dZ, X)          (the reduction runs over the rows: row-major tiles are read with the
                                                             LDS transpose read; fp32 / odd widths: explicit transposes + GEMM)
    db = column sums of dZ (deterministic two-stage reduction), dZ = dY * gelu'(pre) when GELU was fused (the
    pre-activation is recomputed by one extra GEMM instead of being stored by the forward).
"""
from __future__ import annotations

import os

from typing import Optional

import torch
from torch import Tensor

from . import ops


def _t_padded(a: Tensor, mult: int = 64) -> Tensor:
    """[N, C] -> contiguous [C, N_pad] (zero-padded so that the reduction length suits the MFMA path)."""
    return ops.transpose_pad(a, mult)


_SAVE_PRE = os.environ.get("ANEMOI_SAVE_PREACT", "1") == "1"  # 0: recompute GELU's argument in backward (saves [N, O] per layer)


def _granule_rows(t: Tensor) -> Tensor:
    """Rows that start on 16-byte boundaries (what the LDS-DMA granules of the wgrad kernel need), copying only if necessary."""
    ok = t.stride(1) == 1 and (t.shape[0] == 1 or t.stride(0) % 8 == 0) and t.data_ptr() % 16 == 0
    return t if ok else t.contiguous()


def _tn_ok(dz: Tensor, x: Tensor) -> bool:
    return ops.linear_wgrad_eligible(dz, x) and os.environ.get("ANEMOI_WGRAD_TN", "1") == "1"


def _weight_grad(dz: Tensor, x: Tensor) -> Tensor:
    """dW [O, K] = dz^T [O, N] @ x [N, K]: small output, reduction over all rows.  16-bit operands with O, K multiples of 8 go
    to the transpose-read kernel (csrc/wgrad.hip: no HBM transposes, deterministic split reduction); anything else is transposed
    into K-contiguous buffers for the forward GEMM kernels (split-K with fp32 atomics on the 16-bit path)."""
    n, o = dz.shape
    k = x.shape[1]
    if _tn_ok(dz, x):
        return ops.linear_wgrad(_granule_rows(dz), _granule_rows(x))
    if dz.dtype == torch.float32 or k % 4:
        return ops._linear_fwd(_t_padded(dz), _t_padded(x))
    tiles = ((o + 63) // 64) * ((k + 127) // 128)
    splits = max(1, min(((n + 63) // 64) // 4, (512 + tiles - 1) // tiles))
    return ops.linear_splitk(_t_padded(dz, 64 * splits), _t_padded(x, 64 * splits), splits)


class LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b + g1[idx1] + g2[idx2]) + residual.  segN = (ptr, ids): output rows grouped by idxN (adjoint of the
    gather = deterministic segment sum)."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], act: Optional[str], residual: Optional[Tensor],
                g1: Optional[Tensor] = None, idx1: Optional[Tensor] = None, seg1=None, g2: Optional[Tensor] = None,
                idx2: Optional[Tensor] = None, seg2=None):
        ctx.act = act
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.seg1, ctx.seg2 = seg1, seg2
        pre = None
        if act == "gelu" and _SAVE_PRE:  # the GEMM stores the pre-activation next to the output: no recomputing GEMM in backward
            y, pre = ops._linear_fwd(x, weight, bias, act=act, residual=residual, g1=g1, idx1=idx1, g2=g2, idx2=idx2, want_pre=True)
        else:
            y = ops._linear_fwd(x, weight, bias, act=act, residual=residual, g1=g1, idx1=idx1, g2=g2, idx2=idx2)
        ctx.save_for_backward(x, weight, bias, g1, idx1, g2, idx2, pre)
        return y

    @staticmethod
    def backward(ctx, d_y: Tensor):
        x, weight, bias, g1, idx1, g2, idx2, pre = ctx.saved_tensors
        d_y = d_y.contiguous()
        dz = d_y
        if ctx.act == "gelu":
            if pre is None:  # shapes outside the DMA-ring kernels: recompute
                pre = ops._linear_fwd(x, weight, bias, g1=g1, idx1=idx1, g2=g2, idx2=idx2)
            dz = ops.gelu_backward(pre, d_y)
        dx = dw = db = dg1 = dg2 = None
        if ctx.needs_input_grad[0]:
            dx = ops._linear_fwd(dz, weight.t().contiguous())
        x2d = x.reshape(-1, x.shape[-1])
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and want_db and _tn_ok(dz, x2d):  # bias gradient rides along in the wgrad kernel
            dw, db = ops.linear_wgrad(_granule_rows(dz), _granule_rows(x2d), with_bias_grad=True)
            dw, db = dw.to(weight.dtype), db.to(bias.dtype)
            want_db = False
        elif ctx.needs_input_grad[1]:
            dw = _weight_grad(dz, x2d).to(weight.dtype)
        if want_db:
            db = ops.colsum(dz).to(bias.dtype)
        if g1 is not None and ctx.needs_input_grad[5]:
            dg1 = ops.segment_sum_rows(dz, *ctx.seg1)
        if g2 is not None and ctx.needs_input_grad[8]:
            dg2 = ops.segment_sum_rows(dz, *ctx.seg2)
        return (dx, dw, db, None, (d_y if ctx.has_res and ctx.needs_input_grad[4] else None), dg1, None, None, dg2, None, None)


class EdgeLnResidualSegmentSumFunction(torch.autograd.Function):
    """(e_new, agg) = (LayerNorm(z) + e_old, segment sum of e_new over the in-edges of every destination): GraphConv's
    ``edge_mlp(...).layer_norm + edge_attr`` and ``scatter(sum)`` (reference layers/conv.py:73-81)."""

    @staticmethod
    def forward(ctx, z: Tensor, e_old: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float, csc):
        ctx.eps, ctx.csc, ctx.has_ln, ctx.has_beta = eps, csc, gamma is not None, beta is not None
        ctx.save_for_backward(z, gamma)
        return ops._edge_ln_residual_segment_sum_fwd(z, e_old, gamma, beta, eps, csc)

    @staticmethod
    def backward(ctx, d_e_new: Optional[Tensor], d_agg: Optional[Tensor]):
        z, gamma = ctx.saved_tensors
        csc = ctx.csc
        if d_agg is None:
            total = d_e_new.contiguous()
        elif d_e_new is None:
            total = ops.gather_rows(d_agg.contiguous(), csc.dst)
        else:
            total = ops.gather_add_rows(d_e_new.contiguous(), d_agg.contiguous(), csc.dst)
        dz, dg, db = total, None, None
        if ctx.has_ln:
            need_p = ctx.needs_input_grad[2] or (ctx.has_beta and ctx.needs_input_grad[3])
            dz, dg, db = ops.layer_norm_backward(total, z, gamma, ctx.eps, need_param_grads=need_p)
            dg = dg.to(gamma.dtype) if ctx.needs_input_grad[2] else None
            db = db.to(gamma.dtype) if ctx.has_beta and ctx.needs_input_grad[3] else None
        return (dz if ctx.needs_input_grad[0] else None, total if ctx.needs_input_grad[1] else None, dg, db, None, None)


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
        if need_p:  # dgamma and dbeta are the two rows of one fp32 buffer: one cast for both
            gb = dg._base.to(weight.dtype)
            dg, db = gb[0], gb[1]
        return (dx if ctx.needs_input_grad[0] else None, dg if ctx.needs_input_grad[1] else None,
                db if ctx.has_bias and ctx.needs_input_grad[2] else None, None)


class CondLayerNormFunction(torch.autograd.Function):
    """y = LN(x) (scale + 1) + shift with per-row scale / shift."""

    @staticmethod
    def forward(ctx, x: Tensor, scale: Tensor, shift: Tensor, eps: float):
        ctx.eps = eps
        ctx.save_for_backward(x, scale)
        return ops._cond_layer_norm_fwd(x, scale, shift, eps)

    @staticmethod
    def backward(ctx, d_y: Tensor):
        x, scale = ctx.saved_tensors
        d_y = d_y.contiguous()
        dx, ds = ops.cond_layer_norm_backward(d_y, x, scale, ctx.eps)
        return dx, ds, d_y.reshape(ds.shape), None


class GluFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate_value: Tensor, kind: str):
        ctx.kind = kind
        ctx.save_for_backward(gate_value)
        return ops._glu_fwd(gate_value, kind)

    @staticmethod
    def backward(ctx, d_out: Tensor):
        (gv,) = ctx.saved_tensors
        return ops.glu_backward(gv, d_out.contiguous(), ctx.kind), None


class GeluFunction(torch.autograd.Function):
    """Stand-alone exact GELU (layer_kernels ``Activation``): forward ``anemoi_gelu_fwd``, backward ``anemoi_gelu_bwd``."""

    @staticmethod
    def forward(ctx, x: Tensor):
        ctx.save_for_backward(x)
        return ops._gelu_fwd(x)

    @staticmethod
    def backward(ctx, d_y: Tensor):
        (x,) = ctx.saved_tensors
        D = x.shape[-1]
        return ops.gelu_backward(x.reshape(-1, D), d_y.reshape(-1, D).contiguous()).view(x.shape)


class GatherRowsFunction(torch.autograd.Function):
    """out[i] = x[idx[i]]; backward: d_x[r] = sum of d_out rows with idx == r (fp32 accumulation)."""

    @staticmethod
    def forward(ctx, x: Tensor, idx: Tensor):
        ctx.n = x.shape[0]
        ctx.save_for_backward(idx)
        return ops._gather_rows_fwd(x, idx)

    @staticmethod
    def backward(ctx, d_out: Tensor):
        (idx,) = ctx.saved_tensors
        acc = torch.zeros((ctx.n, d_out.shape[1]), dtype=torch.float32, device=d_out.device)
        acc.index_add_(0, idx.long(), d_out.float())
        return acc.to(d_out.dtype), None


def attention(q: Tensor, k: Tensor, v: Tensor, e: Tensor, csc: "ops.CSC", num_heads: int, reverse) -> Tensor:
    """Differentiable edge attention on [rows, H*C] tensors through the op mirror (materialised E, CSC order)."""
    H = num_heads
    C = q.shape[1] // H
    rowptr, edge_ids, edge_dst = reverse
    out, _saved, _m = ops.graph_transformer_attention(q.reshape(-1, H, C), k.reshape(-1, H, C), v.reshape(-1, H, C), e.reshape(-1, H, C),
                                                      csc.row, csc.colptr, rowptr, edge_ids, edge_dst)
    return out.reshape(-1, H * C)


class AttentionConvFunction(torch.autograd.Function):
    """``GraphTransformerConv`` as one differentiable op (reference layers/conv.py:103-147): attention with a materialised edge
    tensor and, in training mode, dropout on the softmax weights.  The dropout mask is never stored: forward and backward derive
    it from the same (p, seed) (csrc/common.h: attn_dropout_scale)."""

    @staticmethod
    def forward(ctx, q, k, v, e, csc, num_heads, reverse, dropout_p, dropout_seed):
        out, lse = ops.gt_attention(q, k, v, e, csc, num_heads, return_lse=True, dropout_p=dropout_p, dropout_seed=dropout_seed)
        ctx.csc, ctx.H, ctx.reverse, ctx.drop = csc, num_heads, reverse, (dropout_p, dropout_seed)
        ctx.save_for_backward(q, k, v, e, out, lse)
        return out

    @staticmethod
    def backward(ctx, d_out):
        q, k, v, e, out, lse = ctx.saved_tensors
        reverse = ctx.reverse if ctx.reverse is not None else ops.build_reverse_csr(ctx.csc)
        dq, dk, dv, de = ops.gt_attention_backward(d_out.contiguous(), q, k, v, e, out, lse, ctx.csc, reverse, ctx.H,
                                                   dropout_p=ctx.drop[0], dropout_seed=ctx.drop[1])
        return dq, dk, dv, de, None, None, None, None, None


def attention_conv(q: Tensor, k: Tensor, v: Tensor, e: Tensor, csc: "ops.CSC", num_heads: int, reverse=None, dropout_p: float = 0.0,
                   dropout_seed: int = 0) -> Tensor:
    return AttentionConvFunction.apply(q, k, v, e, csc, num_heads, reverse, dropout_p, dropout_seed)


class FusedAttentionFunction(torch.autograd.Function):
    """out = attention(q, k, v, e) + self_term, with q / k / v / self_term given as COLUMN SLABS of the fused projection buffers
    they were computed into (processor: one [N, 4A] buffer; mapper: [N_dst, 2A] and [N_src, 2A]).  The backward kernels write
    dq / dk / dv straight into the matching slabs of ONE gradient buffer per projection, so autograd sees a single gradient
    per GEMM output instead of four zero-filled slice gradients and their sums.

    ``spec``: {"A": width, "q": (buffer index, first column), "k": ..., "v": ..., "s": ...}."""

    @staticmethod
    def forward(ctx, spec, csc, num_heads, reverse, e, *bufs):
        A = spec["A"]
        slab = lambda key: bufs[spec[key][0]][:, spec[key][1]: spec[key][1] + A]  # noqa: E731
        out, lse = ops.gt_attention(slab("q"), slab("k"), slab("v"), e, csc, num_heads, return_lse=True)
        ctx.spec, ctx.csc, ctx.H, ctx.reverse = spec, csc, num_heads, reverse
        ctx.save_for_backward(e, out, lse, *bufs)
        return out + slab("s")

    @staticmethod
    def backward(ctx, d_y):
        e, out, lse, *bufs = ctx.saved_tensors
        spec, A = ctx.spec, ctx.spec["A"]
        d_y = d_y.contiguous()
        covered = [0] * len(bufs)
        for key in ("q", "k", "v", "s"):
            covered[spec[key][0]] += A
        grads = [torch.empty_like(b, memory_format=torch.contiguous_format) if covered[i] == b.shape[1] else torch.zeros_like(b)
                 for i, b in enumerate(bufs)]
        slab = lambda ts, key: ts[spec[key][0]][:, spec[key][1]: spec[key][1] + A]  # noqa: E731
        _, _, _, de = ops.gt_attention_backward(d_y, slab(bufs, "q"), slab(bufs, "k"), slab(bufs, "v"), e, out, lse, ctx.csc, ctx.reverse,
                                                ctx.H, grads_out=(slab(grads, "q"), slab(grads, "k"), slab(grads, "v")))
        slab(grads, "s").copy_(d_y)
        return (None, None, None, None, de, *grads)


def fused_attention(spec: dict, bufs, e: Tensor, csc: "ops.CSC", num_heads: int, reverse) -> Tensor:
    return FusedAttentionFunction.apply(spec, csc, num_heads, reverse, e, *bufs)


class PackEdgeFeaturesFunction(torch.autograd.Function):
    """[M, Fe] edge attributes -> fp32 [M, fe_pad] = [attributes | 1 | 0] (the operand of the fused-edge attention); shared by
    the layers of a processor, so the attribute gradient is accumulated in this layout and sliced once."""

    @staticmethod
    def forward(ctx, edge_attr: Tensor):
        ctx.fe, ctx.dtype = edge_attr.shape[1], edge_attr.dtype
        return ops.pack_edge_features(edge_attr)

    @staticmethod
    def backward(ctx, d_feat: Tensor):
        return d_feat[:, :ctx.fe].to(ctx.dtype)


class FusedEdgeAttentionFunction(torch.autograd.Function):
    """out = attention(q, k, v, E = lin_edge(edge attributes)) + self_term with lin_edge FUSED in forward and backward (E and dE
    are never materialised) and q / k / v / self_term given as column slabs of the fused projection buffers (see
    ``FusedAttentionFunction``).  Inputs: packed edge features (``PackEdgeFeaturesFunction``), lin_edge weight and bias."""

    @staticmethod
    def forward(ctx, spec, csc, num_heads, reverse, feat, weight, bias, *bufs):
        A = spec["A"]
        slab = lambda key: bufs[spec[key][0]][:, spec[key][1]: spec[key][1] + A]  # noqa: E731
        wp = ops.pack_edge_weights(weight.contiguous(), bias)
        y, lse = ops.gt_attention_fused_edge(slab("q"), slab("k"), slab("v"), feat, wp, csc, num_heads, addend=slab("s"), return_lse=True)
        ctx.spec, ctx.csc, ctx.H, ctx.reverse = spec, csc, num_heads, reverse
        ctx.fe, ctx.wdt, ctx.bdt = weight.shape[1], weight.dtype, None if bias is None else bias.dtype
        ctx.save_for_backward(feat, wp, y, lse, *bufs)  # y includes the self term; the backward kernel subtracts it again
        return y

    @staticmethod
    def backward(ctx, d_y):
        feat, wp, out, lse, *bufs = ctx.saved_tensors
        spec, A = ctx.spec, ctx.spec["A"]
        d_y = d_y.contiguous()
        covered = [0] * len(bufs)
        for key in ("q", "k", "v", "s"):
            covered[spec[key][0]] += A
        grads = [torch.empty_like(b, memory_format=torch.contiguous_format) if covered[i] == b.shape[1] else torch.zeros_like(b)
                 for i, b in enumerate(bufs)]
        slab = lambda ts, key: ts[spec[key][0]][:, spec[key][1]: spec[key][1] + A]  # noqa: E731
        _, _, _, d_wp, d_feat = ops.gt_attention_fused_edge_backward(
            d_y, slab(bufs, "q"), slab(bufs, "k"), slab(bufs, "v"), feat, wp, out, lse, ctx.csc, ctx.reverse, ctx.H,
            grads_out=(slab(grads, "q"), slab(grads, "k"), slab(grads, "v")), need_feat_grad=ctx.needs_input_grad[4],
            addend=slab(bufs, "s"), d_addend=slab(grads, "s"))
        d_w = d_b = None
        if ctx.needs_input_grad[5] or ctx.needs_input_grad[6]:
            d_wb = d_wp[:, :ctx.fe + 1].to(ctx.wdt)  # one cast for weight and bias
            d_w = d_wb[:, :ctx.fe] if ctx.needs_input_grad[5] else None
            d_b = d_wb[:, ctx.fe].to(ctx.bdt) if ctx.bdt is not None and ctx.needs_input_grad[6] else None
        return (None, None, None, None, d_feat, d_w, d_b, *grads)


def pack_edge_features(edge_attr: Tensor) -> Tensor:
    return PackEdgeFeaturesFunction.apply(edge_attr)


def fused_edge_attention(spec: dict, bufs, feat: Tensor, lin_edge, csc: "ops.CSC", num_heads: int, reverse) -> Tensor:
    return FusedEdgeAttentionFunction.apply(spec, csc, num_heads, reverse, feat, lin_edge.weight, lin_edge.bias, *bufs)
