"""TEST-ONLY stand-in for anemoi_core_amd.ops on CPU tensors, built on the oracle.

The product path has no CPU fallback; the multi-rank (gloo) tests below need to drive the HOST logic of the sharded
modules (partitioning, halo plan, needed-rows exchange, caches) without a GPU, so they monkeypatch the kernel entry
points with these torch/oracle restatements inside the spawned test processes only."""
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from anemoi_core_amd import ops as real_ops
from oracle import gt_oracle as O


def _edge_index(csc):
    return torch.stack([csc.row.long(), csc.dst.long()])


def gt_attention(q, k, v, e, csc, num_heads, addend=None, return_lse=False):
    H = num_heads
    C = q.shape[1] // H
    f = lambda t: t.reshape(t.shape[0], H, C).float()  # noqa: E731
    ei = _edge_index(csc)
    ee = f(e) if e is not None else torch.zeros(csc.num_edges, H, C)
    out = O.gt_conv(f(q), f(k), f(v), ee, ei, (csc.n_src, csc.n_dst)).reshape(csc.n_dst, H * C)
    if addend is not None:
        out = out + addend.float()
    out = out.to(q.dtype)
    if return_lse:
        return out, O.gt_conv_lse(f(q), f(k), ee, ei, (csc.n_src, csc.n_dst))
    return out


def pack_edge_features(edge_attr):
    M, fe = edge_attr.shape
    out = torch.zeros(M, real_ops.edge_feature_pad(fe))
    out[:, :fe] = edge_attr.float()
    out[:, fe] = 1.0
    return out


def pack_edge_weights(w_edge, b_edge):
    D, fe = w_edge.shape
    out = torch.zeros(D, real_ops.edge_feature_pad(fe))
    out[:, :fe] = w_edge.float()
    if b_edge is not None:
        out[:, fe] = b_edge.float()
    return out


def gt_attention_fused_edge(q, k, v, edge_feat, w_packed, csc, num_heads, addend=None, return_lse=False):
    e = (edge_feat @ w_packed.t()).to(q.dtype)
    return gt_attention(q, k, v, e, csc, num_heads, addend, return_lse)


def layer_norm(x, weight, bias, eps=1e-5, residual=None, out=None):
    y = F.layer_norm(x.float(), (x.shape[-1],), weight.float(), None if bias is None else bias.float(), eps)
    if residual is not None:
        y = y + residual.float()
    if out is not None:
        out.copy_(y.to(x.dtype))
        return out
    return y.to(x.dtype)


def linear(x, weight, bias=None, *, act=None, residual=None, x2=None, g1=None, idx1=None, g2=None, idx2=None, out=None, seg1=None, seg2=None):
    a = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
    y = F.linear(a, weight.float(), None if bias is None else bias.float())
    if g1 is not None:
        y = y + g1.float()[idx1.long()]
    if g2 is not None:
        y = y + g2.float()[idx2.long()]
    if act == "gelu":
        y = F.gelu(y)
    if residual is not None:
        y = y + residual.float()
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def edge_ln_residual_segment_sum(z, e_old, gamma, beta, eps, csc):
    zz = z.float() if gamma is None else F.layer_norm(z.float(), (z.shape[1],), gamma.float(), None if beta is None else beta.float(), eps)
    e_new = (zz + e_old.float()).to(z.dtype)
    agg = torch.zeros(csc.n_dst, z.shape[1]).index_add_(0, csc.dst.long(), e_new.float()).to(z.dtype)
    return e_new, agg


def gather_rows(x, idx):
    return x.index_select(0, idx.long())


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def attention_train(q, k, v, e, csc, num_heads, reverse):
    """Stand-in for anemoi_core_amd.autograd.attention (the differentiable op mirror): plain torch, so autograd works."""
    return gt_attention(q, k, v, e, csc, num_heads)


def fused_attention_train(spec, bufs, e, csc, num_heads, reverse):
    """Stand-in for anemoi_core_amd.autograd.fused_attention: the same column slabs, plain torch."""
    A = spec["A"]
    slab = lambda key: bufs[spec[key][0]][:, spec[key][1]: spec[key][1] + A]  # noqa: E731
    return gt_attention(slab("q"), slab("k"), slab("v"), e, csc, num_heads) + slab("s")


def pack_edge_features_train(edge_attr):
    """Stand-in for anemoi_core_amd.autograd.pack_edge_features: [attributes | 1 | 0] in plain torch."""
    M, fe = edge_attr.shape
    pad = 4 * ((fe + 1 + 3) // 4) - fe - 1
    return torch.cat([edge_attr.float(), torch.ones(M, 1), torch.zeros(M, pad)], 1)


def fused_edge_attention_train(spec, bufs, feat, lin_edge, csc, num_heads, reverse):
    """Stand-in for anemoi_core_amd.autograd.fused_edge_attention."""
    fe = lin_edge.weight.shape[1]
    e = torch.nn.functional.linear(feat[:, :fe].to(lin_edge.weight.dtype), lin_edge.weight, lin_edge.bias)
    return fused_attention_train(spec, bufs, e, csc, num_heads, reverse)


def install(monkeypatch=None):
    """Patch anemoi_core_amd.ops in the current process (plain setattr when no pytest monkeypatch is given)."""
    names = ["gt_attention", "pack_edge_features", "pack_edge_weights", "gt_attention_fused_edge", "layer_norm", "linear",
             "edge_ln_residual_segment_sum", "gather_rows"]
    from anemoi_core_amd import autograd as _ag

    if monkeypatch is not None:
        monkeypatch.setattr(_ag, "attention", attention_train)
        monkeypatch.setattr(_ag, "fused_attention", fused_attention_train)
        monkeypatch.setattr(_ag, "pack_edge_features", pack_edge_features_train)
        monkeypatch.setattr(_ag, "fused_edge_attention", fused_edge_attention_train)
    else:
        _ag.pack_edge_features = pack_edge_features_train
        _ag.fused_edge_attention = fused_edge_attention_train
        _ag.attention = attention_train
        _ag.fused_attention = fused_attention_train
    for n in names:
        if monkeypatch is not None:
            monkeypatch.setattr(real_ops, n, globals()[n])
        else:
            setattr(real_ops, n, globals()[n])
