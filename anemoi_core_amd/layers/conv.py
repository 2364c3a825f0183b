"""Graph operators — mirror of reference layers/conv.py (GraphTransformerConv :84-147, GraphConv :29-81) on the
HIP kernels.  No torch_geometric: the message passing is a CSC walk inside one kernel."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

import os

from .. import ops
from .graphcache import get_csc, get_reverse_csr
from .mlp import MLP
from ..utils.tensors import version

# GraphConv's edge MLP + LayerNorm + residual as ONE row-resident launch (ops.gnn_edge_chain, csrc/gnn_chain.hip) and the node MLPs of the
# GraphConv blocks likewise (gnn_node_chain): O96 GNN forward 7.63 -> 6.29 ms.  ANEMOI_GNN_CHAIN=0: the launch-per-GEMM path (same-box A/Bs).
_GNN_CHAIN = os.environ.get("ANEMOI_GNN_CHAIN", "1") == "1"


def _derived(owner, tag: str, params: list, builder):
    """``builder()`` cached on ``owner`` until one of ``params`` changes (fragment-major images of weights and weight slices)."""
    cache = owner.__dict__.setdefault("_derived_cache", {})
    sig = tuple((p.data_ptr(), version(p), p.dtype, str(p.device)) for p in params)
    hit = cache.get(tag)
    if hit is not None and hit[0] == sig:
        return hit[1]
    with torch.no_grad():
        val = builder()
    cache[tag] = (sig, val)
    return val


# GraphConv's scatter-sum inside the node chain launch that consumes it (ops.gnn_node_chain(seg_ptr=...)); 0: its own launch
_NODE_SEGSUM = os.environ.get("ANEMOI_GNN_NODE_SEGSUM", "1") == "1"


class DeferredAggregate:
    """``scatter(edges_new, dst, "sum")`` not yet taken: the dst-sorted edge rows and the CSC column pointer, handed to the node chain
    kernel, which sums a panel's in-edge rows while it loads the panel (same arithmetic as ops.segment_sum_rows)."""
    __slots__ = ("rows", "ptr")

    def __init__(self, rows: Tensor, ptr: Tensor):
        self.rows, self.ptr = rows, ptr

    @property
    def dtype(self):
        return self.rows.dtype

    def materialize(self) -> Tensor:
        return ops.segment_sum_rows(self.rows, self.ptr)


def mlp_chain_ok(m: MLP, k_in: int, x: Tensor) -> bool:
    """A GraphConv-style MLP the row-resident chain kernels take: inference, 16-bit, 512 channels, Linear-GELU-Linear-GELU-Linear with
    a plain affine LayerNorm (mlp_extra_layers = 0, mlp_implementation = "mlp")."""
    D = ops.CHAIN_CHANNELS
    return (_GNN_CHAIN and x.is_cuda and x.dtype != torch.float32 and x.shape[-1] == D
            and m.mlp_implementation == "mlp" and len(m.mlp) == 5 and m.layer_norm is not None
            and type(m.layer_norm).__name__ in ("LayerNorm", "AutocastLayerNorm") and m.layer_norm.weight is not None
            and m.mlp[0].weight.shape == (D, k_in) and m.mlp[2].weight.shape == (D, D) and m.mlp[4].weight.shape == (D, D)
            and m.mlp[0].weight.dtype == x.dtype and all(m.mlp[i].bias is not None for i in (0, 2, 4))
            # (the chain ops build no autograd graph: ANY trainable parameter of the MLP keeps the differentiable path)
            and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in m.parameters()))))


def node_mlp_chain(m: MLP, x: Tensor, agg: Tensor, *, wt: Optional[Tensor] = None, t_out_features: int = 0):
    """``m(x, x2=agg, residual=x)`` of a GraphConv block (LayerNorm(MLP([x | agg])) + x) as one launch; optionally the trailing projection
    ``x_out wt^T`` (returns ``(x_out, t_out)``)."""
    P = ops.pack_weight_frag
    wa = _derived(m, "na", [m.mlp[0].weight], lambda: P(m.mlp[0].weight))
    wb = _derived(m, "nb", [m.mlp[2].weight], lambda: P(m.mlp[2].weight))
    wc = _derived(m, "nc", [m.mlp[4].weight], lambda: P(m.mlp[4].weight))
    ln = m.layer_norm
    kw = dict(wt=wt, t_out_features=t_out_features) if wt is not None else {}
    if isinstance(agg, DeferredAggregate):
        agg, kw["seg_ptr"] = agg.rows, agg.ptr
    return ops.gnn_node_chain(x, agg, wa, m.mlp[0].bias, wb, m.mlp[2].bias, wc, m.mlp[4].bias, ln.weight, ln.bias, ln.eps, **kw)


class GraphTransformerConv(nn.Module):
    """forward(query[N_dst,H,C], key[N_src,H,C], value[N_src,H,C], edge_attr[M,H,C]|None, edge_index[2,M], size)
    -> [N_dst,H,C]; softmax over the in-edges of every destination, edge features added to keys and values."""

    def __init__(self, out_channels: int, dropout: float = 0.0, **kwargs):
        super().__init__()
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        self.out_channels = out_channels
        self.dropout = dropout  # conv.py:145: dropout on the attention weights, active in training mode only

    def forward(self, query: Tensor, key: Tensor, value: Tensor, edge_attr: Optional[Tensor], edge_index: Tensor,
                size=None, edges_are_dst_sorted: bool = False) -> Tensor:
        n_dst, H, C = query.shape
        size = (key.shape[0], n_dst) if size is None else size
        csc = get_csc(edge_index, size, edges_are_dst_sorted)
        e = None
        if edge_attr is not None:
            e = edge_attr if csc.perm is None else edge_attr.index_select(0, csc.perm)
            e = e.reshape(e.shape[0], H * C)
        flat = lambda t: t.reshape(t.shape[0], H * C)  # noqa: E731
        drop = self.dropout if self.training else 0.0  # conv.py:145: F.dropout(alpha, p, training)
        if drop > 0.0 or ops._needs_grad(query, key, value, edge_attr):
            from .. import autograd

            if e is None:  # the backward kernels walk a materialised edge tensor
                e = query.new_zeros((csc.num_edges, H * C))
            # the seed comes from torch's (CPU) default generator: torch.manual_seed reproduces the mask, no device sync
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) if drop > 0.0 else 0
            return autograd.attention_conv(flat(query), flat(key), flat(value), e, csc, H, get_reverse_csr(csc), drop, seed).view(n_dst, H, C)
        return ops.gt_attention(flat(query), flat(key), flat(value), e, csc, H).view(n_dst, H, C)


class GraphConv(nn.Module):
    """e' = edge_mlp(cat[x_i, x_j, e]) + e ;  out[d] = sum_{e -> d} e'   (reference conv.py:66-81).

    MI355X formulation: cat[x_i, x_j, e] W_a^T = (x_dst W_a1^T)[dst] + (x_src W_a2^T)[src] + e W_a3^T — two NODE-level
    GEMMs plus one edge-level GEMM with a gather-add epilogue (2*(2N + M)*D^2 flop instead of 2*3*M*D^2; M ~ 8N),
    and the trailing LayerNorm + residual + scatter-sum is one segmented pass over the dst-sorted edges."""

    def __init__(self, in_channels: int, out_channels: int, layer_kernels, mlp_extra_layers: int = 0,
                 mlp_implementation: str = "mlp", **kwargs) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.edge_mlp = MLP(3 * in_channels, out_channels, out_channels, layer_kernels=layer_kernels,
                            n_extra_layers=mlp_extra_layers + 1, mlp_implementation=mlp_implementation)

    def _stacked_node_weight(self, w: Tensor, D: int) -> Tensor:
        sig = (w.data_ptr(), version(w), w.dtype, str(w.device))
        hit = self.__dict__.get("_stacked")
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, torch.cat([w[:, :D], w[:, D:2 * D]], dim=0).contiguous())
            self.__dict__["_stacked"] = hit
        return hit[1]

    def stacked_frag(self) -> Tensor:
        """fragment-major image of [W_i; W_j] (the node-level halves of the edge MLP's first Linear): what the PREVIOUS block's node
        chain multiplies its output with, so that this block's edge chain finds its gather operands ready."""
        w = self.edge_mlp.mlp[0].weight
        D = self.in_channels
        return _derived(self, "stacked_frag", [w], lambda: ops.pack_weight_frag(torch.cat([w[:, :D], w[:, D:2 * D]], dim=0)))

    def chain_ok(self, x_dst: Tensor, edge_attr: Tensor) -> bool:
        return (edge_attr.dim() == 2 and edge_attr.dtype == x_dst.dtype and edge_attr.shape[1] == ops.CHAIN_CHANNELS
                and mlp_chain_ok(self.edge_mlp, 3 * ops.CHAIN_CHANNELS, x_dst) and not (torch.is_grad_enabled() and edge_attr.requires_grad))

    def forward(self, x, edge_attr: Tensor, edge_index: Tensor, size=None, edges_are_dst_sorted: bool = True, p: Optional[Tensor] = None,
                defer_sum: bool = False):
        x_src, x_dst = (x, x) if isinstance(x, Tensor) else x
        size = (x_src.shape[0], x_dst.shape[0]) if size is None else size
        csc = get_csc(edge_index, size, edges_are_dst_sorted)
        if csc.perm is not None:
            raise ValueError("GraphConv requires dst-sorted edges (edge features are carried between layers)")
        D = self.in_channels
        lin0 = self.edge_mlp.mlp[0]
        gated = self.edge_mlp.mlp_implementation != "mlp"
        if not gated and self.chain_ok(x_dst, edge_attr) and x_src.dtype == x_dst.dtype and x_src.shape[1] == D:
            # inference: the edge MLP (gather-add form) + LayerNorm + residual as ONE row-resident launch, then the scatter-sum;
            # ``p`` = [x W_i^T | x W_j^T] when the previous block's node chain has computed it already
            em, w = self.edge_mlp, lin0.weight
            if p is not None:
                p_dst, p_src = p[:, :D], p[:, D:]
            elif x_src is x_dst:
                pp = ops.linear(x_dst, self._stacked_node_weight(w, D))
                p_dst, p_src = pp[:, :D], pp[:, D:]
            else:
                p_dst, p_src = ops.linear(x_dst, w[:, :D]), ops.linear(x_src, w[:, D:2 * D])
            P = ops.pack_weight_frag
            w0 = _derived(self, "w0e", [w], lambda: P(w[:, 2 * D:]))
            w1 = _derived(self, "e1", [em.mlp[2].weight], lambda: P(em.mlp[2].weight))
            w2 = _derived(self, "e2", [em.mlp[4].weight], lambda: P(em.mlp[4].weight))
            ln = em.layer_norm
            edges_new = ops.gnn_edge_chain(edge_attr, p_dst, csc.dst, p_src, csc.row, w0, lin0.bias, w1, em.mlp[2].bias, w2, em.mlp[4].bias,
                                           ln.weight, ln.bias, ln.eps)
            if defer_sum and _NODE_SEGSUM and csc.colptr.dtype == torch.int32:  # the caller's node chain takes the scatter-sum with it
                return DeferredAggregate(edges_new, csc.colptr), edges_new
            return ops.segment_sum_rows(edges_new, csc.colptr), edges_new
        if gated:  # first layer = gating(gate_proj(cat)) * value_proj(cat): the same gather-add GEMM on the fused [gate; value] weight
            w, bias0 = lin0.fused_weights()
        else:
            w, bias0 = lin0.weight, lin0.bias  # [out, 3D] = [W_i | W_j | W_e]
        if x_src is x_dst and not gated and not ops._needs_grad(x_src, w):
            # processor (one node set), inference: [x W_i^T | x W_j^T] as ONE node-level GEMM with the stacked weight
            # [W_i; W_j] (rebuilt only when the parameter changes); the edge GEMM gathers from the two column halves
            p = ops.linear(x_dst, self._stacked_node_weight(w, D))
            p_dst, p_src = p[:, :w.shape[0]], p[:, w.shape[0]:]
        else:
            p_dst = ops.linear(x_dst, w[:, :D])
            p_src = ops.linear(x_src, w[:, D:2 * D])
        seg = {}
        if ops._needs_grad(p_dst, p_src, edge_attr, w):  # training: the adjoint of the two row gathers = segment sums
            rowptr, edge_ids, _ = get_reverse_csr(csc)
            seg = dict(seg1=(csc.colptr, None), seg2=(rowptr, edge_ids))
        h = ops.linear(edge_attr, w[:, 2 * D:], bias0, act=None if gated else "gelu", g1=p_dst, idx1=csc.dst, g2=p_src, idx2=csc.row, **seg)
        if gated:
            h = ops.glu(h, lin0.kind)
        z = self.edge_mlp(h, skip_first=True, skip_layer_norm=True)
        ln = self.edge_mlp.layer_norm
        edges_new, out = ops.edge_ln_residual_segment_sum(z, edge_attr, None if ln is None else ln.weight,
                                                          None if ln is None else ln.bias, 1e-5 if ln is None else ln.eps, csc)
        return out, edges_new
