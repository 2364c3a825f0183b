"""Message-passing blocks — mirror of reference layers/block.py (GraphTransformer{Mapper,Processor}Block :482-1273,
GraphConv{Processor,Mapper}Block :293-479): same constructor keywords, forward signatures and state_dict keys, so a
checkpoint of the reference loads unchanged.  The forward pass is a short chain of fused HIP kernels:

  GT processor block (7 launches):  LN -> [q|k|v|self] GEMM -> edge attention (lin_edge fused, + self term)
                                    -> projection GEMM (+ skip) -> LN -> MLP-1 GEMM (+GELU) -> MLP-2 GEMM (+ skip)

instead of the reference's ~25 eager ops with [M, H, C] temporaries.
"""
from __future__ import annotations

import os
from typing import Optional, Union

import torch
from torch import Tensor, nn

from .. import ops
from ..distributed import primitives as comm
from ..distributed.halo import HaloInfo, build_halo_info
from ..distributed.partition import build_graph_partition_from_shard_info
from ..distributed.shapes import BipartiteGraphShardInfo, GraphShardInfo, comm_rank, comm_size, model_is_distributed
from .conv import DeferredAggregate, GraphConv, mlp_chain_ok, node_mlp_chain
from . import conv as _conv
from .graphcache import get_csc, get_edge_features, get_reverse_csr
from .kernels import PaddedLinear
from .normalization import ConditionalLayerNorm, apply_layer_norm
from .mlp import MLP
from .utils import compute_mlp_hidden_dim
from ..utils.tensors import version

ANEMOI_DEBUG_SHARDING = os.environ.get("ANEMOI_DEBUG_SHARDING", "") != ""
# LayerNorm folded into the neighbouring GEMMs (anemoi_linear_stats_fwd / anemoi_linear_lnfold_fwd): 36 of the 39 LayerNorm
# launches of the O96 forward disappear.  On by default since both sides run on the fast interior-tile epilogue (-3 % forward
# time in same-box A/Bs; it was neutral with the generic epilogue: DESIGN.md section 5).  ANEMOI_LN_FOLD=0: LayerNorm + GEMM.
_FUSED_EDGE_BWD = os.environ.get("ANEMOI_FUSED_EDGE_BWD", "1") == "1"  # 0: train through the materialised-E op (reference op boundary)
_LN_FOLD = os.environ.get("ANEMOI_LN_FOLD", "1") == "1"
# Below ~4k rows the fold's consumer runs in the 64 x 128 / 192 x 128 kernels (statistics from the producer's strip sums, read through
# L1).  Per forward, LayerNorm + GEMM / folded: 642 hidden nodes 1.75 / 1.68 ms, 2 562 nodes 1.94 / 1.91 ms, one rank's 1 281 rows of an
# 8-way sharded mesh 1.50 / 1.50 ms, 10 242 nodes 3.09 / 3.00 ms.  (Until late round 3 the small-tile consumer lost 60 us per launch to
# ONE lane walking a tail row load by load, which had made the fold look slower below 4 096 rows.)  Not below 512 rows: not measured.
_LN_FOLD_MIN_ROWS = int(os.environ.get("ANEMOI_LN_FOLD_MIN_ROWS", "512"))
# The row-local tail of a block (projection + skip, LayerNorm, MLP + skip, the NEXT block's LayerNorm + q|k|v|self projection) as ONE
# launch that keeps a 48-row panel in LDS and streams the weights (ops.gt_layer_chain2, csrc/gt_chain2.hip: the workgroup's waves in two
# groups working different GEMM segments, the LayerNorms' affine parts folded into the weights here, once per parameter version, each CU
# warming the L2 with its share of the next step's weights) instead of four GEMM launches with the LayerNorm fold between them.
# Where it wins (measured): from ~4 200 rows on - one panel per CU (the hidden mesh of res 5: O96 forward 2.98 -> 2.57 ms), four
# rounds of panels (res 6; the O96 decoder's 40 320 rows), the N320 decoder's 45 rounds.  Below, a launch costs the same ~75-90 us (every
# busy CU streams all of the layer's weights for its 48 rows, and few CUs are busy) while the GEMM launches shrink with the rows: 2 562
# hidden nodes 1.91 ms per forward on the launches against 2.30 ms on the chain - and a rank's share of a sharded mesh is that small
# (2 ranks: 5 121 + 277 rows 2.39 -> 2.16 ms on the chain; 4 ranks: 2 561 + 260 rows 1.59 -> 1.84 ms; profiles/r05_rank_floor_gate.txt).
# ANEMOI_LAYER_CHAIN=1: every eligible block; =0: never; ANEMOI_LAYER_CHAIN_MIN_ROWS: the gate (4 096 rows).
_chain_env = os.environ.get("ANEMOI_LAYER_CHAIN", "")
_LAYER_CHAIN = _chain_env != "0"
_LAYER_CHAIN_MIN_ROWS = int(os.environ.get("ANEMOI_LAYER_CHAIN_MIN_ROWS", "0" if _chain_env == "1" else "4096"))
# Below that gate (round 6): the CLUSTER chain (ops.gt_cluster_chain, csrc/gt_cluster_chain.hip) - four CUs of one XCD share a 48-row panel
# as a tensor-parallel group over the MLP's hidden width, so that a block tail of a few thousand rows (a rank's share of a sharded mesh,
# the hidden meshes of res 3 / 4) puts four times as many CUs to work, each streaming 2 MiB of weights instead of 6.5.  ANEMOI_CLUSTER_CHAIN=0:
# the LayerNorm-fold GEMM launches below the gate, as in round 5.
_CLUSTER_CHAIN = os.environ.get("ANEMOI_CLUSTER_CHAIN", "1") != "0"
# ... and on a SHARDED mesh that launch also computes the next block's projections of the local rows, so that the halo exchange moves the owners'
# k | v rows and the LayerNorm launch and the q|k|v|self GEMM over local + halo rows disappear.  ANEMOI_CLUSTER_HALO=0: LayerNorm'd rows on the wire.
_CLUSTER_HALO = os.environ.get("ANEMOI_CLUSTER_HALO", "1") != "0"


_IDENTITY: dict = {}


def _identity_index(x: Tensor) -> Tensor:
    """int32 [0 .. rows) on x's device (cached per device and length)."""
    key = (str(x.device), x.shape[0])
    t = _IDENTITY.get(key)
    if t is None:
        if len(_IDENTITY) > 16:
            _IDENTITY.clear()
        t = _IDENTITY[key] = torch.arange(x.shape[0], dtype=torch.int32, device=x.device)
    return t


class _FusedWeights:
    """Concatenated projection weights ([Wq;Wk;Wv;Ws] ...) rebuilt only when a parameter changes."""

    def __init__(self):
        self._cache = {}

    def get(self, tag: str, linears: list) -> tuple[Tensor, Tensor]:
        ps = [p for lin in linears for p in (lin.weight, lin.bias) if p is not None]
        sig = tuple((p.data_ptr(), version(p), p.dtype, str(p.device)) for p in ps)
        if torch.is_grad_enabled() and any(p.requires_grad for p in ps):  # training: gradients flow through the cat
            return (torch.cat([lin.weight for lin in linears], dim=0),
                    torch.cat([lin.bias if lin.bias is not None else lin.weight.new_zeros(lin.out_features) for lin in linears]))
        hit = self._cache.get(tag)
        if hit is not None and hit[0] == sig:
            return hit[1], hit[2]
        with torch.no_grad():
            w = torch.cat([lin.weight for lin in linears], dim=0).contiguous()
            b = torch.cat([lin.bias if lin.bias is not None else lin.weight.new_zeros(lin.out_features) for lin in linears]).contiguous()
        self._cache[tag] = (sig, w, b)
        return w, b

    def ln_folded(self, tag: str, linears: list, ln) -> tuple[Tensor, Tensor, Tensor]:
        """(W diag(gamma) in the model dtype, c = its fp32 row sums, d = W beta + b in fp32) for a LayerNorm folded into the
        GEMM that follows it (ops.linear_ln_folded); rebuilt only when a parameter changes."""
        ps = [p for lin in linears for p in (lin.weight, lin.bias) if p is not None] + [p for p in (ln.weight, ln.bias) if p is not None]
        sig = tuple((p.data_ptr(), version(p), p.dtype, str(p.device)) for p in ps)
        hit = self._cache.get("ln:" + tag)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            w = torch.cat([lin.weight for lin in linears], dim=0).float()
            b = torch.cat([lin.bias.float() if lin.bias is not None else w.new_zeros(lin.out_features) for lin in linears])
            ws = (w * ln.weight.float()).to(linears[0].weight.dtype).contiguous()
            c = ws.float().sum(1).contiguous()
            d = (b if ln.bias is None else w @ ln.bias.float() + b).contiguous()
        self._cache["ln:" + tag] = (sig, (ws, c, d))
        return ws, c, d

    def derived(self, tag: str, params: list, builder):
        """``builder()`` cached until one of ``params`` changes (fragment-major images of weight slices / stacks)."""
        ps = [p for p in params if p is not None]
        sig = tuple((p.data_ptr(), version(p), p.dtype, str(p.device)) for p in ps)
        hit = self._cache.get("der:" + tag)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            val = builder()
        self._cache["der:" + tag] = (sig, val)
        return val

    def packed_edge(self, lin_edge) -> Tensor:
        """fp32 [D, fe_pad] image of lin_edge for the fused attention, rebuilt only when the parameters change."""
        ps = [p for p in (lin_edge.weight, lin_edge.bias) if p is not None]
        sig = tuple((p.data_ptr(), version(p), p.dtype, str(p.device)) for p in ps)
        hit = self._cache.get("edge")
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            w = ops.pack_edge_weights(lin_edge.weight.contiguous(), lin_edge.bias)
        self._cache["edge"] = (sig, w)
        return w


class BaseBlock(nn.Module):
    """Base class for network blocks."""


# ============================================================================================ GraphTransformer blocks
class GraphTransformerBaseBlock(BaseBlock):
    def __init__(self, *, in_channels: int, hidden_dim: int, out_channels: int, num_heads: int, edge_dim: int,
                 bias: bool = True, qk_norm: bool = False, mlp_implementation: str = "mlp", update_src_nodes: bool = False,
                 layer_kernels, attn_channels: Optional[int] = None, graph_attention_backend: str = "hip",
                 edge_pre_mlp: bool = False, **kwargs) -> None:
        super().__init__()
        self.update_src_nodes = update_src_nodes
        self.attn_channels = out_channels if attn_channels is None else attn_channels
        if self.attn_channels <= 0:
            raise ValueError(f"attn_channels must be > 0, got {self.attn_channels}")
        if self.attn_channels % num_heads != 0:
            raise ValueError(f"attn_channels ({self.attn_channels}) must be divisible by num_heads ({num_heads}) in {self.__class__.__name__}.")
        self.out_channels_conv = self.attn_channels // num_heads
        self.num_heads = num_heads
        self.qk_norm = qk_norm

        Linear, LayerNorm = layer_kernels.Linear, layer_kernels.LayerNorm
        A = num_heads * self.out_channels_conv
        self.lin_key = Linear(in_channels, A)
        self.lin_query = Linear(in_channels, A)
        self.lin_value = Linear(in_channels, A)
        self.lin_self = Linear(in_channels, A, bias=bias)
        self.lin_edge = Linear(edge_dim, A)
        self.projection = Linear(self.attn_channels, out_channels)
        if self.qk_norm:
            self.q_norm = layer_kernels.QueryNorm(self.out_channels_conv)
            self.k_norm = layer_kernels.KeyNorm(self.out_channels_conv)
        self.layer_norm_attention = LayerNorm(normalized_shape=in_channels)
        self.layer_norm_mlp_dst = LayerNorm(normalized_shape=out_channels)
        self.node_dst_mlp = MLP(in_features=out_channels, hidden_dim=hidden_dim, out_features=out_channels,
                                layer_kernels=layer_kernels, n_extra_layers=0, layer_norm=False,
                                mlp_implementation=mlp_implementation)
        if edge_pre_mlp:  # build_feedforward_layer -> Sequential(Linear, GELU): keys edge_pre_mlp.0.* (mlp.py:55-95)
            self.edge_pre_mlp = nn.Sequential(Linear(edge_dim, edge_dim), layer_kernels.Activation())
        else:
            self.edge_pre_mlp = nn.Identity()
        # the reference's "triton" / "pyg" both map to the one HIP implementation
        if graph_attention_backend not in ("hip", "triton", "pyg"):
            raise ValueError(f"Backend '{graph_attention_backend}' not supported for GraphTransformerBlock")
        self.graph_attention_backend = "hip"
        self._fused = _FusedWeights()
        self._pad_edge = PaddedLinear()

    # -- pieces ---------------------------------------------------------------------------------------------
    def _attention(self, query: Tensor, key: Tensor, value: Tensor, x_r: Tensor, edge_attr: Tensor, csc: ops.CSC,
                   fused: Optional[dict] = None, edge_prep: Optional[dict] = None) -> Tensor:
        """(attention output + x_r) [N_dst, A]; q/k/v may be column slices of a fused projection buffer.  ``fused`` names
        those buffers and columns ({"bufs": (...), "q": (buffer, column), "k", "v", "s"}) so that the training path can hand
        autograd one gradient per buffer; ``edge_prep``: per-forward cache of the prepared (cast, zero-padded) edge attributes
        shared by the layers of a processor."""
        H, C = self.num_heads, self.out_channels_conv
        if self.qk_norm:  # per-head LayerNorm over C, no bias (block.py:655-660)
            query = self.q_norm(query.reshape(-1, H, C)).view(-1, H * C)
            key = self.k_norm(key.reshape(-1, H, C)).view(-1, H * C)
        if ops._needs_grad(query, key, value, x_r, edge_attr, self.lin_edge.weight):
            # training (scope row f1): E = lin_edge(edge_pre_mlp(edge_attr)) is materialised, as in the reference
            # (block.py:623-635), and the attention runs through the op mirror, whose backward is registered
            from .. import autograd as ag
            from ..autograd import attention, fused_attention

            wdt = self.lin_edge.weight.dtype
            if fused is not None and _FUSED_EDGE_BWD and ops.fused_edge_backward_supported(query.shape[1], H, edge_attr.shape[1]):
                # lin_edge fused into the attention in forward AND backward: E / dE never exist
                if isinstance(self.edge_pre_mlp, nn.Identity):
                    fkey = (id(edge_attr), id(csc.perm), "feat")  # the packed features are shared by the layers of a processor
                    feat = None if edge_prep is None else edge_prep.get(fkey)
                    if feat is None:
                        feat = ag.pack_edge_features(edge_attr if csc.perm is None else edge_attr.index_select(0, csc.perm))
                        if edge_prep is not None:
                            edge_prep[fkey] = feat
                            edge_prep.setdefault("anchors", []).append((edge_attr, csc.perm))
                else:
                    # edge_pre_mlp (block.py:585-586) is this block's own Linear + GELU on the [M, edge_dim] attributes: it
                    # runs as such (autograd through the GEMM path), its output is packed for the fused kernel, and the
                    # kernel's feature gradient flows back into it
                    ea = edge_attr if csc.perm is None else edge_attr.index_select(0, csc.perm)
                    feat = ag.pack_edge_features(self._pad_edge(ea.to(wdt), self.edge_pre_mlp[0], act="gelu"))
                bufs = tuple(fused["bufs"])
                spec = {"A": query.shape[1], **{kk: fused[kk] for kk in ("q", "k", "v", "s")}}
                if self.qk_norm:
                    # q_norm / k_norm made NEW tensors of q and k: they enter as two more (whole) buffers; v and the self term
                    # stay column slabs of the projection buffers, whose q / k columns get their gradient through the norms
                    bufs = (*bufs, query.contiguous(), key.contiguous())
                    spec["q"], spec["k"] = (len(bufs) - 2, 0), (len(bufs) - 1, 0)
                return ag.fused_edge_attention(spec, bufs, feat, self.lin_edge, csc, H, get_reverse_csr(csc))
            pkey = (id(edge_attr), id(csc.perm), wdt)
            ea = None if edge_prep is None else edge_prep.get(pkey)
            if ea is None:
                ea = edge_attr if csc.perm is None else edge_attr.index_select(0, csc.perm)
                ea = ea.to(wdt)
                # edge_dim (11) is no multiple of 8: zero-pad K so that forward, dX and dW all run on the MFMA kernels
                if wdt != torch.float32 and ea.shape[1] % 8:
                    ea = torch.nn.functional.pad(ea, (0, (-ea.shape[1]) % 8))
                if edge_prep is not None:
                    edge_prep[pkey] = ea
                    edge_prep.setdefault("anchors", []).append((edge_attr, csc.perm))  # keep the ids alive for this forward
            if not isinstance(self.edge_pre_mlp, nn.Identity):
                ea = self._pad_edge(ea, self.edge_pre_mlp[0], act="gelu")
            e = self._pad_edge(ea, self.lin_edge)
            if fused is not None and not self.qk_norm:
                spec = {"A": query.shape[1], **{kk: fused[kk] for kk in ("q", "k", "v", "s")}}
                return fused_attention(spec, fused["bufs"], e, csc, H, get_reverse_csr(csc))
            return attention(query, key, value, e, csc, H, get_reverse_csr(csc)) + x_r
        if isinstance(self.edge_pre_mlp, nn.Identity):
            feat = get_edge_features(edge_attr, csc.perm)
        else:
            lin = self.edge_pre_mlp[0]
            ea = edge_attr if csc.perm is None else edge_attr.index_select(0, csc.perm)
            ea = ops.linear(ea.to(lin.weight.dtype), lin.weight, lin.bias, act="gelu")
            feat = ops.pack_edge_features(ea)
        return ops.gt_attention_fused_edge(query, key, value, feat, self._fused.packed_edge(self.lin_edge), csc, H, addend=x_r)

    # -- heads ("Ulysses") strategy: reference block.py:689-759, 838-854 ----------------------------------------------
    def _heads_full_graph(self, edge_attr, edge_index, edge_sizes, group, train: bool, cache: Optional[dict]):
        """(edge_index, edge_attr) of the WHOLE graph on every rank.  Edge slices are dst-owned and contiguous, so rank
        order = global dst-sorted order.  The index part is cached; static attributes are gathered once, differentiable
        ones per step with their gradients summed over the ranks (every rank works on every edge, for its heads)."""
        P, rank = comm_size(group), comm_rank(group)
        if edge_sizes is None:
            return edge_index, edge_attr
        key = (edge_index.data_ptr(), version(edge_index), P, rank)
        full = None if cache is None else cache.get("heads_full")
        if full is None or full[0] != key:
            full = (key, comm.gather_tensor(edge_index.t().contiguous(), 0, edge_sizes, group).t().contiguous(), (edge_index,))
            if cache is not None:
                cache["heads_full"] = full
        if train:
            return full[1], comm.gather_tensor(edge_attr.contiguous(), 0, edge_sizes, group, reduce_in_backward=True)
        akey = (edge_attr.data_ptr(), version(edge_attr), key)
        hit = None if cache is None else cache.get("heads_attr")
        if hit is None or hit[0] != akey:
            hit = (akey, comm.gather_tensor(edge_attr.contiguous(), 0, edge_sizes, group), edge_attr)
            if cache is not None:
                cache["heads_attr"] = hit
        return full[1], hit[1]

    def _heads_attention(self, q_loc: Tensor, k_loc: Tensor, v_loc: Tensor, ea_full: Tensor, ei_full: Tensor, src_sizes, dst_sizes,
                         group, train: bool) -> Tensor:
        """Nodes sharded for the dense work, HEADS sharded for the attention: q (local destination rows) and k, v (local
        source rows) are transposed by one all-to-all each into all rows x (H / P) heads, the attention runs on the whole
        graph for this rank's heads, and a fourth all-to-all brings the local destination rows back with all heads
        ([n_dst_local, A]).  Unlike the reference the edge term is not moved at all (there: a fourth [M, H*C] transpose
        per layer): every rank applies its rows of ``lin_edge`` to the (gathered) edge attributes."""
        P, rank = comm_size(group), comm_rank(group)
        H, C, A = self.num_heads, self.out_channels_conv, self.attn_channels
        if H % P:
            raise ValueError(f"shard_strategy='heads': num_heads ({H}) must be divisible by the model-parallel size ({P})")
        Hl = H // P
        src_sizes, dst_sizes = list(src_sizes), list(dst_sizes)
        csc = get_csc(ei_full, (sum(src_sizes), sum(dst_sizes)), True)

        def to_heads(t, sizes):  # [n_loc, A] all heads -> [n_full, Hl*C] my heads
            n_loc = t.shape[0]
            send = t.reshape(n_loc, P, Hl * C).permute(1, 0, 2).reshape(P * n_loc, Hl * C)
            return comm.all_to_all_rows(send, [n_loc] * P, sizes, group)

        q, k, v = to_heads(q_loc, dst_sizes), to_heads(k_loc, src_sizes), to_heads(v_loc, src_sizes)
        if self.qk_norm:  # per-head LayerNorm over C: applied to this rank's heads (block.py:748)
            q = self.q_norm(q.reshape(-1, Hl, C)).view(-1, Hl * C)
            k = self.k_norm(k.reshape(-1, Hl, C)).view(-1, Hl * C)
        rows = slice(rank * Hl * C, (rank + 1) * Hl * C)  # this rank's heads of lin_edge
        if train:  # materialised E for this rank's heads, attention through the differentiable op (scope rows f1 x f2)
            from ..autograd import attention

            lin = self.lin_edge
            ea = ea_full if csc.perm is None else ea_full.index_select(0, csc.perm)
            ea = ea.to(lin.weight.dtype)
            if not isinstance(self.edge_pre_mlp, nn.Identity):
                # edge_pre_mlp (block.py:585-586) acts on the [M, edge_dim] attributes before any head exists: every rank applies
                # it to the gathered attributes (replicated, M x edge_dim^2 flops); its parameter gradients are this rank's heads'
                # share and are completed by reduce_parameter_gradients like all others
                ea = self._pad_edge(ea, self.edge_pre_mlp[0], act="gelu")
            w_e = lin.weight[rows]
            pad = (-ea.shape[1]) % 8
            if pad and ea.dtype != torch.float32:  # 16-bit operand rows must be 16-byte aligned
                ea, w_e = torch.nn.functional.pad(ea, (0, pad)), torch.nn.functional.pad(w_e, (0, pad))
            e = ops.linear(ea, w_e, None if lin.bias is None else lin.bias[rows])
            o = attention(q, k, v, e, csc, Hl, get_reverse_csr(csc))
        else:
            if isinstance(self.edge_pre_mlp, nn.Identity):
                feat = get_edge_features(ea_full, csc.perm)
            else:
                pre = self.edge_pre_mlp[0]
                ea = ea_full if csc.perm is None else ea_full.index_select(0, csc.perm)
                feat = ops.pack_edge_features(ops.linear(ea.to(pre.weight.dtype), pre.weight, pre.bias, act="gelu"))
            o = ops.gt_attention_fused_edge(q, k, v, feat, self._fused.packed_edge(self.lin_edge)[rows], csc, Hl)  # [n_dst_full, Hl*C]
        n_loc = q_loc.shape[0]
        back = comm.all_to_all_rows(o, dst_sizes, [n_loc] * P, group)  # [P*n_loc, Hl*C]: block r = heads of rank r
        return back.reshape(P, n_loc, Hl * C).permute(1, 0, 2).reshape(n_loc, A)

    def _ln_fold_ok(self, ln, x: Tensor) -> bool:
        """The LayerNorm-fold path: inference, 16-bit, plain affine LayerNorm (see include/anemoi_hip.h)."""
        return (_LN_FOLD and x.shape[0] >= _LN_FOLD_MIN_ROWS and type(ln).__name__ in ("LayerNorm", "AutocastLayerNorm")
                and x.dtype != torch.float32 and x.is_cuda
                and not (torch.is_grad_enabled() and (x.requires_grad or ln.weight.requires_grad)))

    def _chain_ok(self, ln, x: Tensor, cluster: bool = False) -> bool:
        """The row-resident chain kernel takes this block's projection / LayerNorm / MLP: inference, 16-bit, 512 channels, plain
        affine LayerNorm, Linear-GELU-Linear MLP with a hidden width that is a multiple of 512.  ``cluster``: the same question for the
        cluster chain, which takes the row counts BELOW the chain's gate (and a hidden width of 2048)."""
        mlp = self.node_dst_mlp
        if cluster and not (_CLUSTER_CHAIN and 0 < x.shape[0] < _LAYER_CHAIN_MIN_ROWS and mlp.mlp_implementation == "mlp" and len(mlp.mlp) == 3
                            and mlp.mlp[0].weight.shape[0] == 4 * ops.CHAIN_CHANNELS):
            return False
        return (_LAYER_CHAIN and x.is_cuda and x.dtype != torch.float32
                and (cluster or x.shape[0] >= _LAYER_CHAIN_MIN_ROWS)
                and type(ln).__name__ in ("LayerNorm", "AutocastLayerNorm") and ln.weight is not None
                and mlp.mlp_implementation == "mlp" and len(mlp.mlp) == 3 and mlp.layer_norm is None
                and self.projection.weight.shape == (ops.CHAIN_CHANNELS, ops.CHAIN_CHANNELS) and self.projection.bias is not None
                and mlp.mlp[0].weight.shape[1] == ops.CHAIN_CHANNELS and mlp.mlp[0].weight.shape[0] % ops.CHAIN_CHANNELS == 0
                and mlp.mlp[2].weight.shape[0] == ops.CHAIN_CHANNELS and mlp.mlp[0].bias is not None and mlp.mlp[2].bias is not None
                # (mixed layouts - an fp32-kept LayerNorm or weights next to 16-bit activations - take the GEMM launches; the chain ops raise on them)
                and all(p is None or p.dtype == x.dtype for p in (ln.weight, ln.bias, self.projection.weight, self.projection.bias,
                                                                   mlp.mlp[0].weight, mlp.mlp[0].bias, mlp.mlp[2].weight, mlp.mlp[2].bias))
                # (the chain ops build no autograd graph: ANY trainable parameter of the tail keeps the differentiable path)
                and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for m in (ln, self.projection, mlp) for p in m.parameters()))))

    def _qkvs_chain_ok(self, x: Tensor) -> bool:
        """This block's layer_norm_attention + fused [q|k|v|self] projection can ride at the end of the PREVIOUS block's chain launch."""
        ln = self.layer_norm_attention
        A = self.attn_channels
        return (type(ln).__name__ in ("LayerNorm", "AutocastLayerNorm") and ln.weight is not None and not self.qk_norm
                and x.shape[1] == ops.CHAIN_CHANNELS and (4 * A) % ops.CHAIN_CHANNELS == 0 and self.lin_query.weight.shape[1] == ops.CHAIN_CHANNELS
                and all(p is None or p.dtype == x.dtype for m in (ln, self.lin_query, self.lin_key, self.lin_value, self.lin_self) for p in m.parameters())
                and not (torch.is_grad_enabled() and any(p.requires_grad for m in (ln, self.lin_query, self.lin_key, self.lin_value, self.lin_self)
                                                         for p in m.parameters())))

    def _kv_chain_ok(self, x: Tensor) -> bool:
        """(mapper blocks) layer_norm_attention_src + fused [k|v] projection of the SOURCE rows can ride at the end of the chain launch that produces
        those rows - the last processor block's, behind the latent skip."""
        ln = getattr(self, "layer_norm_attention_src", None)
        A = self.attn_channels
        mods = (ln, self.lin_key, self.lin_value)
        return (ln is not None and type(ln).__name__ in ("LayerNorm", "AutocastLayerNorm") and ln.weight is not None and not self.qk_norm
                and x.shape[1] == ops.CHAIN_CHANNELS and (2 * A) % ops.CHAIN_CHANNELS == 0 and self.lin_key.weight.shape[1] == ops.CHAIN_CHANNELS
                and all(p is None or p.dtype == x.dtype for m in mods for p in m.parameters())
                and not (torch.is_grad_enabled() and any(p.requires_grad for m in mods for p in m.parameters())))

    def _post_attention(self, attn_plus_self: Tensor, x_skip: Tensor, cond: Optional[Tensor] = None, chain: Optional[dict] = None,
                        extra: Optional[Tensor] = None, next_block=None, halo=None) -> Tensor:
        """projection + residual, LayerNorm, MLP + residual.  Inference: the projection GEMM also emits the row statistics of
        its output and the MLP's first GEMM applies the LayerNorm from them (no LayerNorm launch); with ``chain`` the last
        GEMM does the same for the NEXT block's first LayerNorm.  ``extra`` (last block of a processor): the model's latent
        skip (encoder_processor_decoder.py:295-296), added by the last GEMM's epilogue as a second residual."""
        # ``halo`` (a sharded processor block at inference: (HaloPlan, group)): where the cluster chain takes the tail it also computes the NEXT
        # block's LayerNorm + projections of the LOCAL rows - q | self into a buffer of their own, k | v straight into the head of the
        # [local + halo, 2A] buffer whose tail the halo exchange fills.  The next block then starts with that exchange (``chain["halo_pre"]``):
        # what crosses the wire are the owners' k | v rows (2 KiB per row instead of the 1-KiB LayerNorm'd row the reference exchanges,
        # layers/block.py:1159-1172 - the exchange is latency-bound) and no rank projects a halo row again.
        if extra is not None and (ops._needs_grad(attn_plus_self, x_skip, extra, self.projection.weight) or extra.shape != x_skip.shape):
            return self._post_attention(attn_plus_self, x_skip, cond, chain) + extra
        ln, mlp = self.layer_norm_mlp_dst, self.node_dst_mlp
        plain_mlp = mlp.mlp_implementation == "mlp" and len(mlp.mlp) == 3 and mlp.layer_norm is None
        cluster = cond is None and self._chain_ok(ln, attn_plus_self, cluster=True)
        if (cond is None and (cluster or self._chain_ok(ln, attn_plus_self)) and attn_plus_self.shape == x_skip.shape
                and (extra is None or extra.shape == x_skip.shape)):
            # the NEXT consumer of this tail's output whose LayerNorm + projection ride at the end of the launch: the next processor block
            # (q|k|v|self), or - behind the LAST processor block and the latent skip - the decoder's block (k|v of its source rows)
            to_mapper = isinstance(next_block, GraphTransformerMapperBlock)
            nb = None
            if next_block is not None and chain is not None:
                if to_mapper:
                    nb = next_block if (not cluster and halo is None and next_block._kv_chain_ok(x_skip)) else None
                elif extra is None and next_block._qkvs_chain_ok(x_skip):
                    nb = next_block
            if halo is not None and not (_CLUSTER_HALO and cluster and nb is not None and getattr(nb, "shard_strategy", None) == "edges"):
                nb = None  # (a sharded block's k | v need the halo rows: only the cluster chain has the LayerNorm'd rows as an output)
            hidden = mlp.mlp[0].weight.shape[0]
            # a decoder's node_data_extractor (LayerNorm + Linear(512, out), layers/mapper.py:688-704) as the chain launch's NARROW trailing
            # projection: the 40 320-row output of the tail is then neither written nor read back, and two launches disappear
            tail = None
            if chain is not None and nb is None and extra is None and not cluster and chain.get("tail_proj") is not None:
                ln_t, lin_t = chain["tail_proj"]
                o_pad = (lin_t.out_features + 127) // 128 * 128
                if (o_pad < ops.CHAIN_CHANNELS and lin_t.in_features == ops.CHAIN_CHANNELS and type(ln_t).__name__ in ("LayerNorm", "AutocastLayerNorm")
                        and ln_t.weight is not None and ops.gt_layer_chain2_supported(attn_plus_self, hidden, o_pad)
                        and all(q_ is None or q_.dtype == attn_plus_self.dtype for m_ in (ln_t, lin_t) for q_ in m_.parameters())
                        and not (torch.is_grad_enabled() and any(q_.requires_grad for m_ in (ln_t, lin_t) for q_ in m_.parameters()))):
                    tail = (ln_t, lin_t, o_pad)
            supported = ops.gt_cluster_chain_supported if cluster else ops.gt_layer_chain2_supported
            if nb is not None and not supported(attn_plus_self, hidden, 4 * nb.attn_channels):
                nb = None  # the per-column vectors of tail + trailing projection do not fit the kernel's LDS region: the tail alone
            if supported(attn_plus_self, hidden, 0 if nb is None else 4 * nb.attn_channels):
                lin1, lin2 = mlp.mlp[0], mlp.mlp[2]
                qlins = [] if nb is None else ([nb.lin_key, nb.lin_value] if to_mapper else
                                              [nb.lin_query, nb.lin_self, nb.lin_key, nb.lin_value] if halo is not None else
                                              [nb.lin_query, nb.lin_key, nb.lin_value, nb.lin_self])
                q_out = sum(lin.out_features for lin in qlins)
                lnq = None if nb is None else (nb.layer_norm_attention_src if to_mapper else nb.layer_norm_attention)
                params = [self.projection.weight, self.projection.bias, ln.weight, ln.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias]
                params += [q for lin in qlins for q in (lin.weight, lin.bias)] + ([] if lnq is None else [lnq.weight, lnq.bias])
                if tail is not None:
                    params += [tail[0].weight, tail[0].bias, tail[1].weight, tail[1].bias]

                def build():
                    w1g, d1 = ops.fold_layer_norm(lin1.weight, lin1.bias, ln.weight, ln.bias)
                    parts = [self.projection.bias.float(), d1, lin2.bias.float()]
                    wqg = None
                    if nb is not None:
                        wq = torch.cat([lin.weight for lin in qlins], dim=0)
                        bq = torch.cat([lin.bias if lin.bias is not None else lin.weight.new_zeros(lin.out_features) for lin in qlins])
                        wq, dq = ops.fold_layer_norm(wq, bq, lnq.weight, lnq.bias)
                        wqg = ops.pack_weight_frag(wq)
                        parts.append(dq)
                    elif tail is not None:
                        ln_t, lin_t, o_pad = tail
                        pad = o_pad - lin_t.out_features
                        wt = torch.nn.functional.pad(lin_t.weight, (0, 0, 0, pad))  # zero rows: the padded output columns are zeros
                        bt = torch.nn.functional.pad(lin_t.bias if lin_t.bias is not None else lin_t.weight.new_zeros(lin_t.out_features), (0, pad))
                        wt, dq = ops.fold_layer_norm(wt, bt, ln_t.weight, ln_t.bias)
                        wqg = ops.pack_weight_frag(wt)
                        parts.append(dq)
                    return (ops.pack_weight_frag(self.projection.weight), ops.pack_weight_frag(w1g), ops.pack_weight_frag(lin2.weight),
                            torch.cat(parts).to(lin1.weight.dtype).contiguous(), wqg)

                wp, w1g, w2, vec, wqg = self._fused.derived(("chain2:tail" if tail is not None else "chain2") if nb is None else f"chain2:{id(nb)}:{len(qlins)}",
                                                            params, build)
                if tail is not None:
                    _, q_t = ops.gt_layer_chain2(attn_plus_self, x_skip, wp, w1g, w2, vec, hidden, ln.eps, wqg=wqg, q_out_features=tail[2],
                                                 lnq_eps=tail[0].eps, want_x_out=False)
                    chain["tail_out"] = q_t[:, :tail[1].out_features]  # (a strided [N, out] view, as the mapper's own GEMM returns)
                    return q_t
                kw = {}
                if halo is not None and nb is not None:
                    plan, group = halo
                    nl = x_skip.shape[0]
                    buf = comm.recv_buffer(nl, plan.send_counts, plan.recv_counts, 2 * nb.attn_channels, x_skip.dtype, x_skip.device, group)
                    kw["q_out2"], kw["q_split"] = buf[:nl], 2 * nb.attn_channels // ops.CHAIN_CHANNELS
                res = (ops.gt_cluster_chain if cluster else ops.gt_layer_chain2)(
                    attn_plus_self, x_skip, wp, w1g, w2, vec, hidden, ln.eps, extra=extra, wqg=wqg,
                    q_out_features=q_out, lnq_eps=1e-5 if lnq is None else lnq.eps, **kw)
                if kw:
                    chain["halo_pre"] = {"x": res[0], "buf": buf, "qs": res[1]}
                    return res[0]
                if nb is not None:
                    chain["qkvs_x"], chain["qkvs"] = res
                    return res[0]
                return res
            # (a hidden width whose vectors do not fit the kernel's LDS region: the GEMM launches below)
        if plain_mlp and self._ln_fold_ok(ln, attn_plus_self):
            r = ops.linear_with_row_stats(attn_plus_self, self.projection.weight, self.projection.bias, x_skip)
            if r is not None:
                out, stats = r
                lin1, lin2 = mlp.mlp[0], mlp.mlp[2]
                ws, c, d = self._fused.ln_folded("mlp1", [lin1], ln)
                h = ops.linear_ln_folded(out, ws, c, d, stats, ln.eps, act="gelu")
                if h is None:
                    h = ops.linear(ops.layer_norm(out, ln.weight, ln.bias, ln.eps), lin1.weight, lin1.bias, act="gelu")
                if extra is not None:  # second residual = the gather-add epilogue with the identity index
                    return ops.linear(h, lin2.weight, lin2.bias, residual=out, g1=extra, idx1=_identity_index(extra))
                r2 = ops.linear_with_row_stats(h, lin2.weight, lin2.bias, out) if chain is not None else None
                if r2 is not None:
                    chain["x"], chain["stats"] = r2
                    return r2[0]
                return ops.linear(h, lin2.weight, lin2.bias, residual=out)
        out = ops.linear(attn_plus_self, self.projection.weight, self.projection.bias, residual=x_skip)
        h = apply_layer_norm(ln, out, cond)
        y = self.node_dst_mlp(h, residual=out)
        return y if extra is None else y + extra


class GraphTransformerMapperBlock(GraphTransformerBaseBlock):
    """Bipartite block: LN_src / LN_dst, q & self from dst, k & v from src, MLP on dst (block.py:870-1029)."""

    def __init__(self, *, in_channels: int, hidden_dim: int, out_channels: int, num_heads: int, edge_dim: int,
                 bias: bool = True, qk_norm: bool = False, mlp_implementation: str = "mlp", update_src_nodes: bool = False,
                 layer_kernels, shard_strategy: str = "edges", graph_attention_backend: str = "hip",
                 edge_pre_mlp: bool = False, **kwargs) -> None:
        super().__init__(in_channels=in_channels, hidden_dim=hidden_dim, out_channels=out_channels, edge_dim=edge_dim,
                         layer_kernels=layer_kernels, num_heads=num_heads, bias=bias, qk_norm=qk_norm,
                         mlp_implementation=mlp_implementation, update_src_nodes=update_src_nodes,
                         graph_attention_backend=graph_attention_backend, edge_pre_mlp=edge_pre_mlp, **kwargs)
        LayerNorm = layer_kernels.LayerNorm
        self.layer_norm_attention_src = LayerNorm(normalized_shape=in_channels)
        self.layer_norm_attention_dest = self.layer_norm_attention  # alias: both keys appear in the state_dict
        if self.update_src_nodes:
            self.layer_norm_mlp_src = LayerNorm(normalized_shape=out_channels)
            self.node_src_mlp = MLP(in_features=out_channels, hidden_dim=hidden_dim, out_features=out_channels,
                                    layer_kernels=layer_kernels, n_extra_layers=0, layer_norm=False,
                                    mlp_implementation=mlp_implementation)
        else:
            self.layer_norm_mlp_src = nn.Identity()
            self.node_src_mlp = nn.Identity()
        if shard_strategy not in ("edges", "heads"):
            raise ValueError(f"Invalid shard strategy '{shard_strategy}'")
        self.shard_strategy = shard_strategy

    def forward(self, x, edge_attr: Tensor, edge_index: Tensor, shard_info: BipartiteGraphShardInfo, batch_size: int,
                size, model_comm_group=None, cond=None, edges_are_dst_sorted: bool = True, **layer_kwargs):
        x_src, x_dst = x
        size = (size, size) if isinstance(size, int) else tuple(size)
        heads = self.shard_strategy == "heads" and model_is_distributed(model_comm_group)
        if heads:  # x_src / x_dst are this rank's rows, edge_index / edge_attr the whole graph (see the mapper)
            if batch_size != 1:
                raise ValueError("shard_strategy='heads' requires batch_size=1 when model sharding is enabled.")
        else:
            csc = get_csc(edge_index, size, edges_are_dst_sorted)
        A = self.attn_channels
        ln_s, ln_d = self.layer_norm_attention_src, self.layer_norm_attention_dest
        cond_src, cond_dst = cond if cond is not None else (None, None)  # block.py:979-980
        # LayerNorm + projection per side; with the row statistics left by the mapper's embedding (``ln_stats``, inference) the
        # LayerNorm is applied inside the projection GEMM from raw rows
        st = layer_kwargs.get("ln_stats") or {}
        # (the mapper's row chain launches computed a side's projection together with its embedding: layers/mapper.py)
        qs, kv = st.get("proj:dst"), st.get("proj:src")
        if cond is not None:
            qs = kv = None
        if cond is None:
            e = st.get("dst")
            if qs is None and e is not None and e[0] is x_dst and self._ln_fold_ok(ln_d, x_dst):
                ws, c, d = self._fused.ln_folded("qs", [self.lin_query, self.lin_self], ln_d)
                qs = ops.linear_ln_folded(x_dst, ws, c, d, e[1], ln_d.eps)
            e = st.get("src")
            if kv is None and e is not None and e[0] is x_src and self._ln_fold_ok(ln_s, x_src):
                ws, c, d = self._fused.ln_folded("kv", [self.lin_key, self.lin_value], ln_s)
                kv = ops.linear_ln_folded(x_src, ws, c, d, e[1], ln_s.eps)
        if qs is None:
            w_qs, b_qs = self._fused.get("qs", [self.lin_query, self.lin_self])
            qs = ops.linear(apply_layer_norm(ln_d, x_dst, cond_dst), w_qs, b_qs)
        if kv is None:
            w_kv, b_kv = self._fused.get("kv", [self.lin_key, self.lin_value])
            kv = ops.linear(apply_layer_norm(ln_s, x_src, cond_src), w_kv, b_kv)
        if heads:
            train = ops._needs_grad(qs, kv, edge_attr, self.lin_edge.weight)
            out = self._heads_attention(qs[:, :A], kv[:, :A], kv[:, A:], edge_attr, edge_index, shard_info.src_nodes,
                                        shard_info.dst_nodes, model_comm_group, train) + qs[:, A:]
        else:
            out = self._attention(qs[:, :A], kv[:, :A], kv[:, A:], qs[:, A:], edge_attr, csc,
                                  fused=dict(bufs=(qs, kv), q=(0, 0), s=(0, A), k=(1, 0), v=(1, A)))
        ch = layer_kwargs.get("ln_chain")  # chain: statistics (or the ready projections) for the processor's first block
        nodes_new_dst = self._post_attention(out, x_dst, cond_dst, ch, next_block=None if ch is None else ch.pop("next_block", None))
        if self.update_src_nodes:
            ln = self.layer_norm_mlp_src
            nodes_new_src = self.node_src_mlp(apply_layer_norm(ln, x_src, cond_src), residual=x_src)
        else:
            nodes_new_src = x_src
        return (nodes_new_src, nodes_new_dst), edge_attr


class GraphTransformerProcessorBlock(GraphTransformerBaseBlock):
    """Hidden-mesh block (block.py:1032-1273).  With a model-parallel group the LayerNorm'd rows of cut edges are
    exchanged once per layer (halo), k/v are computed on local+halo rows and attention runs on the relabelled graph."""

    def __init__(self, *, in_channels: int, hidden_dim: int, out_channels: int, num_heads: int, edge_dim: int,
                 bias: bool = True, qk_norm: bool = False, mlp_implementation: str = "mlp", update_src_nodes: bool = False,
                 layer_kernels, shard_strategy: str = "edges", graph_attention_backend: str = "hip",
                 edge_pre_mlp: bool = False, **kwargs) -> None:
        super().__init__(in_channels=in_channels, hidden_dim=hidden_dim, out_channels=out_channels, edge_dim=edge_dim,
                         layer_kernels=layer_kernels, num_heads=num_heads, bias=bias, qk_norm=qk_norm,
                         mlp_implementation=mlp_implementation, update_src_nodes=update_src_nodes,
                         graph_attention_backend=graph_attention_backend, edge_pre_mlp=edge_pre_mlp, **kwargs)
        if shard_strategy not in ("edges", "heads"):
            raise ValueError(f"Invalid shard strategy '{shard_strategy}'")
        self.shard_strategy = shard_strategy
        self._cached_halo = None  # (specs, HaloPlan)

    def _halo_plan(self, x: Tensor, edge_index: Tensor, shard_info: GraphShardInfo, batch_size: int, group, shared: Optional[dict]):
        if batch_size != 1:
            raise ValueError("GraphTransformerProcessorBlock halo exchange requires batch_size=1 when model sharding is enabled.")
        specs = (comm_size(group), comm_rank(group), tuple(shard_info.nodes or ()), tuple(shard_info.edges or ()),
                 edge_index.data_ptr(), version(edge_index))
        if shared is not None and shared.get("specs") == specs:
            return shared["plan"]
        if self._cached_halo is not None and self._cached_halo[0] == specs:
            return self._cached_halo[1]
        assert shard_info.edges_are_sharded(), "Halo strategy requires edges to be sharded"
        bip = BipartiteGraphShardInfo(src_nodes=shard_info.nodes, dst_nodes=shard_info.nodes, edges=shard_info.edges)
        partition = build_graph_partition_from_shard_info(edge_index, (x, x), bip, group)
        plan = HaloPlan(build_halo_info(partition, edge_index, comm_rank(group), edges_are_local=True, debug=ANEMOI_DEBUG_SHARDING))
        self._cached_halo = (specs, plan)
        if shared is not None:
            shared["specs"], shared["plan"] = specs, plan
        return plan

    def forward(self, x: Tensor, edge_attr: Tensor, edge_index: Tensor, shard_info: GraphShardInfo, batch_size: int,
                size, model_comm_group=None, cond=None, edges_are_dst_sorted: bool = True, halo_cache: Optional[dict] = None,
                **kwargs):
        A = self.attn_channels
        ln = self.layer_norm_attention
        chain = kwargs.get("ln_chain")
        nxt = None if chain is None else chain.pop("next_block", None)
        if (not model_is_distributed(model_comm_group) and chain is not None and cond is None
                and (self._ln_fold_ok(ln, x) or chain.get("qkvs_x") is x or self._chain_ok(self.layer_norm_mlp_dst, x))):
            qkvs = None
            if chain.get("qkvs_x") is x:  # the previous block's chain launch computed this block's projections already
                qkvs = chain["qkvs"]
            elif chain.get("x") is x and self._ln_fold_ok(ln, x):  # the previous block's last GEMM left the row statistics of x
                ws, c, d = self._fused.ln_folded("qkvs", [self.lin_query, self.lin_key, self.lin_value, self.lin_self], ln)
                qkvs = ops.linear_ln_folded(x, ws, c, d, chain["stats"], ln.eps)
            if qkvs is None:
                w, b = self._fused.get("qkvs", [self.lin_query, self.lin_key, self.lin_value, self.lin_self])
                qkvs = ops.linear(ops.layer_norm(x, ln.weight, ln.bias, ln.eps), w, b)
            chain.clear()
            q, k, v, x_r = qkvs[:, :A], qkvs[:, A:2 * A], qkvs[:, 2 * A:3 * A], qkvs[:, 3 * A:]
            csc = get_csc(edge_index, (x.shape[0], x.shape[0]), edges_are_dst_sorted)
            out = self._attention(q, k, v, x_r, edge_attr, csc)
            return self._post_attention(out, x, cond, chain, kwargs.get("extra_residual"), next_block=nxt), edge_attr
        sharded = model_is_distributed(model_comm_group) and self.shard_strategy != "heads"
        x_plus_halo = None
        pre = chain.pop("halo_pre", None) if chain is not None else None
        if (pre is not None and pre["x"] is x and sharded and cond is None and not ops._needs_grad(x, ln.weight, self.lin_key.weight)):
            # the previous block's cluster-chain launch left this block's q | self projection and, in the head of the [local + halo, 2A]
            # buffer, the k | v rows of its own nodes: the exchange brings the halo nodes' k | v rows from their owners, then attention, tail
            plan = self._halo_plan(x, edge_index, shard_info, batch_size, model_comm_group, halo_cache)
            nl = x.shape[0]
            kv, qs = pre["buf"], pre["qs"]
            comm.halo_exchange_into(kv, nl, plan.send_index, plan.send_counts, plan.recv_counts, model_comm_group, ops.gather_rows)
            csc = get_csc(plan.edge_index_local, (plan.info.total_nodes, plan.info.num_local_nodes), True)
            out = self._attention(qs[:, :A], kv[:, :A], kv[:, A:], qs[:, A:], edge_attr, csc,
                                  fused=dict(bufs=(qs, kv), q=(0, 0), s=(0, A), k=(1, 0), v=(1, A)), edge_prep=kwargs.get("edge_prep"))
            return self._post_attention(out, x, cond, chain, kwargs.get("extra_residual"), next_block=nxt, halo=(plan, model_comm_group)), edge_attr
        if (sharded and cond is None and not isinstance(ln, ConditionalLayerNorm) and not ops._needs_grad(x, ln.weight)):
            # inference on a shard: LayerNorm writes the head of the [local + halo] buffer, the all-to-all receives into its tail
            plan = self._halo_plan(x, edge_index, shard_info, batch_size, model_comm_group, halo_cache)
            nl = x.shape[0]
            x_plus_halo = comm.recv_buffer(nl, plan.send_counts, plan.recv_counts, x.shape[1], x.dtype, x.device, model_comm_group)
            xn = ops.layer_norm(x, ln.weight, ln.bias, ln.eps, out=x_plus_halo[:nl])
            comm.halo_exchange_into(x_plus_halo, nl, plan.send_index, plan.send_counts, plan.recv_counts, model_comm_group, ops.gather_rows)
        else:
            xn = apply_layer_norm(ln, x, cond)
        if model_is_distributed(model_comm_group) and self.shard_strategy == "heads":
            return self._forward_heads(x, xn, edge_attr, edge_index, shard_info, batch_size, model_comm_group, cond, halo_cache,
                                       extra=kwargs.get("extra_residual")), edge_attr
        if model_is_distributed(model_comm_group):
            plan = self._halo_plan(x, edge_index, shard_info, batch_size, model_comm_group, halo_cache)
            if x_plus_halo is None:
                x_plus_halo = comm.halo_exchange(xn, plan.send_index, plan.send_counts, plan.recv_counts, model_comm_group,
                                                 gather_fn=ops.gather_rows)
            nl = xn.shape[0]
            if not ops._needs_grad(xn, self.lin_query.weight) and x_plus_halo.shape[0] <= 2 * nl:
                # inference on a shard: ONE fused [q|k|v|self] GEMM over local + halo rows (q / self of the halo rows are
                # wasted work, but at a rank's row counts a GEMM launch costs more than those flops)
                w, b = self._fused.get("qkvs", [self.lin_query, self.lin_key, self.lin_value, self.lin_self])
                qkvs = ops.linear(x_plus_halo, w, b)
                q, k, v, x_r = qkvs[:nl, :A], qkvs[:, A:2 * A], qkvs[:, 2 * A:3 * A], qkvs[:nl, 3 * A:]
                fused = None
            else:
                w_qs, b_qs = self._fused.get("qs", [self.lin_query, self.lin_self])
                w_kv, b_kv = self._fused.get("kv", [self.lin_key, self.lin_value])
                qs = ops.linear(xn, w_qs, b_qs)
                kv = ops.linear(x_plus_halo, w_kv, b_kv)
                q, k, v, x_r = qs[:, :A], kv[:, :A], kv[:, A:], qs[:, A:]
                fused = dict(bufs=(qs, kv), q=(0, 0), s=(0, A), k=(1, 0), v=(1, A))
            csc = get_csc(plan.edge_index_local, (plan.info.total_nodes, plan.info.num_local_nodes), True)
        else:
            n = x.shape[0]
            w, b = self._fused.get("qkvs", [self.lin_query, self.lin_key, self.lin_value, self.lin_self])
            qkvs = ops.linear(xn, w, b)
            q, k, v, x_r = qkvs[:, :A], qkvs[:, A:2 * A], qkvs[:, 2 * A:3 * A], qkvs[:, 3 * A:]
            fused = dict(bufs=(qkvs,), q=(0, 0), k=(0, A), v=(0, 2 * A), s=(0, 3 * A))
            csc = get_csc(edge_index, (n, n), edges_are_dst_sorted)
        out = self._attention(q, k, v, x_r, edge_attr, csc, fused=fused, edge_prep=kwargs.get("edge_prep"))
        if x_plus_halo is not None and sharded and chain is not None:  # inference on a shard: the tail may prepare the next block's exchange
            return self._post_attention(out, x, cond, chain, kwargs.get("extra_residual"), next_block=nxt, halo=(plan, model_comm_group)), edge_attr
        return self._post_attention(out, x, cond, extra=kwargs.get("extra_residual")), edge_attr


    def _forward_heads(self, x, xn, edge_attr, edge_index, shard_info, batch_size, group, cond, cache: Optional[dict], extra=None):
        """shard_strategy="heads" (see ``_heads_attention``): the fused q/k/v/self projection runs on the local rows."""
        if batch_size != 1:
            raise ValueError("shard_strategy='heads' requires batch_size=1 when model sharding is enabled.")
        A = self.attn_channels
        train = ops._needs_grad(xn, edge_attr, self.lin_edge.weight)
        sizes = list(shard_info.nodes)
        ei_full, ea_full = self._heads_full_graph(edge_attr, edge_index, shard_info.edges if shard_info.edges_are_sharded() else None,
                                                  group, train, cache)
        w, b = self._fused.get("qkvs", [self.lin_query, self.lin_key, self.lin_value, self.lin_self])
        qkvs = ops.linear(xn, w, b)
        out = self._heads_attention(qkvs[:, :A], qkvs[:, A:2 * A], qkvs[:, 2 * A:3 * A], ea_full, ei_full, sizes, sizes, group, train)
        return self._post_attention(out + qkvs[:, 3 * A:], x, cond, extra=extra)


class HaloPlan:
    """HaloInfo + the device-side buffers the exchange needs (built once, reused by every layer and step)."""

    def __init__(self, info: HaloInfo):
        self.info = info
        dev = info.edge_index_local.device
        idx = torch.cat(list(info.send_indices)) if len(info.send_indices) else torch.zeros(0, dtype=torch.long, device=dev)
        self.send_index = idx.to(torch.int32).contiguous()
        self.send_counts = list(info.send_counts)
        self.recv_counts = list(info.recv_counts)
        self.edge_index_local = info.edge_index_local


# ============================================================================================ GraphConv (GNN) blocks
class GraphConvBaseBlock(BaseBlock):
    def __init__(self, *, in_channels: int, out_channels: int, num_chunks: int, mlp_extra_layers: int = 0,
                 mlp_hidden_ratio: float = 1.0, mlp_implementation: str = "mlp", update_src_nodes: bool = True,
                 layer_kernels, edge_dim: Optional[int] = None, **kwargs) -> None:
        super().__init__()
        hidden_dim = compute_mlp_hidden_dim(out_channels, mlp_hidden_ratio)
        if edge_dim:
            self.emb_edges = MLP(in_features=edge_dim, hidden_dim=hidden_dim, out_features=out_channels,
                                 layer_kernels=layer_kernels, n_extra_layers=mlp_extra_layers + 1,
                                 mlp_implementation=mlp_implementation)
        else:
            self.emb_edges = None
        self.update_src_nodes = update_src_nodes
        self.num_chunks = num_chunks  # memory chunking is unnecessary with 288 GB of HBM; kept for API compatibility
        self.node_mlp = MLP(in_features=2 * in_channels, hidden_dim=hidden_dim, out_features=out_channels,
                            layer_kernels=layer_kernels, n_extra_layers=mlp_extra_layers + 1,
                            mlp_implementation=mlp_implementation)
        self.conv = GraphConv(in_channels=in_channels, out_channels=out_channels, layer_kernels=layer_kernels,
                              mlp_extra_layers=mlp_extra_layers, mlp_implementation=mlp_implementation)


class GraphConvProcessorBlock(GraphConvBaseBlock):
    def forward(self, x: Tensor, edge_attr: Tensor, edge_index: Tensor, shard_info: GraphShardInfo, model_comm_group=None,
                size=None, **layer_kwargs):
        if self.emb_edges is not None:
            edge_attr = self.emb_edges(edge_attr)
        chain = layer_kwargs.get("gnn_chain")
        nxt = None if chain is None else chain.pop("next_block", None)
        if model_is_distributed(model_comm_group):  # block.py:375: all node rows are needed as sources
            x_in = comm.gather_tensor(x, 0, shard_info.nodes, model_comm_group, reduce_in_backward=True)
            n_loc = x.shape[0]
            d0 = sum(shard_info.nodes[: comm_rank(model_comm_group)])
            out_full, edges_new = self._conv_sharded(x_in, x, d0, edge_attr, edge_index, layer_kwargs.get("local_edge_cache"))
            out = out_full
            assert out.shape[0] == n_loc
        else:
            # inference on the chain kernels: the previous block's node chain may have left this block's stacked node-level terms
            p = chain["p"] if (chain is not None and chain.get("p_x") is x and self.conv.chain_ok(x, edge_attr)) else None
            if chain is not None:
                chain.clear()
            out, edges_new = self.conv(x, edge_attr, edge_index, size=size, p=p, defer_sum=mlp_chain_ok(self.node_mlp, 2 * ops.CHAIN_CHANNELS, x))
        if mlp_chain_ok(self.node_mlp, 2 * ops.CHAIN_CHANNELS, x) and out.dtype == x.dtype:
            kw = {}
            if (chain is not None and not model_is_distributed(model_comm_group) and isinstance(nxt, GraphConvProcessorBlock)
                    and nxt.emb_edges is None and nxt.conv.chain_ok(x, edges_new)):
                kw = dict(wt=nxt.conv.stacked_frag(), t_out_features=2 * ops.CHAIN_CHANNELS)
            res = node_mlp_chain(self.node_mlp, x, out, **kw)
            if kw:
                chain["p_x"], chain["p"] = res
                return res[0], edges_new
            return res, edges_new
        if isinstance(out, DeferredAggregate):
            out = out.materialize()
        nodes_new = self.node_mlp(x, x2=out, residual=x)
        return nodes_new, edges_new

    def _conv_sharded(self, x_all: Tensor, x_loc: Tensor, d0: int, edge_attr: Tensor, edge_index: Tensor, cache: Optional[dict] = None):
        # local edges carry GLOBAL ids: sources index the gathered table, destinations are shifted to local rows.  The
        # relabelled index is shared by all layers of a processor (``cache``): one CSC for the whole stack.
        cache = self.__dict__.setdefault("_own_cache", {}) if cache is None else cache
        key = (edge_index.data_ptr(), version(edge_index), d0)
        if cache.get("key") != key:
            cache["key"], cache["ei"], cache["anchor"] = key, torch.stack([edge_index[0], edge_index[1] - d0]), edge_index
        return self.conv((x_all, x_loc), edge_attr, cache["ei"], size=(x_all.shape[0], x_loc.shape[0]))


class GraphConvMapperBlock(GraphConvBaseBlock):
    def forward(self, x, edge_attr: Tensor, edge_index: Tensor, shard_info: BipartiteGraphShardInfo, model_comm_group=None,
                size=None, **layer_kwargs):
        x_src, x_dst = x
        if model_is_distributed(model_comm_group):
            # block.py:441-479: node shards in, this rank's (dst-owned, globally numbered) edges; every source row is made
            # available (the mappers of this package call forward_local with only the rows they need instead)
            x_src_all = comm.gather_tensor(x_src, 0, shard_info.src_nodes, model_comm_group, reduce_in_backward=True)
            d0 = sum(shard_info.dst_nodes[: comm_rank(model_comm_group)])
            ei = torch.stack([edge_index[0], edge_index[1] - d0])
            return self.forward_local(x_src_all, x_dst, x_src, edge_attr, ei)
        return self.forward_local(x_src, x_dst, x_src, edge_attr, edge_index, size=size)

    def forward_local(self, x_src_conv: Tensor, x_dst: Tensor, x_src_update: Tensor, edge_attr: Tensor, edge_index: Tensor, size=None):
        """``x_src_conv``: the source rows ``edge_index[0]`` refers to; ``x_src_update``: the source rows this rank owns
        (the same tensor when nothing is sharded)."""
        size = (x_src_conv.shape[0], x_dst.shape[0]) if size is None else size
        out, edges_new = self.conv((x_src_conv, x_dst), edge_attr, edge_index, size=size,
                                   defer_sum=x_dst.dim() == 2 and mlp_chain_ok(self.node_mlp, 2 * ops.CHAIN_CHANNELS, x_dst))

        def node(xr, x2):  # LayerNorm(node_mlp([x | x2])) + x: one row-resident launch where the chain kernel takes the shapes
            if mlp_chain_ok(self.node_mlp, 2 * ops.CHAIN_CHANNELS, xr) and x2.dtype == xr.dtype and xr.dim() == 2:
                return node_mlp_chain(self.node_mlp, xr, x2)
            if isinstance(x2, DeferredAggregate):
                x2 = x2.materialize()
            return self.node_mlp(xr, x2=x2, residual=xr)

        nodes_new_dst = node(x_dst, out)
        nodes_new_src = node(x_src_update, x_src_update) if self.update_src_nodes else x_src_update
        return (nodes_new_src, nodes_new_dst), edges_new
