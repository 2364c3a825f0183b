"""Mappers (encoder data->hidden, decoder hidden->data) — mirror of reference layers/mapper.py
(GraphTransformerBaseMapper :142-477, Forward :480-597, Backward :600-704, GNN mappers :707-1087): same constructor
keywords, forward signatures and state_dict keys.

Differences that do not change results:
 * the reference's dst-range chunk loop (``num_chunks``, mapper.py:365-381) exists to bound activation memory on
   smaller GPUs; with 288 GB of HBM the whole bipartite graph is processed in one pass (``num_chunks`` is accepted and
   ignored);
 * when sharded, only the source rows this rank's edges touch are fetched (local gather if the source table is
   replicated, a needed-rows all-to-all if it is sharded) instead of all-gathering every source row
   (khop_edges.py:386-392); unconnected sources are never embedded (as in the reference, khop_edges.py:474-500);
 * slicing / ``unique`` / relabelling of the static graph is done once and cached.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor, nn

from .. import ops
from ..distributed import primitives as comm
from ..distributed.partition import (
    build_graph_partition_from_shard_info,
    ensure_edges_are_dst_sorted,
    local_bipartite_graph,
    shard_edges_1hop,
)
from ..distributed.shapes import BipartiteGraphShardInfo, comm_rank, comm_size, get_shard_sizes, model_is_distributed
from .block import GraphConvMapperBlock, GraphTransformerMapperBlock
from .kernels import PaddedLinear
from .mlp import MLP
from .utils import compute_mlp_hidden_dim, load_layer_kernels
from ..utils.tensors import version


# One side of a mapper (node embedding -> the block's LayerNorm -> its fused k|v or q|self projection) as ONE row-resident launch
# (ops.gt_row_chain, csrc/gt_rowchain.hip) instead of the embedding GEMM + the LayerNorm-fold GEMM with the embedded rows written and read
# back in between.  ANEMOI_ROW_CHAIN=0: off; ANEMOI_ROW_CHAIN_MIN_ROWS: the gate (a launch streams ~1.2 MB of weights per 48-row panel: it
# needs enough panels to put most CUs to work).
_ROW_CHAIN = os.environ.get("ANEMOI_ROW_CHAIN", "1") != "0"
_ROW_CHAIN_MIN_ROWS = int(os.environ.get("ANEMOI_ROW_CHAIN_MIN_ROWS", "4096"))
# ... and, measured (tools/rowchain_time.py, profiles/r06_rowchain.txt): the launch re-streams its weights once per round of 48-row panels
# (~17-19 us per round with the two wave groups on different panels, 21 before), the two GEMMs share each weight tile among 160 rows - between
# ~16 000 and ~260 000 rows the GEMM pair is as fast or faster (40 320 rows: 84 against 76-82 us), below (one round: 10 242 rows 23 against 27 us,
# 5 040 rows 20 against 33) and far above (542 080 rows 864 against 1 048 us: there the 555-MB round trip of the embedded rows dominates the GEMMs) the
# launch wins.
# The decoder's node_data_extractor (LayerNorm + Linear(512, out)) as the narrow trailing projection of the block's chain launch (inference, block
# tails on the row-resident chain): ANEMOI_TAIL_PROJ=0 keeps its LayerNorm launch + GEMM.
_TAIL_PROJ = os.environ.get("ANEMOI_TAIL_PROJ", "1") != "0"
_ROW_CHAIN_GEMM_BAND = tuple(int(v) for v in os.environ.get("ANEMOI_ROW_CHAIN_GEMM_BAND", "16384:262144").split(":"))


class _LocalGraphCache:
    """Per-mapper cache of the rank-local bipartite graph (static).  A few entries, not one: the needed-rows plans live with
    their graph and are built COLLECTIVELY - a rank that alternates between the sharded forward and an unsharded one (a
    reference forward on rank 0) must not drop its sharded entry and re-plan alone while its peers wait in the exchange."""

    def __init__(self, keep: int = 4):
        self.entries: dict = {}
        self.keep = keep

    def get(self, key, build):
        hit = self.entries.get(key)
        if hit is None:
            hit = build()
            if len(self.entries) >= self.keep:
                self.entries.pop(next(iter(self.entries)))
            self.entries[key] = hit
        return hit


class BaseMapper(nn.Module):
    def __init__(self, *, in_channels_src: int, in_channels_dst: int, hidden_dim: int, out_channels_dst: Optional[int] = None,
                 cpu_offload: bool = False, gradient_checkpointing: bool = True, layer_kernels=None, **kwargs) -> None:
        super().__init__()
        self.in_channels_src = in_channels_src
        self.in_channels_dst = in_channels_dst
        self.hidden_dim = hidden_dim
        self.out_channels_dst = out_channels_dst
        self.gradient_checkpointing = gradient_checkpointing
        self.layer_factory = load_layer_kernels(layer_kernels)
        if cpu_offload:
            raise NotImplementedError("cpu_offload is a training memory feature; not needed with 288 GB of HBM")
        self._local = _LocalGraphCache()
        self._heads_cache = None

    def _local_graph(self, x, shard_info, edge_attr, edge_index, group):
        """Rank-local (dst range, edges, compact sources) — index work only, cached for the static graph."""
        world, rank = comm_size(group), comm_rank(group)
        # keyed on the static edge_index only: in training edge_attr is a fresh (differentiable) tensor every step
        key = (edge_index.data_ptr(), version(edge_index), edge_attr.shape[0], world, rank,
               tuple(shard_info.src_nodes or ()), tuple(shard_info.dst_nodes or ()), tuple(shard_info.edges or ()),
               x[0].shape[0], x[1].shape[0])

        def build():
            partition = build_graph_partition_from_shard_info(edge_index, x, shard_info, group)
            if shard_info.edges_are_sharded():  # edge_index already holds this rank's edges (global ids)
                dr = partition.dst_range(rank)
                loc = edge_index.long()
                src_ids, inv = torch.unique(loc[0], return_inverse=True)
                ei_local = torch.stack([inv, loc[1] - dr.start])
                edge_rows = None
            else:
                lg = local_bipartite_graph(edge_index, partition, rank)
                src_ids, ei_local = lg.src_ids, lg.edge_index_local
                edge_rows = slice(lg.edge_range[0], lg.edge_range[1])
                if edge_rows == slice(0, edge_attr.shape[0]):
                    edge_rows = None
                dr = slice(*lg.dst_range)
            n_src_total = partition.num_nodes[0]
            all_connected = src_ids.shape[0] == n_src_total
            return dict(partition=partition, dst_range=(dr.start, dr.stop), src_ids=src_ids, src_ids32=src_ids.to(torch.int32).contiguous(),
                        edge_index=ei_local.contiguous(), edge_rows=edge_rows, all_connected=all_connected, anchors=(edge_index,),
                        # needed-rows exchange plans live WITH the graph they were built for (same key: edge_index, shard sizes,
                        # world, rank): a new graph / partition / group rebuilds them instead of reusing stale send indices
                        plans={})

        g = dict(self._local.get(key, build))
        g["edge_attr"] = edge_attr if g["edge_rows"] is None else edge_attr[g["edge_rows"]]
        return g


class GraphTransformerBaseMapper(BaseMapper):
    def __init__(self, *, in_channels_src: int, in_channels_dst: int, hidden_dim: int, out_channels_dst: Optional[int] = None,
                 num_chunks: int, num_heads: int, mlp_hidden_ratio: float, edge_dim: int, attn_channels: Optional[int] = None,
                 qk_norm: bool = False, mlp_implementation: str = "mlp", cpu_offload: bool = False, layer_kernels=None,
                 shard_strategy: str = "edges", graph_attention_backend: str = "hip", edge_pre_mlp: bool = False, **kwargs) -> None:
        super().__init__(in_channels_src=in_channels_src, in_channels_dst=in_channels_dst, hidden_dim=hidden_dim,
                         out_channels_dst=out_channels_dst, cpu_offload=cpu_offload, layer_kernels=layer_kernels, **kwargs)
        self.num_chunks = num_chunks
        assert shard_strategy in ["heads", "edges"], (
            f"Invalid shard strategy '{shard_strategy}' for {self.__class__.__name__}. Supported strategies are 'heads' and 'edges'."
        )
        self.shard_strategy = shard_strategy
        self.proc = GraphTransformerMapperBlock(
            in_channels=hidden_dim, hidden_dim=compute_mlp_hidden_dim(hidden_dim, mlp_hidden_ratio), out_channels=hidden_dim,
            attn_channels=attn_channels, num_heads=num_heads, edge_dim=edge_dim, qk_norm=qk_norm,
            mlp_implementation=mlp_implementation, layer_kernels=self.layer_factory, shard_strategy=shard_strategy,
            graph_attention_backend=graph_attention_backend, edge_pre_mlp=edge_pre_mlp,
        )
        self.emb_nodes_dst = self.layer_factory.Linear(self.in_channels_dst, self.hidden_dim)
        self._emb_src, self._emb_dst = PaddedLinear(), PaddedLinear()

    def _tail_projection(self):
        """(LayerNorm, Linear) of a post_process that is row-local and can ride at the end of the block's chain launch; None: there is none."""
        return None

    def _row_chain_ok(self, x: Tensor, lin, ln, projs: list) -> bool:
        """The embedding -> LayerNorm -> projection chain launch (ops.gt_row_chain) takes this side: inference, 16-bit, 512 channels, a
        plain affine LayerNorm, rows whose width is a multiple of 8 (the model pads its inputs), enough rows to fill the chip."""
        if not (_ROW_CHAIN and x.is_cuda and x.dim() == 2 and x.dtype != torch.float32 and x.shape[0] >= _ROW_CHAIN_MIN_ROWS
                and not (_ROW_CHAIN_GEMM_BAND[0] < x.shape[0] < _ROW_CHAIN_GEMM_BAND[1])
                and self.hidden_dim == ops.CHAIN_CHANNELS and type(ln).__name__ in ("LayerNorm", "AutocastLayerNorm") and ln.weight is not None):
            return False
        q_out = sum(p.out_features for p in projs)
        mods = [lin, ln, *projs]
        if not (lin.in_features <= x.shape[1] and lin.bias is not None and ops.gt_row_chain_supported(x, q_out)
                and all(p.in_features == ops.CHAIN_CHANNELS for p in projs)
                and all(q is None or q.dtype == x.dtype for m in mods for q in m.parameters())):
            return False
        return not (torch.is_grad_enabled() and (x.requires_grad or any(q.requires_grad for m in mods for q in m.parameters())))

    def _row_chain(self, x: Tensor, lin, side: str, want_x: bool):
        """(embedded rows or None, the block's fused projection of that side) in one launch, or None if the shapes do not fit."""
        blk = self.proc
        ln = blk.layer_norm_attention_src if side == "src" else blk.layer_norm_attention_dest
        projs = [blk.lin_key, blk.lin_value] if side == "src" else [blk.lin_query, blk.lin_self]
        if not self._row_chain_ok(x, lin, ln, projs):
            return None
        params = [lin.weight, lin.bias, ln.weight, ln.bias] + [q for m in projs for q in (m.weight, m.bias)]
        K = x.shape[1]

        def build():
            w = lin.weight if lin.in_features == K else torch.nn.functional.pad(lin.weight, (0, K - lin.in_features))  # (rows that carry alignment zeros)
            wq = torch.cat([m.weight for m in projs], dim=0)
            bq = torch.cat([m.bias if m.bias is not None else m.weight.new_zeros(m.out_features) for m in projs])
            wqg, dq = ops.fold_layer_norm(wq, bq, ln.weight, ln.bias)
            return ops.pack_embedding_frag(w), ops.pack_weight_frag(wqg), torch.cat([lin.bias.float(), dq]).to(x.dtype).contiguous(), wq.shape[0]

        we, wqg, vec, q_out = blk._fused.derived(f"rowchain:{side}:{K}", params, build)
        return ops.gt_row_chain(x, we, wqg, vec, q_out, ln.eps, want_x_out=want_x)

    def _embed(self, padded: PaddedLinear, x: Tensor, lin, side: str, ln_stats: Optional[dict]) -> Tensor:
        """The node embedding.  With ``ln_stats`` (inference): where the shapes fit, embedding, the block's LayerNorm on that side and its
        fused projection run as ONE row-resident launch (``ln_stats["proj:" + side]`` = the projection; the embedded SOURCE rows are not
        even written unless the block updates them) - else the embedding leaves the row statistics of its output there when the block's
        LayerNorm on that side can be folded into the GEMM behind it."""
        ln = self.proc.layer_norm_attention_src if side == "src" else self.proc.layer_norm_attention_dest
        if ln_stats is not None:
            want_x = side == "dst" or self.proc.update_src_nodes
            r = self._row_chain(x, lin, side, want_x)
            if r is not None:
                y, proj = r
                ln_stats["proj:" + side] = proj
                # (source rows nobody reads: an empty [N, 0] stand-in keeps the row count the block's graph plumbing looks at)
                return y if y is not None else x.new_empty((x.shape[0], 0))
        if ln_stats is not None and x.is_cuda and x.dtype != torch.float32 and self.proc._ln_fold_ok(ln, x):
            y, stats = padded.with_row_stats(x, lin)
            if stats is not None:
                ln_stats[side] = (y, stats)
            return y
        return padded(x, lin)

    # subclasses: pre_process(x_src_compact, x_dst) -> embedded pair, post_process(x_dst)
    @comm.scoped_forward
    def forward(self, x, batch_size: int, shard_info: BipartiteGraphShardInfo, edge_attr: Tensor, edge_index: Tensor,
                model_comm_group=None, keep_x_dst_sharded: bool = False, edges_are_dst_sorted: bool = True, cond=None, **kwargs):
        if self.shard_strategy == "heads" and model_is_distributed(model_comm_group):
            return self._forward_heads(x, batch_size, shard_info, edge_attr, edge_index, model_comm_group, keep_x_dst_sharded,
                                       edges_are_dst_sorted, cond=cond, **kwargs)
        x_src, x_dst = x
        edge_attr, edge_index = ensure_edges_are_dst_sorted(
            edge_attr, edge_index, edges_are_sharded=shard_info.edges_are_sharded(), model_comm_group=model_comm_group,
            edges_are_dst_sorted=edges_are_dst_sorted)
        g = self._local_graph(x, shard_info, edge_attr, edge_index, model_comm_group)
        d0, d1 = g["dst_range"]
        sharded = model_is_distributed(model_comm_group)
        if sharded and not shard_info.dst_is_sharded():
            x_dst = x_dst[d0:d1]
        # source rows this rank needs
        if sharded and shard_info.src_is_sharded():
            x_src_c, g["plans"]["src"] = comm.exchange_rows(x_src, g["src_ids"], shard_info.src_nodes, model_comm_group,
                                                            gather_fn=ops.gather_rows, plan=g["plans"].get("src"))
        elif g["all_connected"]:
            x_src_c = x_src
        else:
            x_src_c = ops.gather_rows(x_src, g["src_ids32"])
        if cond is not None:  # (cond_src, cond_dst) rows follow their nodes (reference mapper.py:279-329)
            c_src, c_dst = cond
            if sharded and not shard_info.dst_is_sharded():
                c_dst = c_dst[d0:d1]
            if sharded and shard_info.src_is_sharded():
                c_src, _ = comm.exchange_rows(c_src, g["src_ids"], shard_info.src_nodes, model_comm_group, gather_fn=ops.gather_rows,
                                              plan=g["plans"].get("src"))
            elif not g["all_connected"]:
                c_src = ops.gather_rows(c_src, g["src_ids32"])
            kwargs["cond"] = (c_src, c_dst)
        # inference: the embeddings also emit the row statistics of their outputs, and the block folds LayerNorm_src / _dst into
        # the k|v and q|self GEMMs (no LayerNorm launches on the 40 320-row side)
        ln_stats = {} if cond is None else None
        src_proj = kwargs.pop("src_proj", None)  # model glue: (source rows, their k|v projection) computed by the launch that produced the rows
        xs, xd = self.pre_process((x_src_c, x_dst), ln_stats=ln_stats)
        if src_proj is not None and ln_stats is not None and src_proj[0] is x_src and xs is x_src:
            ln_stats["proj:src"] = src_proj[1]
        tail = self._tail_projection() if (cond is None and "ln_chain" not in kwargs) else None
        if tail is not None:  # the block's chain launch may run post_process as its trailing projection (layers/block.py)
            kwargs["ln_chain"] = {"tail_proj": tail}
        (_, x_dst_out), _ = self.proc((xs, xd), g["edge_attr"], g["edge_index"], shard_info, batch_size,
                                      (xs.shape[0], xd.shape[0]), model_comm_group, edges_are_dst_sorted=True, ln_stats=ln_stats, **kwargs)
        out_dst = kwargs["ln_chain"].get("tail_out") if tail is not None else None
        if out_dst is None:
            out_dst = self.post_process(x_dst_out)
        if sharded and not keep_x_dst_sharded:
            out_dst = comm.gather_tensor(out_dst.contiguous(), 0, g["partition"].dst_splits, model_comm_group)
        return out_dst


    def _forward_heads(self, x, batch_size, shard_info, edge_attr, edge_index, group, keep_x_dst_sharded, edges_are_dst_sorted,
                       cond=None, **kwargs):
        """shard_strategy="heads" (reference mapper.py:388-444): source and destination rows sharded for the embeddings,
        projections and MLP; the attention runs on the whole bipartite graph for this rank's heads
        (``GraphTransformerBaseBlock._heads_attention``)."""
        # update_src_nodes: the block's source-side MLP is row-local (it runs on this rank's source rows like everywhere else);
        # like the reference's GraphTransformer mappers (mapper.py:445-477) only the destination rows are returned
        x_src, x_dst = x
        edge_attr, edge_index = ensure_edges_are_dst_sorted(
            edge_attr, edge_index, edges_are_sharded=shard_info.edges_are_sharded(), model_comm_group=group,
            edges_are_dst_sorted=edges_are_dst_sorted)
        src_sizes = shard_info.src_nodes if shard_info.src_is_sharded() else get_shard_sizes(x_src, 0, group)
        dst_sizes = shard_info.dst_nodes if shard_info.dst_is_sharded() else get_shard_sizes(x_dst, 0, group)
        if not shard_info.src_is_sharded():
            x_src = comm.shard_tensor(x_src, 0, src_sizes, group)
        if not shard_info.dst_is_sharded():
            x_dst = comm.shard_tensor(x_dst, 0, dst_sizes, group)
        if cond is not None:
            c_src, c_dst = cond
            if not shard_info.src_is_sharded():
                c_src = comm.shard_tensor(c_src, 0, src_sizes, group)
            if not shard_info.dst_is_sharded():
                c_dst = comm.shard_tensor(c_dst, 0, dst_sizes, group)
            kwargs["cond"] = (c_src, c_dst)
        train = ops._needs_grad(x_src, x_dst, edge_attr, self.proc.lin_edge.weight)
        if self._heads_cache is None:
            self._heads_cache = {}
        ei_full, ea_full = self.proc._heads_full_graph(edge_attr, edge_index, shard_info.edges if shard_info.edges_are_sharded() else None,
                                                       group, train, self._heads_cache)
        xs, xd = self.pre_process((x_src, x_dst))
        info = BipartiteGraphShardInfo(src_nodes=list(src_sizes), dst_nodes=list(dst_sizes), edges=None)
        (_, x_dst_out), _ = self.proc((xs, xd), ea_full, ei_full, info, batch_size, (sum(src_sizes), sum(dst_sizes)), group,
                                      edges_are_dst_sorted=True, **kwargs)
        out_dst = self.post_process(x_dst_out)
        if not keep_x_dst_sharded:
            out_dst = comm.gather_tensor(out_dst.contiguous(), 0, list(dst_sizes), group)
        return out_dst


class GraphTransformerForwardMapper(GraphTransformerBaseMapper):
    """Graph Transformer Mapper from data -> hidden."""

    def __init__(self, *, out_channels_dst: Optional[int] = None, **kwargs) -> None:
        assert out_channels_dst is None, "GraphTransformerForwardMapper does not support out_channels_dst."
        super().__init__(out_channels_dst=None, **kwargs)
        self.emb_nodes_src = self.layer_factory.Linear(self.in_channels_src, self.hidden_dim)

    def pre_process(self, x, ln_stats: Optional[dict] = None):
        x_src, x_dst = x
        return (self._embed(self._emb_src, x_src, self.emb_nodes_src, "src", ln_stats),
                self._embed(self._emb_dst, x_dst, self.emb_nodes_dst, "dst", ln_stats))

    def post_process(self, x_dst, **kwargs):
        return x_dst

    def forward(self, x, batch_size: int, shard_info: BipartiteGraphShardInfo, edge_attr: Tensor, edge_index: Tensor,
                model_comm_group=None, keep_x_dst_sharded: bool = True, **kwargs):
        x_dst = super().forward(x, batch_size, shard_info, edge_attr, edge_index, model_comm_group, keep_x_dst_sharded, **kwargs)
        return x[0], x_dst


class GraphTransformerBackwardMapper(GraphTransformerBaseMapper):
    """Graph Transformer Mapper from hidden -> data."""

    def __init__(self, *, out_channels_dst: Optional[int] = None, initialise_data_extractor_zero: bool = False, **kwargs) -> None:
        super().__init__(out_channels_dst=out_channels_dst, **kwargs)
        self.node_data_extractor = nn.Sequential(self.layer_factory.LayerNorm(self.hidden_dim),
                                                 self.layer_factory.Linear(self.hidden_dim, self.out_channels_dst))
        if initialise_data_extractor_zero:
            for module in self.node_data_extractor.modules():
                if isinstance(module, nn.Linear):
                    nn.init.constant_(module.weight, 0.0)
                    if module.bias is not None:
                        nn.init.constant_(module.bias, 0.0)

    def pre_process(self, x, ln_stats: Optional[dict] = None):
        x_src, x_dst = x
        return x_src, self._embed(self._emb_dst, x_dst, self.emb_nodes_dst, "dst", ln_stats)

    def _tail_projection(self):
        if not _TAIL_PROJ or torch.is_grad_enabled():
            return None
        return self.node_data_extractor[0], self.node_data_extractor[1]

    def post_process(self, x_dst):
        ln, lin = self.node_data_extractor[0], self.node_data_extractor[1]
        h = ops.layer_norm(x_dst, ln.weight, ln.bias, ln.eps)
        O = lin.out_features
        if O % 8 and h.is_cuda and h.dtype != torch.float32 and not ops._needs_grad(h, lin.weight, lin.bias):
            # inference: rows of the output padded to 16 bytes (a strided [N, O] view is returned), which is what the DMA-ring
            # GEMM kernels need for their epilogue - the variable count (84 at O96) is rarely a multiple of 8
            buf = torch.empty((h.shape[0], (O + 7) // 8 * 8), dtype=h.dtype, device=h.device)
            return ops.linear(h, lin.weight, lin.bias, out=buf[:, :O])
        return ops.linear(h, lin.weight, lin.bias)


# ============================================================================================ GNN mappers
class GNNBaseMapper(BaseMapper):
    def __init__(self, *, in_channels_src: int, in_channels_dst: int, hidden_dim: int, out_channels_dst: Optional[int] = None,
                 num_chunks: int, mlp_extra_layers: int, edge_dim: int, mlp_hidden_ratio: float = 1.0,
                 mlp_implementation: str = "mlp", cpu_offload: bool = False, layer_kernels=None, **kwargs) -> None:
        super().__init__(in_channels_src=in_channels_src, in_channels_dst=in_channels_dst, hidden_dim=hidden_dim,
                         out_channels_dst=out_channels_dst, cpu_offload=cpu_offload, layer_kernels=layer_kernels, **kwargs)
        self._mlp_hidden_dim = compute_mlp_hidden_dim(hidden_dim, mlp_hidden_ratio)
        self._mlp_kw = dict(layer_kernels=self.layer_factory, n_extra_layers=mlp_extra_layers + 1, mlp_implementation=mlp_implementation)
        self.emb_edges = MLP(in_features=edge_dim, hidden_dim=self._mlp_hidden_dim, out_features=hidden_dim, **self._mlp_kw)
        self._block_kw = dict(in_channels=hidden_dim, out_channels=hidden_dim, layer_kernels=self.layer_factory,
                              mlp_extra_layers=mlp_extra_layers, mlp_hidden_ratio=mlp_hidden_ratio,
                              mlp_implementation=mlp_implementation, num_chunks=num_chunks)

    def mapper_forward(self, x, batch_size, shard_info, edge_attr, edge_index, model_comm_group=None,
                       keep_x_dst_sharded: bool = False, edges_are_dst_sorted: bool = True, **kwargs):
        x_src, x_dst = x
        if model_is_distributed(model_comm_group):
            return self._mapper_forward_sharded(x_src, x_dst, shard_info, edge_attr, edge_index, model_comm_group,
                                                keep_x_dst_sharded, edges_are_dst_sorted)
        edge_attr, edge_index = ensure_edges_are_dst_sorted(edge_attr, edge_index, edges_are_sharded=False,
                                                           edges_are_dst_sorted=edges_are_dst_sorted)
        size = (x_src.shape[0], x_dst.shape[0])
        edge_attr = self.emb_edges(edge_attr)
        x_src, x_dst = self.pre_process((x_src, x_dst))
        (x_src, x_dst), edge_attr = self.proc((x_src, x_dst), edge_attr, edge_index, shard_info, model_comm_group, size=size, **kwargs)
        return x_src, self.post_process(x_dst)

    def _mapper_forward_sharded(self, x_src, x_dst, shard_info, edge_attr, edge_index, group, keep_x_dst_sharded, edges_are_dst_sorted):
        """Reference mapper.py:776-836 + GraphConvMapperBlock (block.py:441-479): nodes sharded in balanced contiguous
        ranges, every rank owns the edges of its destination range.  The node embeddings and the source update run on
        the local shards (work split over the ranks, as in the reference); where the reference then all-gathers every
        embedded source row (``sync_tensor``) only the rows this rank's edges touch are exchanged (``exchange_rows``,
        plan built once).  A source table that came in replicated is returned replicated."""
        edge_attr, edge_index = ensure_edges_are_dst_sorted(edge_attr, edge_index, edges_are_sharded=shard_info.edges_are_sharded(),
                                                           model_comm_group=group, edges_are_dst_sorted=edges_are_dst_sorted)
        g = self._local_graph((x_src, x_dst), shard_info, edge_attr, edge_index, group)
        d0, d1 = g["dst_range"]
        x_dst_loc = x_dst if shard_info.dst_is_sharded() else x_dst[d0:d1]
        src_was_sharded = shard_info.src_is_sharded()
        src_sizes = shard_info.src_nodes if src_was_sharded else get_shard_sizes(x_src, 0, group)
        x_src_loc = x_src if src_was_sharded else comm.shard_tensor(x_src, 0, src_sizes, group)
        e_loc = self.emb_edges(g["edge_attr"])
        xs_loc, xd_loc = self.pre_process((x_src_loc, x_dst_loc))
        pkey = ("src", tuple(src_sizes))
        xs_need, g["plans"][pkey] = comm.exchange_rows(xs_loc, g["src_ids"], src_sizes, group, gather_fn=ops.gather_rows, plan=g["plans"].get(pkey))
        (xs_new, xd_new), _ = self.proc.forward_local(xs_need, xd_loc, xs_loc, e_loc, g["edge_index"])
        out_dst = self.post_process(xd_new)
        if not keep_x_dst_sharded:
            out_dst = comm.gather_tensor(out_dst, 0, g["partition"].dst_splits, group)
        if not src_was_sharded:
            xs_new = comm.gather_tensor(xs_new, 0, src_sizes, group, reduce_in_backward=True)  # consumers slice it by THEIR partition
        return xs_new, out_dst

    @comm.scoped_forward
    def forward(self, x, batch_size: int, shard_info: BipartiteGraphShardInfo, edge_attr: Tensor, edge_index: Tensor,
                model_comm_group=None, keep_x_dst_sharded: bool = False, edges_are_dst_sorted: bool = True, **kwargs):
        return self.mapper_forward(x, batch_size, shard_info, edge_attr, edge_index, model_comm_group, keep_x_dst_sharded,
                                   edges_are_dst_sorted, **kwargs)


class GNNForwardMapper(GNNBaseMapper):
    """Graph Neural Network Mapper data -> hidden."""

    def __init__(self, **kwargs) -> None:
        super().__init__(**kwargs)
        self.proc = GraphConvMapperBlock(update_src_nodes=True, **self._block_kw)
        self.emb_nodes_src = MLP(in_features=self.in_channels_src, hidden_dim=self._mlp_hidden_dim, out_features=self.hidden_dim, **self._mlp_kw)
        self.emb_nodes_dst = MLP(in_features=self.in_channels_dst, hidden_dim=self._mlp_hidden_dim, out_features=self.hidden_dim, **self._mlp_kw)

    def pre_process(self, x):
        return self.emb_nodes_src(x[0]), self.emb_nodes_dst(x[1])

    def post_process(self, x_dst, **kwargs):
        return x_dst


class GNNBackwardMapper(GNNBaseMapper):
    """Graph Neural Network Mapper hidden -> data."""

    def __init__(self, **kwargs) -> None:
        super().__init__(**kwargs)
        self.proc = GraphConvMapperBlock(update_src_nodes=False, **self._block_kw)
        self.node_data_extractor = MLP(in_features=self.hidden_dim, hidden_dim=self._mlp_hidden_dim, out_features=self.out_channels_dst,
                                       layer_kernels=self.layer_factory, n_extra_layers=self._mlp_kw["n_extra_layers"],
                                       layer_norm=False, final_activation=False, mlp_implementation=self._mlp_kw["mlp_implementation"])

    def pre_process(self, x):
        return x

    def post_process(self, x_dst):
        return self.node_data_extractor(x_dst)

    def forward(self, x, batch_size: int, shard_info: BipartiteGraphShardInfo, edge_attr: Tensor, edge_index: Tensor,
                model_comm_group=None, keep_x_dst_sharded: bool = False, edges_are_dst_sorted: bool = True, **kwargs) -> Tensor:
        _, x_dst = super().forward(x, batch_size, shard_info, edge_attr, edge_index, model_comm_group, keep_x_dst_sharded,
                                   edges_are_dst_sorted=edges_are_dst_sorted, **kwargs)
        return x_dst
