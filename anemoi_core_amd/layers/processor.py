"""Processors — mirror of reference layers/processor.py (BaseProcessor :52-147, GNNProcessor :319-455,
GraphTransformerProcessor :458-626).  Same constructor keywords / forward signature / ``proc.{i}.*`` state_dict keys.

Activation checkpointing, CPU offload and dropout belong to training and are accepted but inert (forward-only path).
The static graph structure (CSC, packed edge features, halo plan) is derived once and shared by all layers."""
from __future__ import annotations

from typing import Optional

from torch import Tensor, nn

from ..distributed.partition import edge_shard_plan, ensure_edges_are_dst_sorted, sort_edge_index_by_dst, take_edge_rows
from ..distributed.primitives import gather_tensor, scoped_forward
from ..distributed.shapes import GraphShardInfo
from .block import GraphConvProcessorBlock, GraphTransformerProcessorBlock
from .utils import compute_mlp_hidden_dim, load_layer_kernels
from ..utils.tensors import version


class BaseProcessor(nn.Module):
    def __init__(self, *, num_layers: int, num_channels: int, num_chunks: int, cpu_offload: bool = False,
                 gradient_checkpointing: bool = True, layer_kernels=None, **kwargs) -> None:
        super().__init__()
        assert num_layers % num_chunks == 0, (
            f"Number of processor layers ({num_layers}) has to be divisible by the number of processor chunks ({num_chunks})."
        )
        self.num_layers = num_layers
        self.num_chunks = num_chunks
        self.chunk_size = num_layers // num_chunks
        self.num_channels = num_channels
        self.gradient_checkpointing = gradient_checkpointing
        self.layer_factory = load_layer_kernels(layer_kernels)
        if cpu_offload:
            raise NotImplementedError("cpu_offload is a training memory feature; not needed with 288 GB of HBM")

    def build_layers(self, layer_class, *layer_args, **layer_kwargs) -> None:
        self.proc = nn.ModuleList([layer_class(*layer_args, **layer_kwargs) for _ in range(self.num_layers)])

    def run_layers(self, data: tuple, *args, last_layer_kwargs: Optional[dict] = None, **kwargs) -> tuple:
        chain = kwargs.get("ln_chain")
        if chain is None:
            chain = kwargs.get("gnn_chain")
        after_last = kwargs.pop("after_last_block", None)  # model glue: the decoder's block, whose source-side projection may ride on the last tail
        for i, layer in enumerate(self.proc):
            extra = last_layer_kwargs if (last_layer_kwargs and i == len(self.proc) - 1) else {}
            if chain is not None:  # a block's chain launch may compute the NEXT block's LayerNorm + projections (layers/block.py)
                chain["next_block"] = self.proc[i + 1] if i + 1 < len(self.proc) else after_last
            data = layer(*data, *args, **kwargs, **extra)
        return data


class GraphTransformerProcessor(BaseProcessor):
    def __init__(self, *, num_layers: int, num_channels: int, num_chunks: int, num_heads: int, mlp_hidden_ratio: float,
                 edge_dim: int, attn_channels: Optional[int] = None, qk_norm: bool = False, mlp_implementation: str = "mlp",
                 cpu_offload: bool = False, layer_kernels=None, shard_strategy: str = "edges",
                 graph_attention_backend: str = "hip", edge_pre_mlp: bool = False, **kwargs) -> None:
        super().__init__(num_channels=num_channels, num_layers=num_layers, num_chunks=num_chunks, cpu_offload=cpu_offload,
                         layer_kernels=layer_kernels, **kwargs)
        assert shard_strategy in ["edges", "heads"], (
            f"Invalid shard strategy '{shard_strategy}' for {self.__class__.__name__}. Supported strategies are 'edges' and 'heads'."
        )
        self.shard_strategy = shard_strategy
        self.build_layers(
            GraphTransformerProcessorBlock,
            in_channels=num_channels,
            hidden_dim=compute_mlp_hidden_dim(num_channels, mlp_hidden_ratio),
            out_channels=num_channels,
            attn_channels=attn_channels,
            num_heads=num_heads,
            layer_kernels=self.layer_factory,
            qk_norm=qk_norm,
            mlp_implementation=mlp_implementation,
            shard_strategy=shard_strategy,
            graph_attention_backend=graph_attention_backend,
            edge_dim=edge_dim,
            edge_pre_mlp=edge_pre_mlp,
        )
        self._halo_cache: dict = {}
        self._shard_cache = None

    @scoped_forward
    def forward(self, x: Tensor, batch_size: int, shard_info: GraphShardInfo, edge_attr: Tensor, edge_index: Tensor,
                model_comm_group=None, edges_are_dst_sorted: bool = True, *args, **kwargs) -> Tensor:
        size = sum(shard_info.nodes) if shard_info.nodes_are_sharded() else x.shape[0]
        ln_chain = kwargs.pop("ln_chain", None)  # the encoder's last GEMM may have left the row statistics of x (model glue)
        latent_skip = kwargs.pop("latent_skip", None)  # model glue: x_latent, added by the LAST block's last GEMM (returns x + skip)
        edge_attr, edge_index = ensure_edges_are_dst_sorted(
            edge_attr, edge_index, edges_are_sharded=shard_info.edges_are_sharded(), model_comm_group=model_comm_group,
            edges_are_dst_sorted=edges_are_dst_sorted,
        )
        if not shard_info.edges_are_sharded():  # local slice of the dst-sorted edges (no communication); index part cached
            key = (edge_index.data_ptr(), version(edge_index), size, id(model_comm_group))
            if self._shard_cache is None or self._shard_cache[0] != key:
                self._shard_cache = (key, edge_shard_plan(edge_index, size, size, model_comm_group, edges_are_dst_sorted=True), (edge_index,))
            perm, rows, edge_index, edge_shard_sizes = self._shard_cache[1]
            edge_attr = take_edge_rows(edge_attr, perm, rows)
            shard_info = GraphShardInfo(nodes=shard_info.nodes, edges=edge_shard_sizes)
        x, _ = self.run_layers(
            (x, edge_attr), edge_index=edge_index, shard_info=shard_info, batch_size=batch_size, size=size,
            model_comm_group=model_comm_group, edges_are_dst_sorted=True, halo_cache=self._halo_cache, edge_prep={},
            ln_chain={} if ln_chain is None else ln_chain,  # row statistics handed from a block's last GEMM to the next block's first (LayerNorm fold)
            last_layer_kwargs=None if latent_skip is None else {"extra_residual": latent_skip},
            **kwargs,
        )
        return x


class GNNProcessor(BaseProcessor):
    def __init__(self, *, num_channels: int, num_layers: int, num_chunks: int, mlp_extra_layers: int, edge_dim: int,
                 mlp_hidden_ratio: float = 1.0, mlp_implementation: str = "mlp", cpu_offload: bool = False,
                 layer_kernels=None, **kwargs) -> None:
        super().__init__(num_channels=num_channels, num_layers=num_layers, num_chunks=num_chunks, cpu_offload=cpu_offload,
                         layer_kernels=layer_kernels, **kwargs)
        kwargs_build = dict(mlp_extra_layers=mlp_extra_layers, mlp_hidden_ratio=mlp_hidden_ratio,
                            mlp_implementation=mlp_implementation, layer_kernels=self.layer_factory, edge_dim=None)
        self.build_layers(GraphConvProcessorBlock, in_channels=num_channels, out_channels=num_channels, num_chunks=1, **kwargs_build)
        kwargs_build["edge_dim"] = edge_dim  # only the first layer embeds the raw edge attributes
        self.proc[0] = GraphConvProcessorBlock(in_channels=num_channels, out_channels=num_channels, num_chunks=1, **kwargs_build)
        self._shard_cache = None
        self._local_edge_cache: dict = {}

    @scoped_forward
    def forward(self, x: Tensor, batch_size: int, shard_info: GraphShardInfo, edge_attr: Tensor, edge_index: Tensor,
                model_comm_group=None, edges_are_dst_sorted: bool = True, *args, **kwargs) -> Tensor:
        if not shard_info.edges_are_sharded():  # local slice of the dst-sorted edges (no communication), cached: static graph
            target_nodes = sum(shard_info.nodes) if shard_info.nodes_are_sharded() else x.shape[0]
            key = (edge_index.data_ptr(), version(edge_index), target_nodes, id(model_comm_group), edges_are_dst_sorted)
            if self._shard_cache is None or self._shard_cache[0] != key:
                perm, rows, ei, edge_shard_sizes = edge_shard_plan(edge_index, target_nodes, target_nodes, model_comm_group,
                                                                   edges_are_dst_sorted=edges_are_dst_sorted)
                if edge_shard_sizes is None and not edges_are_dst_sorted:
                    ei, perm = sort_edge_index_by_dst(ei)
                self._shard_cache = (key, (perm, rows, ei, edge_shard_sizes), (edge_index,))
            perm, rows, edge_index, edge_shard_sizes = self._shard_cache[1]
            edge_attr = take_edge_rows(edge_attr, perm, rows)
            shard_info = GraphShardInfo(nodes=shard_info.nodes, edges=edge_shard_sizes)
        x, _ = self.run_layers((x, edge_attr), edge_index, shard_info, model_comm_group, local_edge_cache=self._local_edge_cache, gnn_chain={}, **kwargs)
        return x
