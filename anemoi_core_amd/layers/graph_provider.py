"""Static graph provider — mirror of reference layers/graph_provider.py:145-291.

Owns the dst-sorted edge list, its attributes and the trainable edge tensor.  ``get_edges`` returns
``(edge_attr [B*M, F+trainable], edge_index int64 [2, B*M], edge_shard_sizes)`` like the reference, but the tensors
are cached across calls (static graph, static parameters at inference), which lets every downstream cache
(CSC, packed edge features, halo plan, local mapper graph) hit."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from ..distributed.partition import edge_shard_plan, take_edge_rows, sort_edge_index_by_dst
from .graph import TrainableTensor
from ..utils.tensors import version


class StaticGraphProvider(nn.Module):
    _TRAINABLE_LAYOUT_VERSION = 1
    _TRAINABLE_LAYOUT_VERSION_KEY = "trainable_layout_version"

    def __init__(self, edge_index: Tensor, edge_attr: Tensor, src_size: int, dst_size: int, trainable_size: int) -> None:
        """``edge_index`` [2, M] (src, dst) any order; ``edge_attr`` [M, F] the concatenated sub-graph attributes."""
        super().__init__()
        edge_index = torch.as_tensor(edge_index).long()
        edge_attr = torch.as_tensor(edge_attr, dtype=torch.float32)
        edge_index, perm = sort_edge_index_by_dst(edge_index)  # once, at init (graph_provider.py:185-187)
        self.register_buffer("perm", perm, persistent=False)
        self.register_buffer("edge_attr", edge_attr.index_select(0, perm).contiguous(), persistent=False)
        self.register_buffer("edge_index_base", edge_index.contiguous(), persistent=False)
        self.register_buffer("edge_inc", torch.tensor([[src_size], [dst_size]], dtype=torch.int64), persistent=False)
        self.register_buffer(self._TRAINABLE_LAYOUT_VERSION_KEY, torch.tensor(self._TRAINABLE_LAYOUT_VERSION, dtype=torch.int64), persistent=True)
        self.trainable = TrainableTensor(trainable_size=trainable_size, tensor_size=edge_attr.shape[0])
        self._edge_dim = edge_attr.shape[1] + trainable_size
        self._sizes = (int(src_size), int(dst_size))
        self._cache: dict = {}

    @property
    def edge_dim(self) -> int:
        return self._edge_dim

    @property
    def is_sparse(self) -> bool:
        return False

    def _apply(self, fn, recurse=True):
        """Keep the geometric edge attributes in fp32 when the model is cast to bf16/fp16: the fused attention consumes
        fp32 edge features anyway (the reference's autocast likewise leaves edge_attr in fp32)."""
        ea = self.edge_attr
        super()._apply(fn, recurse)
        if self.edge_attr.dtype != torch.float32:
            self.edge_attr = ea.to(self.edge_attr.device)
        self._cache.clear()
        return self

    def get_edges(self, batch_size: int, src_coords=None, dst_coords=None, model_comm_group=None, shard_edges: bool = True,
                  act_checkpoint: bool = True):
        edge_attr = self.trainable(self.edge_attr, batch_size)  # cached by TrainableTensor outside training
        key = (batch_size, shard_edges, id(model_comm_group))  # index part only: edge_attr is a fresh tensor per training step
        hit = self._cache.get("edges")
        if hit is None or hit[0] != key:
            if batch_size == 1:
                edge_index = self.edge_index_base
            else:
                edge_index = torch.cat([self.edge_index_base + i * self.edge_inc for i in range(batch_size)], dim=1)
            plan = (None, None, edge_index, None)
            if shard_edges:
                src_size, dst_size = self._sizes
                plan = edge_shard_plan(edge_index, src_size * batch_size, dst_size * batch_size, model_comm_group)
            hit = self._cache["edges"] = (key, plan)
        perm, rows, edge_index, splits = hit[1]
        return take_edge_rows(edge_attr, perm, rows), edge_index, splits


class NoOpGraphProvider(nn.Module):
    @property
    def edge_dim(self) -> int:
        return 0

    def get_edges(self, *args, **kwargs):
        return None, None, None


def create_graph_provider(graph=None, edge_attributes: Optional[list] = None, src_size: Optional[int] = None,
                          dst_size: Optional[int] = None, trainable_size: int = 0):
    """``graph``: an edge store with ``edge_index`` and the named attribute tensors (a HeteroData edge store or a dict)."""
    if not graph:
        return NoOpGraphProvider()
    get = (lambda k: graph[k]) if isinstance(graph, dict) else (lambda k: getattr(graph, k) if hasattr(graph, k) else graph[k])
    assert edge_attributes is not None, "Edge attributes must be provided"
    ea = torch.cat([torch.as_tensor(get(a), dtype=torch.float32) for a in edge_attributes], dim=1)
    return StaticGraphProvider(torch.as_tensor(get("edge_index")), ea, src_size, dst_size, trainable_size)
