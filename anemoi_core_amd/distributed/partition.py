"""dst-sorted graph partitioning: contiguous balanced node ranges + the edges whose destination is local.

Restates the integer bookkeeping of the reference's ``distributed/khop_edges.py`` (GraphPartition :51-151,
build_graph_partition :154-189, shard_edges_1hop :266-314, _drop_unconnected_src_nodes :474-500).  Because the
graph is static everything here is computed ONCE on the host and cached by the callers; the reference redoes
``degree()`` + ``.item()`` syncs and ``torch.unique`` on every forward (SURVEY.md §7 "legitimate wins").
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from .shapes import BipartiteGraphShardInfo, comm_rank, comm_size, get_balanced_partition_sizes, get_partition_range


def sort_edge_index_by_dst(edge_index: Tensor) -> tuple[Tensor, Tensor]:
    perm = torch.sort(edge_index[1], stable=True)[1]
    return edge_index[:, perm], perm


def is_edge_index_dst_sorted(edge_index: Tensor) -> bool:
    dst = edge_index[1]
    return True if dst.numel() <= 1 else bool(torch.all(dst[1:] >= dst[:-1]).item())


def ensure_edges_are_dst_sorted(edge_attr: Tensor, edge_index: Tensor, *, edges_are_sharded: bool, model_comm_group=None,
                                edges_are_dst_sorted: bool = True) -> tuple[Tensor, Tensor]:
    """khop_edges.py:236-262."""
    if edges_are_dst_sorted:
        return edge_attr, edge_index
    if edges_are_sharded and comm_size(model_comm_group) > 1:
        raise ValueError("Edge-sharded GraphTransformer inputs must be dst-sorted before use.")
    edge_index, perm = sort_edge_index_by_dst(edge_index)
    return edge_attr[perm], edge_index


@dataclass(frozen=True)
class GraphPartition:
    """Per-partition destination and edge counts of a dst-sorted edge list."""

    num_nodes: tuple
    num_edges: int
    num_parts: int
    dst_splits: list
    edge_splits: list

    def edge_range(self, pid: int) -> slice:
        return slice(*get_partition_range(self.edge_splits, pid))

    def dst_range(self, pid: int) -> slice:
        return slice(*get_partition_range(self.dst_splits, pid))


def build_graph_partition(edge_index: Tensor, num_parts: int, num_nodes: tuple) -> GraphPartition:
    n_dst = int(num_nodes[1])
    dst_splits = get_balanced_partition_sizes(n_dst, num_parts)
    deg = torch.bincount(edge_index[1].long(), minlength=n_dst).cpu()
    edge_splits = [int(c.sum()) for c in torch.split(deg, dst_splits)]
    return GraphPartition(tuple(int(n) for n in num_nodes), int(edge_index.shape[1]), num_parts, dst_splits, edge_splits)


def build_graph_partition_from_shard_info(edge_index: Tensor, x: tuple, shard_info: BipartiteGraphShardInfo,
                                          model_comm_group=None) -> GraphPartition:
    """khop_edges.py:192-233."""
    x_src, x_dst = x
    n_src = sum(shard_info.src_nodes) if shard_info.src_is_sharded() else x_src.shape[0]
    n_dst = sum(shard_info.dst_nodes) if shard_info.dst_is_sharded() else x_dst.shape[0]
    world = comm_size(model_comm_group)
    if shard_info.edges_are_sharded():
        dst_splits = shard_info.dst_nodes if shard_info.dst_is_sharded() else get_balanced_partition_sizes(n_dst, world)
        return GraphPartition((n_src, n_dst), sum(shard_info.edges), world, list(dst_splits), list(shard_info.edges))
    return build_graph_partition(edge_index, world, (n_src, n_dst))


def edge_shard_plan(edge_index: Tensor, src_size: int, dst_size: int, model_comm_group, edges_are_dst_sorted: bool = True):
    """Index part of ``shard_edges_1hop`` (a function of the static ``edge_index`` only, so callers cache it):
    (perm | None, slice | None, local edge_index, edge splits | None).  ``take_edge_rows`` applies it to edge attributes,
    which in training are a fresh differentiable tensor every step."""
    world = comm_size(model_comm_group)
    if world == 1:
        return None, None, edge_index, None
    perm = None
    if not edges_are_dst_sorted:
        edge_index, perm = sort_edge_index_by_dst(edge_index)
    part = build_graph_partition(edge_index, world, (src_size, dst_size))
    r = part.edge_range(comm_rank(model_comm_group))
    return perm, r, edge_index[:, r], part.edge_splits


def take_edge_rows(edge_attr: Tensor, perm, rows) -> Tensor:
    if perm is not None:
        edge_attr = edge_attr[perm]
    return edge_attr if rows is None else edge_attr[rows]


def shard_edges_1hop(edge_attr: Tensor, edge_index: Tensor, src_size: int, dst_size: int, model_comm_group,
                     edges_are_dst_sorted: bool = True):
    """Local slice of the dst-sorted edges owned by this rank (no communication): khop_edges.py:266-314."""
    perm, rows, ei, splits = edge_shard_plan(edge_index, src_size, dst_size, model_comm_group, edges_are_dst_sorted)
    return take_edge_rows(edge_attr, perm, rows), ei, splits


@dataclass(frozen=True)
class LocalBipartiteGraph:
    """This rank's share of a bipartite (mapper) graph: its dst range, the edges into it, and the source rows
    those edges touch, relabelled to a compact local numbering."""

    dst_range: tuple
    edge_range: tuple
    src_ids: Tensor  # [n_src_local] global ids of the connected sources, ascending
    edge_index_local: Tensor  # [2, M_local] (compact src, local dst)


def local_bipartite_graph(edge_index: Tensor, partition: GraphPartition, rank: int) -> LocalBipartiteGraph:
    """Slice + relabel (GraphPartition.materialise / shard_graph_to_local, khop_edges.py:78-132, 317-409) —
    the index part only; feature movement is done by the caller (needed-rows exchange instead of all-gather)."""
    er = partition.edge_range(rank)
    dr = partition.dst_range(rank)
    loc = edge_index[:, er].long()
    src_ids, inv = torch.unique(loc[0], return_inverse=True)
    return LocalBipartiteGraph((dr.start, dr.stop), (er.start, er.stop), src_ids, torch.stack([inv, loc[1] - dr.start]))
