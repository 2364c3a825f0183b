"""Shard metadata crossing the module boundary — same dataclasses as the reference
(models/src/anemoi/models/distributed/shapes.py:23-71, balanced_partition.py:16-102)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Union

import torch.distributed as dist
from torch import Tensor

ShardSizes = Union[list, None]
DatasetShardSizes = dict


def get_balanced_partition_sizes(total_size: int, n_partitions: int) -> list[int]:
    """Sizes differ by at most one; the first ``total % n`` parts get the extra element."""
    base, rem = divmod(int(total_size), int(n_partitions))
    return [base + 1] * rem + [base] * (n_partitions - rem)


def get_partition_range(partition_sizes: list[int], partition_id: int, offset: int = 0) -> tuple[int, int]:
    if partition_id < 0 or partition_id >= len(partition_sizes):
        raise ValueError(f"Invalid partition ID {partition_id}, expected in [0, {len(partition_sizes)})")
    start = sum(partition_sizes[:partition_id]) + offset
    return start, start + partition_sizes[partition_id]


def get_balanced_partition_range(total_size: int, n_partitions: int, partition_id: int, offset: int = 0) -> tuple[int, int]:
    return get_partition_range(get_balanced_partition_sizes(total_size, n_partitions), partition_id, offset)


@dataclass(frozen=True)
class GraphShardInfo:
    nodes: ShardSizes = None
    edges: ShardSizes = None

    def nodes_are_sharded(self) -> bool:
        return self.nodes is not None

    def edges_are_sharded(self) -> bool:
        return self.edges is not None


@dataclass(frozen=True)
class BipartiteGraphShardInfo:
    src_nodes: ShardSizes = None
    dst_nodes: ShardSizes = None
    edges: ShardSizes = None

    def src_is_sharded(self) -> bool:
        return self.src_nodes is not None

    def dst_is_sharded(self) -> bool:
        return self.dst_nodes is not None

    def edges_are_sharded(self) -> bool:
        return self.edges is not None


def comm_size(group) -> int:
    return 1 if group is None else dist.get_world_size(group=group)


def comm_rank(group) -> int:
    return 0 if group is None else dist.get_rank(group=group)


def model_is_distributed(group) -> bool:
    return group is not None and comm_size(group) > 1


def get_shard_sizes(tensor: Tensor, dim: int, model_comm_group=None) -> ShardSizes:
    assert dim < tensor.dim(), f"tensor has {tensor.dim()} dims, cannot split along {dim}"
    return get_balanced_partition_sizes(tensor.shape[dim], comm_size(model_comm_group))
