"""Halo-exchange metadata for the sharded hidden mesh (processor, shard_strategy="edges").

Restates reference ``distributed/halo.py:24-222``: nodes are split into contiguous balanced ranges, every rank owns
the edges whose destination is local, and needs the source rows owned by peers ("halo").  Local numbering:
inner nodes [0, n_local), halo nodes [n_local, n_local + n_halo) ordered by owning rank then global id.  The
processor graph is symmetric, so the rows this rank must SEND to rank r are the local endpoints of its cut edges
whose source lives on r (halo.py:171-175).

Built once per (graph, world size, rank) and cached; also carries the int32 device buffers the HIP path needs
(packed send index, CSC of the relabelled local edges).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from .partition import GraphPartition
from .shapes import get_partition_range


@dataclass(frozen=True)
class HaloInfo:
    num_local_nodes: int
    num_halo_nodes: int
    send_indices: tuple  # per-rank local indices (int64) of the inner rows to send
    recv_counts: tuple
    recv_global_ids: Optional[tuple]
    edge_index_local: Tensor  # [2, M_local]; src in [0, total), dst in [0, n_local)

    @property
    def total_nodes(self) -> int:
        return self.num_local_nodes + self.num_halo_nodes

    @property
    def send_counts(self) -> tuple:
        return tuple(int(t.shape[0]) for t in self.send_indices)


def build_halo_info(partition: GraphPartition, edge_index: Tensor, rank: int, edges_are_local: bool = False,
                    debug: bool = False) -> HaloInfo:
    """``edge_index``: GLOBAL node ids, dst-sorted; the full edge list, or (edges_are_local) this rank's slice."""
    world = partition.num_parts
    local = edge_index if edges_are_local else edge_index[:, partition.edge_range(rank)]
    local = local.long()
    d0, d1 = get_partition_range(partition.dst_splits, rank)
    n_local = d1 - d0
    src, dst = local[0], local[1]
    is_halo = (src < d0) | (src >= d1)
    halo_src, halo_dst = src[is_halo], dst[is_halo]
    bounds = torch.cumsum(torch.tensor(partition.dst_splits, dtype=torch.long, device=src.device), 0)
    owner = torch.searchsorted(bounds, halo_src, right=True)
    send, recv = [], []
    for r in range(world):
        mask = owner == r
        recv.append(halo_src[mask].unique(sorted=True))
        send.append(halo_dst[mask].unique(sorted=True) - d0)
    halo_nodes = torch.cat(recv) if recv else src.new_zeros(0)
    n_halo = int(halo_nodes.shape[0])
    new_src = src - d0
    if n_halo > 0:
        relabel = torch.empty(partition.num_nodes[0], dtype=torch.long, device=src.device)
        relabel[halo_nodes] = torch.arange(n_halo, device=src.device) + n_local
        new_src = torch.where(is_halo, relabel[src], new_src)
    return HaloInfo(
        num_local_nodes=n_local,
        num_halo_nodes=n_halo,
        send_indices=tuple(send),
        recv_counts=tuple(int(t.shape[0]) for t in recv),
        recv_global_ids=tuple(recv) if debug else None,
        edge_index_local=torch.stack([new_src, dst - d0]),
    )
