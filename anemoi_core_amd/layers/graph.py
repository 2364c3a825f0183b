"""Node attributes — mirror of reference layers/graph.py:20-118 (TrainableTensor, NamedNodesAttributes)."""
from __future__ import annotations

from collections import defaultdict

import torch
from torch import Tensor, nn
from ..utils.tensors import version


class TrainableTensor(nn.Module):
    def __init__(self, tensor_size: int, trainable_size: int) -> None:
        super().__init__()
        if trainable_size > 0:
            trainable = nn.Parameter(torch.zeros(tensor_size, trainable_size))
        else:
            trainable = None
        self.register_parameter("trainable", trainable)
        self._cache = None

    def forward(self, x: Tensor, batch_size: int) -> Tensor:
        """cat[x repeated B times, trainable repeated B times] along features; cached while the inputs are unchanged
        (the result is a pure function of static buffers and parameters)."""
        t = self.trainable
        if t is not None and t.requires_grad and torch.is_grad_enabled():  # training: differentiable, never cached
            return torch.cat([x.repeat(batch_size, 1), t.to(device=x.device, dtype=x.dtype).repeat(batch_size, 1)], dim=-1)
        key = (x.data_ptr(), version(x), x.dtype, str(x.device), batch_size, None if t is None else (t.data_ptr(), version(t), t.dtype))
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        with torch.no_grad():
            latent = [x.repeat(batch_size, 1)]
            if t is not None:
                latent.append(t.to(device=x.device, dtype=x.dtype).repeat(batch_size, 1))
            out = torch.cat(latent, dim=-1).contiguous()
        self._cache = (key, out, x)
        return out


class NamedNodesAttributes(nn.Module):
    """``latlons_<name>`` buffers = [sin(latlon), cos(latlon)] + trainable tensors, per node set."""

    def __init__(self, trainable_parameters: dict, graph_nodes: dict) -> None:
        """``graph_nodes``: {name: coords [N, 2] in radians (lat, lon)}."""
        super().__init__()
        trainable_parameters = defaultdict(int, trainable_parameters)
        self.num_nodes = {name: int(x.shape[0]) for name, x in graph_nodes.items()}
        self.attr_ndims = {name: 2 * x.shape[1] + trainable_parameters[name] for name, x in graph_nodes.items()}
        self.trainable_tensors = nn.ModuleDict()
        for name, coords in graph_nodes.items():
            coords = torch.as_tensor(coords, dtype=torch.float32)
            self.register_buffer(f"latlons_{name}", torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1), persistent=True)
            self.trainable_tensors[name] = TrainableTensor(self.num_nodes[name], trainable_parameters[name])

    def get_coordinates(self, name: str) -> Tensor:
        sc = getattr(self, f"latlons_{name}")
        n = sc.shape[1] // 2
        return torch.atan2(sc[:, :n], sc[:, n:])

    def forward(self, name: str, batch_size: int) -> Tensor:
        return self.trainable_tensors[name](getattr(self, f"latlons_{name}"), batch_size)
