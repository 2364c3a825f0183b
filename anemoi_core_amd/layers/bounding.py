"""Output boundings — mirror of the reference's ``anemoi.models.layers.bounding`` (bounding.py:26-307): same class names
and constructor keywords.  Each bounding describes itself as a COLUMN PROGRAM (kind, column, total column, parameters);
the model concatenates the programs of all configured boundings and runs them in one in-place kernel over the output
tensor (``ops.bound_columns_``) instead of one indexed read-modify-write per bounding.  With autograd recording the same
program is evaluated with differentiable torch ops (the model edge is not a hot spot of the backward pass)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .. import ops

RELU, LEAKY_RELU, NORM_RELU, NORM_LEAKY_RELU, HARDTANH, LEAKY_HARDTANH, FRACTION, LEAKY_FRACTION = range(1, 9)


def _leaky_hardtanh(x: Tensor, lo: float, hi: float, slope: float = 0.01) -> Tensor:  # layers/activations.py:16-42
    y = torch.clamp(x, lo, hi)
    y = torch.where(x < lo, lo + slope * (x - lo), y)
    return torch.where(x > hi, hi + slope * (x - hi), y)


def apply_program_torch(x: Tensor, program: list) -> Tensor:
    """Differentiable, out-of-place evaluation of a column program (training path and CPU-side checks)."""
    cols = list(x.unbind(-1))
    for kind, col, tot, p0, p1 in program:
        v = cols[col]
        if kind == RELU:
            v = torch.relu(v)
        elif kind == LEAKY_RELU:
            v = torch.nn.functional.leaky_relu(v)
        elif kind == NORM_RELU:
            v = torch.relu(v - p0) + p0
        elif kind == NORM_LEAKY_RELU:
            v = torch.nn.functional.leaky_relu(v - p0) + p0
        elif kind == HARDTANH:
            v = torch.clamp(v, p0, p1)
        elif kind == LEAKY_HARDTANH:
            v = _leaky_hardtanh(v, p0, p1)
        elif kind == FRACTION:
            v = torch.clamp(v, p0, p1) * cols[tot]
        elif kind == LEAKY_FRACTION:
            v = _leaky_hardtanh(v, p0, p1) * cols[tot]
        cols[col] = v
    return torch.stack(cols, dim=-1)


def program_tables(program: list, device) -> tuple[Tensor, Tensor]:
    ops_t = torch.tensor([[k, c, t, 0] for k, c, t, _, _ in program], dtype=torch.int32, device=device).reshape(-1, 4)
    par_t = torch.tensor([[a, b] for _, _, _, a, b in program], dtype=torch.float32, device=device).reshape(-1, 2)
    return ops_t, par_t


class BaseBounding(nn.Module):
    kind = 0

    def __init__(self, *, variables: list, name_to_index: dict, statistics: Optional[dict] = None,
                 name_to_index_stats: Optional[dict] = None) -> None:
        super().__init__()
        self.name_to_index = name_to_index
        self.variables = variables
        self.data_index = self._create_index(variables)
        self.statistics = statistics
        self.name_to_index_stats = name_to_index_stats
        self._tables = None

    def _create_index(self, variables: list) -> Tensor:  # bounding.py:62-63: order of name_to_index, absent names skipped
        return torch.tensor([i for name, i in self.name_to_index.items() if name in variables], dtype=torch.int)

    def program(self) -> list:
        return [(self.kind, int(c), 0, 0.0, 0.0) for c in self.data_index.tolist()]

    def forward(self, x: Tensor) -> Tensor:
        prog = self.program()
        if torch.is_grad_enabled() and x.requires_grad:
            return apply_program_torch(x, prog)
        if self._tables is None or self._tables[0].device != x.device:
            self._tables = program_tables(prog, x.device)
        return ops.bound_columns_(x if x.is_contiguous() else x.contiguous(), *self._tables)


class ReluBounding(BaseBounding):
    kind = RELU


class LeakyReluBounding(BaseBounding):
    kind = LEAKY_RELU


class NormalizedReluBounding(BaseBounding):
    kind = NORM_RELU

    def __init__(self, *, variables: list, name_to_index: dict, min_val: list, normalizer: list, statistics: dict,
                 name_to_index_stats: dict) -> None:
        if len(normalizer) != len(variables):
            raise ValueError("The length of the normalizer list must match the number of variables in NormalizedReluBounding.")
        if len(min_val) != len(variables):
            raise ValueError("The length of the min_val list must match the number of variables in NormalizedReluBounding.")
        if not all(norm in {"mean-std", "min-max", "max", "std"} for norm in normalizer):
            raise ValueError("Each normalizer must be one of: 'mean-std', 'min-max', 'max', 'std' in NormalizedReluBounding.")
        super().__init__(variables=variables, name_to_index=name_to_index, statistics=statistics, name_to_index_stats=name_to_index_stats)
        kept = [(i, v) for i, v in enumerate(variables) if v in name_to_index]
        self.variables = [v for _, v in kept]
        self.min_val = [min_val[i] for i, _ in kept]
        self.normalizer = [normalizer[i] for i, _ in kept]
        self.data_index = torch.tensor([name_to_index[v] for v in self.variables], dtype=torch.int)  # configuration order
        nmv = torch.zeros(len(self.variables), dtype=torch.float32)
        for i, v in enumerate(self.variables):  # bounding.py:157-172
            si = name_to_index_stats[v]
            how = self.normalizer[i]
            if how == "mean-std":
                nmv[i] = (self.min_val[i] - statistics["mean"][si]) / statistics["stdev"][si]
            elif how == "min-max":
                nmv[i] = (self.min_val[i] - statistics["min"][si]) / (statistics["max"][si] - statistics["min"][si])
            elif how == "max":
                nmv[i] = self.min_val[i] / statistics["max"][si]
            else:
                nmv[i] = self.min_val[i] / statistics["stdev"][si]
        self.register_buffer("norm_min_val", nmv)

    def program(self) -> list:
        return [(self.kind, int(c), 0, float(m), 0.0) for c, m in zip(self.data_index.tolist(), self.norm_min_val.tolist())]


class NormalizedLeakyReluBounding(NormalizedReluBounding):
    kind = NORM_LEAKY_RELU


class HardtanhBounding(BaseBounding):
    kind = HARDTANH

    def __init__(self, *, variables: list, name_to_index: dict, min_val: float, max_val: float, statistics: Optional[dict] = None,
                 name_to_index_stats: Optional[dict] = None) -> None:
        super().__init__(variables=variables, name_to_index=name_to_index)
        self.min_val, self.max_val = min_val, max_val

    def program(self) -> list:
        return [(self.kind, int(c), 0, float(self.min_val), float(self.max_val)) for c in self.data_index.tolist()]


class LeakyHardtanhBounding(HardtanhBounding):
    kind = LEAKY_HARDTANH


class FractionBounding(HardtanhBounding):
    kind = FRACTION

    def __init__(self, *, variables: list, name_to_index: dict, min_val: float, max_val: float, total_var: str,
                 statistics: Optional[dict] = None, name_to_index_stats: Optional[dict] = None) -> None:
        super().__init__(variables=variables, name_to_index=name_to_index, min_val=min_val, max_val=max_val)
        self.total_variable = self._create_index([total_var])

    def program(self) -> list:
        tot = int(self.total_variable.tolist()[0])
        return [(self.kind, int(c), tot, float(self.min_val), float(self.max_val)) for c in self.data_index.tolist()]


class LeakyFractionBounding(FractionBounding):
    kind = LEAKY_FRACTION


def build_boundings_for(cfgs, name_to_index: dict, statistics, name_to_index_stats) -> nn.ModuleList:
    """bounding.py:312-372 for one dataset: instantiate every configured bounding (the reference's ``_target_`` strings are
    accepted) with the shared keyword arguments injected."""
    mods = []
    for c in cfgs or []:
        c = dict(c)
        target = str(c.pop("_target_")).rsplit(".", 1)[-1]
        cls = globals().get(target)
        if cls is None or not (isinstance(cls, type) and issubclass(cls, BaseBounding)):
            raise NotImplementedError(f"unknown bounding '{target}'")
        mods.append(cls(name_to_index=name_to_index, statistics=statistics, name_to_index_stats=name_to_index_stats, **c))
    return nn.ModuleList(mods)
