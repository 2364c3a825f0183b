"""InputNormalizer — mirror of the reference's ``anemoi.models.preprocessing.normalizer.InputNormalizer``
(preprocessing/normalizer.py:24-252): same constructor arguments (processor config, data indices, statistics), same persistent
buffers (``_norm_mul``, ``_norm_add``, ``_input_idx``, ``_output_idx``, ``_model_output_idx``), same ``transform`` /
``inverse_transform`` semantics (which statistics a tensor of a given width takes).

On an MI355X tensor the arithmetic is one HIP kernel (``anemoi_affine_columns``); inside
``AnemoiModelEncProcDec.predict_step`` no kernel of its own runs at all: the transform is a column program of the input
assembly kernel, the inverse an op of the output column program (``column_program`` / ``inverse_program`` below)."""
from __future__ import annotations

import warnings
from typing import Optional

import numpy as np
import torch
from torch import Tensor

from . import BasePreprocessor

_METHODS = ("mean-std", "std", "min-max", "max", "none")


def _affine_table(minimum: np.ndarray, maximum: np.ndarray, mean: np.ndarray, stdev: np.ndarray) -> dict:
    """method -> (mul, add) over ALL variables at once (float64): x_normalised = x * mul + add.

    mean-std: (x - mean) / stdev;  std: x / stdev;  min-max: (x - minimum) / (maximum - minimum);  max: x / maximum;
    none: x (the documented normalisation methods of anemoi's ``normalizer`` processor)."""
    one, zero = np.ones_like(mean), np.zeros_like(mean)
    span = maximum - minimum
    with np.errstate(divide="ignore", invalid="ignore"):  # degenerate fields are reported below, only where they are used
        return {
            "mean-std": (one / stdev, -mean / stdev),
            "std": (one / stdev, zero),
            "min-max": (one / span, -minimum / span),
            "max": (one / maximum, zero),
            "none": (one, zero),
        }


def _as_index(v) -> Tensor:
    return torch.as_tensor(v).to(torch.int32).reshape(-1).clone()


class InputNormalizer(BasePreprocessor):
    """Per-variable affine normalisation of the dataset's variables, method chosen per variable by the config."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        position = self.data_indices.data.input.name_to_index  # variable name -> column of the statistics vectors
        names = np.array(list(position.keys()), dtype=object)
        cols = np.fromiter(position.values(), dtype=np.int64, count=len(position))

        rows = [np.asarray(statistics[k], dtype=np.float64).reshape(-1) for k in ("minimum", "maximum", "mean", "stdev")]
        assert len({r.size for r in rows}) == 1, f"InputNormalizer: statistics of different lengths {[r.size for r in rows]}"
        stats = np.stack(rows)
        # "remap": a variable borrows the statistics of another one; all sources are read before any column is
        # overwritten (one fancy-indexed gather then one scatter), so chains such as {a: b, b: c} do not depend on order
        if self.remap:
            takers = np.array([position[t] for t in self.remap.keys()], dtype=np.int64)
            givers = np.array([position[g] for g in self.remap.values()], dtype=np.int64)
            stats[:, takers] = stats[:, givers].copy()

        method_of = self.methods
        unknown = sorted(set(method_of) - set(position))
        assert not unknown, f"{unknown[0]} is not a valid variable name"
        invalid = sorted(set(method_of.values()) - set(_METHODS))
        assert not invalid, f"{invalid[0]} is not a valid normalisation method"
        assert self.default in _METHODS, f"{self.default} is not a valid normalisation method"
        listed = sum(len(v) for v in self.spec.variables.values())
        assert len(method_of) == listed, f"InputNormalizer: {listed - len(method_of)} variable(s) are listed under more than one method"

        table = _affine_table(*stats)
        chosen = np.array([_METHODS.index(method_of.get(n, self.default)) for n in names], dtype=np.int64)
        self._report_degenerate(names, cols, chosen, *stats)
        mul = np.ones(stats.shape[1], dtype=np.float32)
        add = np.zeros(stats.shape[1], dtype=np.float32)
        mul[cols] = np.stack([table[m][0] for m in _METHODS])[chosen, cols]
        add[cols] = np.stack([table[m][1] for m in _METHODS])[chosen, cols]

        self.register_buffer("_norm_mul", torch.from_numpy(mul), persistent=True)
        self.register_buffer("_norm_add", torch.from_numpy(add), persistent=True)
        self.register_buffer("_input_idx", _as_index(self.data_indices.data.input.full), persistent=True)
        self.register_buffer("_output_idx", _as_index(self.data_indices.data.output.full), persistent=True)
        # the variables the MODEL predicts, as the sub-sequence of the data-output index: a dataset may carry target-only
        # variables that appear in data.output but not in model.output
        predicted = self.data_indices.model.output.name_to_index
        keep = [i for n, i in self.data_indices.data.output.name_to_index.items() if n in predicted]
        sel = torch.isin(self._output_idx, torch.as_tensor(keep, dtype=torch.int32))
        assert int(sel.sum()) == len(keep), "InputNormalizer: a predicted variable is missing from the data-output index"
        self.register_buffer("_model_output_idx", self._output_idx[sel], persistent=True)
        self._gathered: dict = {}

    @staticmethod
    def _report_degenerate(names, cols, chosen, minimum, maximum, mean, stdev) -> None:
        """Warn about constant fields, per method family, without a Python loop over the variables."""
        m = np.array(_METHODS, dtype=object)[chosen]
        flat = np.isin(m, ("mean-std", "std")) & (stdev[cols] < mean[cols] * 1e-6)
        narrow = (m == "min-max") & ((maximum - minimum)[cols] < 1e-9)
        for n in names[flat | narrow]:
            warnings.warn(f"Normalizing: the field {n} seems to have only one value.")

    # ---------------------------------------------------------------------------------- which statistics for which tensor
    def _select(self, width: int, inverse: bool, data_index=None) -> tuple[Tensor, Tensor]:
        """(mul, add) fp32 vectors of length ``width`` for a tensor whose last dimension has ``width`` variables, following
        the reference's dispatch (normalizer.py:177-190, 217-252); gathered once per case and device."""
        if data_index is not None:
            idx = torch.as_tensor(data_index, device=self._norm_mul.device).long()
            return self._norm_mul[idx].contiguous(), self._norm_add[idx].contiguous()
        if not inverse:
            which = "_input_idx" if width == len(self._input_idx) else None
        else:
            which = "_output_idx" if width == len(self._output_idx) else ("_model_output_idx" if width == len(self._model_output_idx) else None)
        key = (which, str(self._norm_mul.device), self._norm_mul._version, self._norm_add._version)
        hit = self._gathered.get(key)
        if hit is None:
            if which is None:
                hit = (self._norm_mul.contiguous(), self._norm_add.contiguous())
            else:
                idx = getattr(self, which).long()
                hit = (self._norm_mul[idx].contiguous(), self._norm_add[idx].contiguous())
            self._gathered[key] = hit
        if hit[0].shape[0] != width:
            raise ValueError(f"InputNormalizer: a tensor with {width} variables matches none of the known index sets "
                             f"({len(self._input_idx)} inputs, {len(self._output_idx)} outputs, {len(self._model_output_idx)} model outputs, "
                             f"{self._norm_mul.shape[0]} data variables)")
        return hit

    def column_program(self, width: int) -> tuple[Tensor, Tensor]:
        """The transform as per-column (mul, add) for ``ops.assemble_input`` (model-edge fusion)."""
        return self._select(width, inverse=False)

    def inverse_program(self, width: int) -> list:
        """The inverse transform as ops of the output column program (kind 9: (x - add) / mul), one per column."""
        mul, add = self._select(width, inverse=True)
        return [(9, c, 0, float(a), float(m)) for c, (m, a) in enumerate(zip(mul.tolist(), add.tolist()))]

    # ---------------------------------------------------------------------------------- reference API
    def transform(self, x: Tensor, in_place: bool = True, data_index=None) -> Tensor:
        """x [..., nvars] -> x * mul + add (in place unless told otherwise)."""
        return self._affine(x, in_place, data_index, inverse=False)

    def inverse_transform(self, x: Tensor, in_place: bool = True, data_index=None) -> Tensor:
        """x [..., nvars | nvars_pred] -> (x - add) / mul."""
        return self._affine(x, in_place, data_index, inverse=True)

    def _affine(self, x: Tensor, in_place: bool, data_index, inverse: bool) -> Tensor:
        mul, add = self._select(x.shape[-1], inverse, data_index)
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.is_contiguous():
            from .. import ops

            return ops.affine_columns(x, mul, add, inverse=inverse, out=x if in_place else None)
        # host tensors (data loading side) and exotic layouts: the reference's own torch expression
        if not in_place:
            x = x.clone()
        if inverse:
            return x.subtract_(add.to(x.device)).div_(mul.to(x.device))
        return x.mul_(mul.to(x.device)).add_(add.to(x.device))
