"""InputNormalizer — mirror of the reference's ``anemoi.models.preprocessing.normalizer.InputNormalizer``
(preprocessing/normalizer.py:24-252): same constructor (processor config, data indices, statistics), same persistent
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


def _as_index(v) -> Tensor:
    return torch.as_tensor(v).to(torch.int32).reshape(-1).clone()


class InputNormalizer(BasePreprocessor):
    """Normalizes input data with a configurable method per variable."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__(config, data_indices, statistics)
        name_to_index = self.data_indices.data.input.name_to_index
        minimum, maximum = np.array(statistics["minimum"], dtype=np.float64), np.array(statistics["maximum"], dtype=np.float64)
        mean, stdev = np.array(statistics["mean"], dtype=np.float64), np.array(statistics["stdev"], dtype=np.float64)

        # optionally reuse the statistics of one variable for another one (two steps: independent of the order)
        remapped = {}
        for remap, source in self.remap.items():
            s, r = name_to_index[source], name_to_index[remap]
            remapped[r] = (minimum[s], maximum[s], mean[s], stdev[s])
        for idx, st in remapped.items():
            minimum[idx], maximum[idx], mean[idx], stdev[idx] = st
        self._validate_normalization_inputs(name_to_index, minimum, maximum, mean, stdev)

        mul = np.ones((minimum.size,), dtype=np.float32)
        add = np.zeros((minimum.size,), dtype=np.float32)
        for name, i in name_to_index.items():
            method = self.methods.get(name, self.default)
            if method == "mean-std":
                if stdev[i] < (mean[i] * 1e-6):
                    warnings.warn(f"Normalizing: the field seems to have only one value {mean[i]}")
                mul[i] = 1 / stdev[i]
                add[i] = -mean[i] / stdev[i]
            elif method == "std":
                if stdev[i] < (mean[i] * 1e-6):
                    warnings.warn(f"Normalizing: the field seems to have only one value {mean[i]}")
                mul[i] = 1 / stdev[i]
                add[i] = 0
            elif method == "min-max":
                x = maximum[i] - minimum[i]
                if x < 1e-9:
                    warnings.warn(f"Normalizing: the field {name} seems to have only one value {maximum[i]}.")
                mul[i] = 1 / x
                add[i] = -minimum[i] / x
            elif method == "max":
                mul[i] = 1 / maximum[i]
            elif method == "none":
                pass
            else:
                raise ValueError(f"Unknown normalisation method for {name}: {method}")

        self.register_buffer("_norm_mul", torch.from_numpy(mul), persistent=True)
        self.register_buffer("_norm_add", torch.from_numpy(add), persistent=True)
        self.register_buffer("_input_idx", _as_index(self.data_indices.data.input.full), persistent=True)
        self.register_buffer("_output_idx", _as_index(self.data_indices.data.output.full), persistent=True)

        # variables the MODEL predicts, as positions of the data-output index (normalizer.py:126-146)
        model_output_names = list(self.data_indices.model.output.name_to_index.keys())
        data_output = self.data_indices.data.output.name_to_index
        mask = torch.zeros(len(self._output_idx), dtype=torch.bool)
        for var_name in list(data_output.keys()):
            if var_name in model_output_names:
                pos = (self._output_idx == data_output[var_name]).nonzero(as_tuple=True)[0].item()
                mask[pos] = True
        self.register_buffer("_model_output_idx", self._output_idx[mask], persistent=True)
        self._gathered: dict = {}

    def _validate_normalization_inputs(self, name_to_index: dict, minimum, maximum, mean, stdev) -> None:
        assert len(self.methods) == sum(len(v) for v in self.method_config.values()), (
            f"Error parsing methods in InputNormalizer methods ({len(self.methods)}) "
            f"and entries in config ({sum(len(v) for v in self.method_config)}) do not match."
        )
        n = minimum.size
        assert maximum.size == n, (maximum.size, n)
        assert mean.size == n, (mean.size, n)
        assert stdev.size == n, (stdev.size, n)
        assert isinstance(self.methods, dict)
        for name, method in self.methods.items():
            assert name in name_to_index, f"{name} is not a valid variable name"
            assert method in _METHODS, f"{method} is not a valid normalisation method"

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
