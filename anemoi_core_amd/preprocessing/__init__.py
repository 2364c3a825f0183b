"""Pre- / post-processors at the model edge — mirror of the reference's ``anemoi.models.preprocessing`` package for the one
processor that sits on the hot path, the input normaliser (preprocessing/__init__.py:22-206, normalizer.py:24-252).

``BasePreprocessor`` parses the reference's processor config (default / remap / method lists), ``Processors`` chains
processors (reversed when ``inverse``) exactly like the reference.  ``InputNormalizer`` keeps the reference's buffers and
state_dict keys; its arithmetic is a per-variable affine map that the model fuses into its input / output assembly kernels
(``AnemoiModelEncProcDec.predict_step``)."""
from __future__ import annotations

import logging
from typing import Optional

import torch
from torch import Tensor, nn

LOGGER = logging.getLogger(__name__)


class BasePreprocessor(nn.Module):
    """Base class for data pre- and post-processors (reference preprocessing/__init__.py:22-149)."""

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__()
        self.default, self.remap, self.normalizer, self.method_config, self.method_kwargs = self._process_config(config)
        self.methods = self._invert_key_value_list(self.method_config)
        self.data_indices = data_indices

    @classmethod
    def _process_config(cls, config):
        special = ["default", "remap", "normalizer", "method_kwargs"]  # keys that do not hold a list of variables
        default = config.get("default", "none")
        remap = config.get("remap", {})
        normalizer = config.get("normalizer", "none")
        method_kwargs = config.get("method_kwargs", {})
        method_config = {k: v for k, v in config.items() if k not in special and v is not None and v != "none"}
        if not method_config:
            LOGGER.warning("%s: Using default method %s for all variables not specified in the config.", cls.__name__, default)
        for m in method_config:
            if isinstance(method_config[m], str):
                method_config[m] = {method_config[m]: f"{m}_{method_config[m]}"}
            elif isinstance(method_config[m], list):
                method_config[m] = {method: f"{m}_{method}" for method in method_config[m]}
        return default, remap, normalizer, method_config, method_kwargs

    @staticmethod
    def _invert_key_value_list(method_config: dict) -> dict:
        return {variable: method for method, variables in method_config.items() if not isinstance(variables, str) for variable in variables}

    def forward(self, x, in_place: bool = True, inverse: bool = False, **kwargs) -> Tensor:
        if "skip_imputation" in kwargs and not getattr(self, "supports_skip_imputation", False):
            kwargs = {k: v for k, v in kwargs.items() if k != "skip_imputation"}
        if inverse:
            return self.inverse_transform(x, in_place=in_place, **kwargs)
        return self.transform(x, in_place=in_place, **kwargs)

    def transform(self, x, in_place: bool = True, **kwargs) -> Tensor:
        return x if in_place else x.clone()

    def inverse_transform(self, x, in_place: bool = True, **kwargs) -> Tensor:
        return x if in_place else x.clone()


class Processors(nn.Module):
    """A collection of processors (reference preprocessing/__init__.py:152-206): ``processors`` is a list of
    ``[name, module]`` pairs; with ``inverse`` they run in reverse order with ``inverse=True``."""

    def __init__(self, processors: list, inverse: bool = False) -> None:
        super().__init__()
        self.inverse = inverse
        self.first_run = True
        if inverse:
            processors = processors[::-1]
        self.processors = nn.ModuleDict(processors)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__} [{'inverse' if self.inverse else 'forward'}]({self.processors})"

    def forward(self, x, in_place: bool = True, **kwargs) -> Tensor:
        for processor in self.processors.values():
            x = processor(x, in_place=in_place, inverse=self.inverse, **kwargs)
        if self.first_run:
            self.first_run = False
            self._run_checks(x)
        return x

    def _run_checks(self, x):
        if not self.inverse:
            assert not torch.isnan(x).any(), f"NaNs ({torch.isnan(x).sum()}) found in processed tensor after {self.__class__.__name__}."


from .normalizer import InputNormalizer  # noqa: E402,F401
