"""load_layer_kernels / compute_mlp_hidden_dim — mirror of reference layers/utils.py:25-52, 87-142."""
from __future__ import annotations

import functools
import math
from typing import Optional

from ..utils.config import DotDict, InstantiationException, instantiate, locate
from . import kernels

# torch / reference targets that mean "the standard layer": served by the HIP kernels
_ALIASES = {
    "torch.nn.Linear": kernels.Linear,
    "torch.nn.LayerNorm": kernels.LayerNorm,
    "torch.nn.GELU": kernels.GELU,
    "anemoi.models.layers.normalization.AutocastLayerNorm": kernels.AutocastLayerNorm,
    "anemoi_core_amd.layers.normalization.AutocastLayerNorm": kernels.AutocastLayerNorm,
    "anemoi_core_amd.layers.normalization.LayerNorm": kernels.LayerNorm,
    "anemoi.models.layers.normalization.ConditionalLayerNorm": kernels.ConditionalLayerNorm,
    "anemoi_core_amd.layers.normalization.ConditionalLayerNorm": kernels.ConditionalLayerNorm,
    "anemoi_core_amd.layers.kernels.ConditionalLayerNorm": kernels.ConditionalLayerNorm,
    "anemoi_core_amd.layers.kernels.Linear": kernels.Linear,
    "anemoi_core_amd.layers.kernels.LayerNorm": kernels.LayerNorm,
    "anemoi_core_amd.layers.kernels.AutocastLayerNorm": kernels.AutocastLayerNorm,
    "anemoi_core_amd.layers.kernels.GELU": kernels.GELU,
}

DEFAULT_KERNELS = {
    "Linear": {"_target_": "torch.nn.Linear"},
    "LayerNorm": {"_target_": "torch.nn.LayerNorm"},
    "Activation": {"_target_": "torch.nn.GELU"},
    "QueryNorm": {"_target_": "anemoi.models.layers.normalization.AutocastLayerNorm", "_partial_": True, "bias": False},
    "KeyNorm": {"_target_": "anemoi.models.layers.normalization.AutocastLayerNorm", "_partial_": True, "bias": False},
}


def compute_mlp_hidden_dim(num_channels: int, mlp_hidden_ratio: float) -> int:
    """int(num_channels * ratio + 0.5), validated like the reference (layers/utils.py:25-52)."""
    if not math.isfinite(mlp_hidden_ratio):
        raise ValueError(f"`mlp_hidden_ratio` must be finite, got {mlp_hidden_ratio}.")
    if mlp_hidden_ratio <= 0:
        raise ValueError(f"`mlp_hidden_ratio` must be > 0, got {mlp_hidden_ratio}.")
    hidden_dim = int(num_channels * mlp_hidden_ratio + 0.5)
    if hidden_dim <= 0:
        raise ValueError(f"Computed hidden_dim must be > 0, got {hidden_dim}.")
    return hidden_dim


def load_layer_kernels(kernel_config: Optional[dict] = None, instance: bool = True) -> DotDict:
    """Same contract as the reference: returns factories ``Linear``, ``LayerNorm``, ``Activation``, ``QueryNorm``,
    ``KeyNorm``.  Only the standard layers are accepted (they map to the HIP kernels): the fused blocks rely on their
    semantics, and there is no eager-PyTorch fallback."""
    if kernel_config is not None and all(callable(v) for v in kernel_config.values()) and len(kernel_config) > 0:
        return DotDict(kernel_config) if not isinstance(kernel_config, DotDict) else kernel_config  # already loaded
    cfg = {**DEFAULT_KERNELS, **(dict(kernel_config) if kernel_config else {})}
    out = DotDict()
    for name, entry in cfg.items():
        if not instance:
            out[name] = entry
            continue
        entry = dict(entry)
        target = entry.pop("_target_")
        entry.pop("_partial_", None)
        if target not in _ALIASES:
            try:
                locate(target)
            except InstantiationException:
                raise
            raise NotImplementedError(
                f"layer_kernels.{name} = '{target}': only the standard Linear / LayerNorm / GELU layers are supported "
                "by the MI355X path (they are fused into hand-written kernels)."
            )
        out[name] = functools.partial(_ALIASES[target], **entry) if entry else _ALIASES[target]
    return out
