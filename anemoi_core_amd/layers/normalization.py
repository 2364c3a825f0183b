"""Normalisation layers - mirror of the reference's ``anemoi.models.layers.normalization`` (AutocastLayerNorm :19-31,
ConditionalLayerNorm :34-94) on the HIP kernels, plus the one call site (``apply_layer_norm``) the blocks use for whichever of
them a ``layer_kernels`` config selected."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import ops
from .kernels import LayerNorm, Linear, PaddedLinear


class AutocastLayerNorm(LayerNorm):
    """Reference layers/normalization.py:19-31: output in the input dtype — always true for the HIP kernel."""


class ConditionalLayerNorm(nn.Module):
    """Reference layers/normalization.py:34-94: ``LN(x) * (scale(cond) + 1) + bias(cond)``, same parameters and
    state_dict keys (``scale.*``, ``bias.*``; ``norm`` has none).  The two Linear maps of the conditioning run as ONE fused
    GEMM [N, 2D] whose halves feed the modulated LayerNorm kernel (forward and backward)."""

    def __init__(self, normalized_shape, condition_shape: int = 16, zero_init: bool = True, autocast: bool = True) -> None:
        super().__init__()
        D = normalized_shape if isinstance(normalized_shape, int) else int(tuple(normalized_shape)[0])
        self.norm = nn.LayerNorm(D, elementwise_affine=False)
        self.scale = Linear(condition_shape, D)
        self.bias = Linear(condition_shape, D)
        self.autocast = autocast
        self.eps = self.norm.eps
        if zero_init:
            for lin in (self.scale, self.bias):
                nn.init.zeros_(lin.weight)
                nn.init.zeros_(lin.bias)
        self._pad = PaddedLinear()

    def forward(self, x: Tensor, cond: Tensor, residual: Tensor | None = None) -> Tensor:  # noqa: D102
        D = x.shape[-1]
        w = torch.cat([self.scale.weight, self.bias.weight], 0)
        b = torch.cat([self.scale.bias, self.bias.bias], 0)
        c2 = cond.reshape(-1, cond.shape[-1]).to(w.dtype)
        pad = (-c2.shape[1]) % 8
        if pad and c2.dtype != torch.float32:  # 16-bit operand rows must be 16-byte aligned
            c2, w = torch.nn.functional.pad(c2, (0, pad)), torch.nn.functional.pad(w, (0, pad))
        mod = ops.linear(c2, w, b)  # [N, 2D] = [scale | shift]
        y = ops.cond_layer_norm(x.reshape(-1, D).to(w.dtype), mod[:, :D], mod[:, D:], self.eps).view(x.shape)
        y = y.to(x.dtype) if self.autocast else y
        return y if residual is None else y + residual


def apply_layer_norm(ln: nn.Module, x: Tensor, cond: Tensor | None = None, residual: Tensor | None = None) -> Tensor:
    """One call site for both kinds of normalisation layer a ``layer_kernels`` config can select."""
    if isinstance(ln, ConditionalLayerNorm):
        if cond is None:
            raise ValueError("ConditionalLayerNorm needs the conditioning tensor (cond=...)")
        return ln(x, cond, residual)
    return ops.layer_norm(x, ln.weight, ln.bias, ln.eps, residual)
