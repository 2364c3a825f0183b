"""``layer_kernels`` building blocks backed by the HIP library.

The reference lets a config swap the Linear / LayerNorm / Activation classes through ``layer_kernels`` ``_target_``
strings (models/src/anemoi/models/layers/utils.py:87-142).  These classes are those plug-ins for MI355X: they subclass
the torch modules (identical parameters, initialisation and state_dict keys) and override ``forward`` with one fused
kernel launch.  They also work stand-alone inside the unmodified reference blocks.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import ops
from ..utils.tensors import version


class Linear(nn.Linear):
    """torch.nn.Linear parameters, MFMA forward (bias fused)."""

    def forward(self, x: Tensor) -> Tensor:  # noqa: D102
        y = ops.linear(x.reshape(-1, x.shape[-1]), self.weight, self.bias)
        return y.view(*x.shape[:-1], self.out_features)


class LayerNorm(nn.LayerNorm):
    """torch.nn.LayerNorm parameters (1-D normalized_shape), one-pass wave-per-row forward with fp32 statistics."""

    def __init__(self, normalized_shape, eps: float = 1e-5, elementwise_affine: bool = True, bias: bool = True, **kw):
        super().__init__(normalized_shape, eps=eps, elementwise_affine=elementwise_affine, bias=bias, **kw)
        if len(self.normalized_shape) != 1 or not elementwise_affine:
            raise NotImplementedError("only affine LayerNorm over the last dimension is supported")

    def forward(self, x: Tensor, residual: Tensor | None = None) -> Tensor:  # noqa: D102
        return ops.layer_norm(x, self.weight, self.bias, self.eps, residual)


class GELU(nn.GELU):
    """Exact (erf) GELU.  This package's MLP / blocks fuse it into the preceding Linear's epilogue (no launch of its own);
    called as a module - e.g. as ``layer_kernels.Activation`` inside the reference's own MLP (layers/mlp.py:158-169) or a
    plain ``nn.Sequential`` - it is one row-wise HIP kernel (``anemoi_gelu_fwd`` / ``anemoi_gelu_bwd``)."""

    def __init__(self, approximate: str = "none") -> None:
        if approximate != "none":
            raise NotImplementedError("only the exact (erf) GELU is implemented (torch.nn.GELU default, layers/utils.py:111)")
        super().__init__(approximate="none")

    def forward(self, x: Tensor) -> Tensor:  # noqa: D102
        return ops.gelu(x)


class PaddedLinear:
    """Apply an ``nn.Linear`` whose in_features is not a multiple of 8 on the MFMA path: 16-bit operand rows must be
    16-byte aligned, so the K dimension of both the input and a cached copy of the weight is zero-padded to a multiple
    of 8 (exact: the extra products are 0).  An input that already carries zero columns (any multiple of 8 >= in_features)
    is taken as it is and only the weight is padded to its width.  fp32 inputs take the generic kernel and need no padding."""

    def __init__(self):
        self._sig = None
        self._w = None

    def _prep(self, x: Tensor, lin: nn.Linear, wide: bool = False):
        """(x, weight) with the K dimension zero-padded as the MFMA kernels need it: a multiple of 8 (16-byte rows), or - for
        tall inputs (or ``wide``) whose K is small or within 16 columns of it - of 64, so that the GEMM takes the DMA-ring
        kernels (an 81 840-row edge embedding with K = 3 is bound by its 84 MB of output either way; the register-staged
        kernel needs 44-90 us for it, the ring kernels ~20)."""
        K, W = lin.weight.shape[1], x.shape[-1]
        wdt = lin.weight.dtype
        if W == K:
            to64 = (wide or x.shape[0] >= 4096) and K % 64 and (K < 64 or (-K) % 64 <= 16)
            pad = (-K) % 64 if to64 else (-K) % 8
            if pad:
                if (x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.dtype in (wdt, torch.float32)
                        and not (torch.is_grad_enabled() and x.requires_grad)):
                    # cast (fp32 geometric attributes entering a 16-bit model) + zero columns in ONE kernel (was cast, fill, copy)
                    x = ops.assemble_input(x.unsqueeze(0), None, K + pad, out_dtype=wdt)
                else:
                    x = torch.nn.functional.pad(x.to(wdt), (0, pad))
                W = x.shape[-1]
        elif W < K or W % 8:  # else: the caller already appended zero columns (one padded copy shared by several consumers,
            # possibly up to a multiple of 64 so that the GEMM takes the DMA-ring kernels)
            raise ValueError(f"input width {W} is neither in_features {K} nor a zero-padded width (multiple of 8 >= {K})")
        if x.dtype != wdt:
            x = x.to(wdt)
        if W == K:
            return x, lin.weight
        if torch.is_grad_enabled() and lin.weight.requires_grad:  # training: gradients flow through the padding
            return x, torch.nn.functional.pad(lin.weight, (0, W - K))
        sig = (lin.weight.data_ptr(), version(lin.weight), lin.weight.dtype, str(lin.weight.device), W)
        if self._sig != sig:
            with torch.no_grad():
                self._w = torch.nn.functional.pad(lin.weight, (0, W - K)).contiguous()
            self._sig = sig
        return x, self._w

    def with_row_stats(self, x: Tensor, lin: nn.Linear):
        """(y, stats) as ``ops.linear_with_row_stats`` (the LayerNorm that follows is folded into ITS consumer), or (y, None)
        when the shape does not take that path.  Inference only."""
        if x.dtype != torch.float32 and not ops._needs_grad(x, lin.weight, lin.bias):
            xp, w = self._prep(x, lin, wide=True)
            if xp.shape[-1] % 64 == 0:
                r = ops.linear_with_row_stats(xp, w, lin.bias)
                if r is not None:
                    return r
            return ops.linear(xp, w, lin.bias), None
        return self(x, lin), None

    def __call__(self, x: Tensor, lin: nn.Linear, **kw) -> Tensor:
        if lin.weight.dtype == torch.float32:
            return ops.linear(x.to(torch.float32), lin.weight, lin.bias, **kw)
        x, w = self._prep(x, lin)
        return ops.linear(x, w, lin.bias, **kw)


def __getattr__(name: str):
    """``AutocastLayerNorm`` / ``ConditionalLayerNorm`` / ``apply_layer_norm`` live where the reference keeps them
    (layers/normalization.py, which builds on the classes above); still reachable under their former names here."""
    if name in ("AutocastLayerNorm", "ConditionalLayerNorm", "apply_layer_norm"):
        from . import normalization

        return getattr(normalization, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
