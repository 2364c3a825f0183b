"""MLP — mirror of reference layers/mlp.py:97-179 (mlp_implementation="mlp"), same state_dict keys
(``mlp.<2i>.{weight,bias}``, ``layer_norm.*``), with GELU / LayerNorm / residual fused into the kernel epilogues."""
from __future__ import annotations

from typing import Optional

from torch import Tensor, nn

from .. import ops
from .kernels import GELU


class MLP(nn.Module):
    def __init__(self, in_features: int, hidden_dim: int, out_features: int, layer_kernels, n_extra_layers: int = 0,
                 final_activation: bool = False, layer_norm: bool = True, mlp_implementation: str = "mlp") -> None:
        super().__init__()
        if n_extra_layers < 0:
            raise ValueError(f"`n_extra_layers` must be >= 0, got {n_extra_layers}.")
        if mlp_implementation != "mlp":
            raise NotImplementedError(f"mlp_implementation='{mlp_implementation}' (GLU variants) is scope row f3 (next)")
        Linear, LayerNorm = layer_kernels.Linear, layer_kernels.LayerNorm
        act = layer_kernels.Activation()
        if not isinstance(act, GELU):
            raise NotImplementedError("only GELU activations are fused")
        layers: list[nn.Module] = [Linear(in_features, hidden_dim), act]
        for _ in range(n_extra_layers):
            layers += [Linear(hidden_dim, hidden_dim), layer_kernels.Activation()]
        layers.append(Linear(hidden_dim, out_features))
        if final_activation:
            layers.append(layer_kernels.Activation())
        self.mlp = nn.Sequential(*layers)
        self.layer_norm = LayerNorm(normalized_shape=out_features) if layer_norm else None

    def forward(self, x: Tensor, *, x2: Optional[Tensor] = None, residual: Optional[Tensor] = None,
                skip_layer_norm: bool = False, skip_first: bool = False) -> Tensor:
        """y = [LayerNorm](Linear(...GELU(Linear([x | x2])))) [+ residual].

        ``x2``: second K-slab of the first layer (cat never materialised); ``residual`` is fused into the last kernel
        (LayerNorm if present, else last Linear); ``skip_first``: the caller already applied layer 0 (+GELU) — used by
        GraphConv, whose first edge layer is a gather-add GEMM; ``skip_layer_norm``: caller fuses the LayerNorm."""
        mods = list(self.mlp)
        lin_idx = [i for i, m in enumerate(mods) if isinstance(m, nn.Linear)]
        h = x.reshape(-1, x.shape[-1])
        wdt = mods[lin_idx[0]].weight.dtype
        if h.dtype != wdt:  # e.g. fp32 geometric edge attributes entering a bf16 model (what autocast does in the reference)
            h = h.to(wdt)
        ln = None if skip_layer_norm else self.layer_norm
        for n, i in enumerate(lin_idx):
            if n == 0 and skip_first:
                continue
            lin = mods[i]
            act = "gelu" if i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU) else None
            last = n == len(lin_idx) - 1
            kw = {}
            if n == 0:
                kw["x2"] = x2
            if last and ln is None and residual is not None:
                kw["residual"] = residual.reshape(-1, residual.shape[-1])
            h = ops.linear(h, lin.weight, lin.bias, act=act, **kw)
        if ln is not None:
            h = ops.layer_norm(h, ln.weight, ln.bias, ln.eps, None if residual is None else residual.reshape(-1, residual.shape[-1]))
        return h.view(*x.shape[:-1], h.shape[-1])
