"""MLP — mirror of reference layers/mlp.py:25-179, same state_dict keys (``mlp.<2i>.{weight,bias}`` for "mlp",
``mlp.<i>.{gate_proj,value_proj}.{weight,bias}`` for the gated variants, ``layer_norm.*``), with GELU / LayerNorm / residual
fused into the kernel epilogues; a gated layer is ONE fused [gate | value] projection followed by the gating kernel."""
from __future__ import annotations

from typing import Optional

from torch import Tensor, nn

import torch

import os

from .. import ops
from .kernels import GELU, PaddedLinear

# the embedding MLPs of the GNN mappers / processor as one row-resident launch each (needs ANEMOI_GNN_CHAIN too); 0: launch per GEMM
_EMB_CHAIN = os.environ.get("ANEMOI_GNN_EMB_CHAIN", "1") == "1"


class GatedMLPLayer(nn.Module):
    """gating(gate_proj(x)) * value_proj(x) (reference mlp.py:38-53)."""

    def __init__(self, in_features: int, out_features: int, layer_kernels, mlp_implementation: str) -> None:
        super().__init__()
        if mlp_implementation not in ops.GLU_KINDS:
            raise ValueError(f"`mlp_implementation` must be one of {tuple(ops.GLU_KINDS)}, got '{mlp_implementation}'.")
        self.gate_proj = layer_kernels.Linear(in_features, out_features)
        self.value_proj = layer_kernels.Linear(in_features, out_features)
        self.kind = mlp_implementation

    def fused_weights(self):
        return torch.cat([self.gate_proj.weight, self.value_proj.weight], 0), torch.cat([self.gate_proj.bias, self.value_proj.bias], 0)

    def forward(self, x: Tensor, x2: Optional[Tensor] = None) -> Tensor:
        w, b = self.fused_weights()
        h = x.reshape(-1, x.shape[-1])
        return ops.glu(ops.linear(h, w, b, x2=x2), self.kind).view(*x.shape[:-1], -1)


class MLP(nn.Module):
    def __init__(self, in_features: int, hidden_dim: int, out_features: int, layer_kernels, n_extra_layers: int = 0,
                 final_activation: bool = False, layer_norm: bool = True, mlp_implementation: str = "mlp") -> None:
        super().__init__()
        if n_extra_layers < 0:
            raise ValueError(f"`n_extra_layers` must be >= 0, got {n_extra_layers}.")
        Linear, LayerNorm = layer_kernels.Linear, layer_kernels.LayerNorm
        self.mlp_implementation = mlp_implementation
        if mlp_implementation == "mlp":
            act = layer_kernels.Activation()
            if not isinstance(act, GELU):
                raise NotImplementedError("only GELU activations are fused")
            layers: list[nn.Module] = [Linear(in_features, hidden_dim), act]
            for _ in range(n_extra_layers):
                layers += [Linear(hidden_dim, hidden_dim), layer_kernels.Activation()]
            layers.append(Linear(hidden_dim, out_features))
            if final_activation:
                layers.append(layer_kernels.Activation())
        else:  # gated variants: layer_kernels.Activation is ignored (mlp.py:88-94)
            if final_activation:
                raise NotImplementedError("final_activation with a gated mlp_implementation is not used on the hot path")
            layers = [GatedMLPLayer(in_features, hidden_dim, layer_kernels, mlp_implementation)]
            for _ in range(n_extra_layers):
                layers.append(GatedMLPLayer(hidden_dim, hidden_dim, layer_kernels, mlp_implementation))
            layers.append(Linear(hidden_dim, out_features))
        self.mlp = nn.Sequential(*layers)
        self._pad_first = PaddedLinear()
        self.layer_norm = LayerNorm(normalized_shape=out_features) if layer_norm else None

    def forward(self, x: Tensor, *, x2: Optional[Tensor] = None, residual: Optional[Tensor] = None,
                skip_layer_norm: bool = False, skip_first: bool = False) -> Tensor:
        """y = [LayerNorm](Linear(...GELU(Linear([x | x2])))) [+ residual].

        ``x2``: second K-slab of the first layer (cat never materialised); ``residual`` is fused into the last kernel
        (LayerNorm if present, else last Linear); ``skip_first``: the caller already applied layer 0 (+GELU) — used by
        GraphConv, whose first edge layer is a gather-add GEMM; ``skip_layer_norm``: caller fuses the LayerNorm."""
        mods = list(self.mlp)
        if self.mlp_implementation != "mlp":
            return self._forward_gated(x, mods, x2, residual, skip_layer_norm, skip_first)
        lin_idx = [i for i, m in enumerate(mods) if isinstance(m, nn.Linear)]
        h = x.reshape(-1, x.shape[-1])
        if x2 is None and not skip_first and not skip_layer_norm and self._embedding_chain_ok(h, mods):
            res = None if residual is None else residual.reshape(-1, residual.shape[-1])
            return self._embedding_chain(h, mods, res).view(*x.shape[:-1], -1)
        wdt = mods[lin_idx[0]].weight.dtype
        pad_first = (not skip_first and x2 is None and mods[lin_idx[0]].weight.shape[1] % 8 and wdt != torch.float32
                     and h.dtype in (wdt, torch.float32))  # the padded first layer casts while it pads (one kernel)
        if h.dtype != wdt and not pad_first:  # e.g. fp32 geometric edge attributes entering a bf16 model (what autocast does in the reference)
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
            if n == 0 and pad_first:
                # e.g. the 11 raw edge attributes entering a GNN's edge embedding: zero-pad K onto the MFMA path
                h = self._pad_first(h, lin, act=act, **{k: v for k, v in kw.items() if k != "x2"})
            else:
                h = ops.linear(h, lin.weight, lin.bias, act=act, **kw)
        if ln is not None:
            h = ops.layer_norm(h, ln.weight, ln.bias, ln.eps, None if residual is None else residual.reshape(-1, residual.shape[-1]))
        return h.view(*x.shape[:-1], h.shape[-1])

    def _embedding_chain_ok(self, h: Tensor, mods: list) -> bool:
        """Linear -> GELU -> Linear -> GELU -> Linear -> LayerNorm into 512 channels of a 16-bit model, inference: the shape of the GNN
        mappers' / processor's embedding MLPs, which csrc/gnn_chain.hip runs as one row-resident launch (ops.gnn_mlp_chain)."""
        from . import conv  # (the switch ANEMOI_GNN_CHAIN lives with the other GraphConv chains)

        D = ops.CHAIN_CHANNELS
        if not (_EMB_CHAIN and conv._GNN_CHAIN and h.is_cuda and len(mods) == 5 and self.layer_norm is not None and h.shape[0] >= 1024):
            return False
        l0, l1, l2 = mods[0], mods[2], mods[4]
        ln = self.layer_norm
        return (isinstance(l0, nn.Linear) and isinstance(mods[1], GELU) and isinstance(l1, nn.Linear) and isinstance(mods[3], GELU)
                and isinstance(l2, nn.Linear) and l0.weight.dtype in (torch.bfloat16, torch.float16) and h.dtype in (l0.weight.dtype, torch.float32)
                and l0.weight.shape[0] == D and l0.weight.shape[1] <= D and l1.weight.shape == (D, D) and l2.weight.shape == (D, D)
                and all(m.bias is not None for m in (l0, l1, l2))
                and type(ln).__name__ in ("LayerNorm", "AutocastLayerNorm") and getattr(ln, "weight", None) is not None
                and not (torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters()))))

    def _embedding_chain(self, h: Tensor, mods: list, residual: Optional[Tensor]) -> Tensor:
        from .conv import _derived

        l0, l1, l2, ln = mods[0], mods[2], mods[4], self.layer_norm
        K, wdt = l0.weight.shape[1], l0.weight.dtype
        Kp = -(-K // 128) * 128  # the first GEMM walks K in groups of 128 columns: zero columns in x AND in the packed weight
        P = ops.pack_weight_frag
        w0 = _derived(self, "e0", [l0.weight], lambda: P(torch.nn.functional.pad(l0.weight, (0, Kp - K))))
        w1 = _derived(self, "e1", [l1.weight], lambda: P(l1.weight))
        w2 = _derived(self, "e2", [l2.weight], lambda: P(l2.weight))
        if K != Kp or h.dtype != wdt or h.stride(1) != 1 or h.stride(0) % 8 or h.data_ptr() % 16:
            if h.stride(1) != 1:
                h = h.contiguous()
            h = ops.assemble_input(h.unsqueeze(0), None, Kp, out_dtype=wdt)  # cast + zero columns, one kernel
        if residual is not None and (residual.dtype != wdt or residual.stride(1) != 1):
            residual = residual.to(wdt).contiguous()
        return ops.gnn_mlp_chain(h, w0, l0.bias, w1, l1.bias, w2, l2.bias, ln.weight, ln.bias, ln.eps, residual)

    def _forward_gated(self, x, mods, x2, residual, skip_layer_norm, skip_first):
        h = x.reshape(-1, x.shape[-1])
        wdt = mods[0].gate_proj.weight.dtype
        if h.dtype != wdt:
            h = h.to(wdt)
        ln = None if skip_layer_norm else self.layer_norm
        for n, m in enumerate(mods[:-1]):
            if n == 0 and skip_first:  # GraphConv applied the first gated layer itself (gather-add GEMM + gating kernel)
                continue
            h = m(h, x2=x2 if n == 0 else None)
        last = mods[-1]
        kw = {}
        if ln is None and residual is not None:
            kw["residual"] = residual.reshape(-1, residual.shape[-1])
        h = ops.linear(h, last.weight, last.bias, **kw)
        if ln is not None:
            h = ops.layer_norm(h, ln.weight, ln.bias, ln.eps, None if residual is None else residual.reshape(-1, residual.shape[-1]))
        return h.view(*x.shape[:-1], h.shape[-1])
