"""Pre- / post-processors at the model edge (scope row f4) for the one processor on the hot path, the input normaliser.

Written from the documented behaviour of the reference's ``anemoi.models.preprocessing`` package, not from its text:

* a processor is configured by a mapping whose reserved entries are ``default`` (method for unlisted variables, "none"
  when absent), ``remap`` ({variable: variable whose statistics it borrows}), ``normalizer`` and ``method_kwargs``; every
  other entry reads ``<method>: <variable> | [<variables>]``; entries that are ``None`` or the string "none" do not count
  (reference preprocessing/__init__.py:63-88 states the same schema);
* a processor is called as ``p(x, in_place=True, inverse=False, **kw)`` and is the identity unless a subclass says otherwise;
* ``Processors([[name, module], ...], inverse=False)`` applies its members first to last, or last to first with
  ``inverse=True``; the first tensor that leaves a forward (non-inverse) chain must be free of NaNs.

State_dict keys (``processors.<name>.<buffer>``) are those of the reference, so its checkpoints load strictly.  On the HIP
path ``AnemoiModelEncProcDec.predict_step`` does not call these modules at all when the chain is a lone ``InputNormalizer``:
the affine map becomes a column program of the assembly kernels; ``Processors.check_first_batch`` keeps the NaN check there."""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Mapping, Optional

import torch
from torch import Tensor, nn

LOGGER = logging.getLogger(__name__)

_RESERVED = ("default", "remap", "normalizer", "method_kwargs")


@dataclass(frozen=True)
class ProcessorSpec:
    """A processor's configuration in normal form: ``variables[method]`` is the tuple of variable names a method was
    listed for, ``method_of[variable]`` the inverse lookup."""

    default: str = "none"
    remap: Mapping = field(default_factory=dict)
    normalizer: str = "none"
    method_kwargs: Mapping = field(default_factory=dict)
    variables: Mapping = field(default_factory=dict)

    @property
    def method_of(self) -> dict:
        out: dict = {}
        for method, names in self.variables.items():
            out.update(dict.fromkeys(names, method))
        return out

    @staticmethod
    def parse(config: Optional[Mapping], owner: str = "processor") -> "ProcessorSpec":
        config = {} if config is None else config
        listed = {}
        for key in config.keys():
            value = config[key]
            if key in _RESERVED or value is None or (isinstance(value, str) and value == "none"):
                continue
            listed[str(key)] = (value,) if isinstance(value, str) else tuple(value)
        if not listed:
            LOGGER.warning("%s: no variable is listed under any method; '%s' applies to all of them.", owner, config.get("default", "none"))
        reserved = {k: config.get(k) for k in _RESERVED if config.get(k) is not None}
        return ProcessorSpec(variables=listed, **reserved)


class BasePreprocessor(nn.Module):
    """Base of the data pre- / post-processors.  Subclasses override ``transform`` / ``inverse_transform``; they read
    ``self.methods`` (variable -> method), ``self.default`` and ``self.remap`` like the reference's subclasses do."""

    supports_skip_imputation = False

    def __init__(self, config=None, data_indices=None, statistics: Optional[dict] = None) -> None:
        super().__init__()
        self.spec = ProcessorSpec.parse(config, type(self).__name__)
        self.data_indices = data_indices

    # the reference's attribute names, derived from the one normal form
    default = property(lambda self: self.spec.default)
    remap = property(lambda self: self.spec.remap)
    normalizer = property(lambda self: self.spec.normalizer)
    method_kwargs = property(lambda self: self.spec.method_kwargs)
    methods = property(lambda self: self.spec.method_of)

    @property
    def method_config(self) -> dict:
        return {m: {v: f"{m}_{v}" for v in names} for m, names in self.spec.variables.items()}

    def forward(self, x, in_place: bool = True, inverse: bool = False, **kwargs) -> Tensor:
        if not self.supports_skip_imputation:
            kwargs.pop("skip_imputation", None)
        step = self.inverse_transform if inverse else self.transform
        return step(x, in_place=in_place, **kwargs)

    def transform(self, x, in_place: bool = True, **kwargs) -> Tensor:
        return x if in_place else x.clone()

    inverse_transform = transform


class Processors(nn.Module):
    """An ordered chain of named processors; ``inverse=True`` runs the members last to first as inverse transforms."""

    def __init__(self, processors: list, inverse: bool = False) -> None:
        super().__init__()
        self.inverse = bool(inverse)
        self.processors = nn.ModuleDict()
        for name, module in processors:
            self.processors[name] = module
        self._awaiting_first_batch = True

    def chain(self) -> list:
        """The members in the order they are applied."""
        members = list(self.processors.values())
        return members[::-1] if self.inverse else members

    def extra_repr(self) -> str:
        return "direction=" + ("inverse" if self.inverse else "forward")

    def forward(self, x, in_place: bool = True, **kwargs) -> Tensor:
        for member in self.chain():
            x = member(x, in_place=in_place, inverse=self.inverse, **kwargs)
        self.check_first_batch(x)
        return x

    def check_first_batch(self, x: Tensor) -> None:
        """One-off sanity check of the first processed batch (forward direction only): NaNs that survive the
        pre-processors would poison the model silently.  Also called by the fused model edge, which bypasses ``forward``."""
        if not self._awaiting_first_batch:
            return
        self._awaiting_first_batch = False
        if self.inverse:
            return
        bad = int(torch.isnan(x).sum())
        assert bad == 0, f"{type(self).__name__}: {bad} NaNs in the first pre-processed batch."


from .normalizer import InputNormalizer  # noqa: E402,F401
