"""Minimal config helpers so the modules can be built from the reference's YAML-style dicts without Hydra.

``DotDict`` mirrors ``anemoi.utils.config.DotDict`` (attribute access on nested dicts); ``instantiate`` resolves a
``_target_`` dotted path with ``_partial_`` / ``_recursive_`` semantics as used by the reference
(models/src/anemoi/models/layers/utils.py:132, models/encoder_processor_decoder.py:51-96).  When Hydra is
installed the real ``hydra.utils.instantiate`` can be used instead; both accept the same dicts.
"""
from __future__ import annotations

import functools
import importlib
from typing import Any


class DotDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            super().__setitem__(k, self._wrap(v))

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, DotDict):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(i) for i in v]
        return v

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def __setitem__(self, key, value):
        super().__setitem__(key, self._wrap(value))


class InstantiationException(Exception):
    pass


def locate(path: str) -> Any:
    module, _, name = path.rpartition(".")
    try:
        return getattr(importlib.import_module(module), name)
    except (ImportError, AttributeError, ValueError) as e:
        raise InstantiationException(f"cannot locate '{path}': {e}") from e


def instantiate(config, *args, **kwargs):
    cfg = dict(config)
    kwargs = dict(kwargs)
    recursive = kwargs.pop("_recursive_", cfg.pop("_recursive_", True))
    partial = kwargs.pop("_partial_", cfg.pop("_partial_", False))
    cfg.pop("_convert_", None)
    kwargs.pop("_convert_", None)
    target = cfg.pop("_target_")
    obj = locate(target) if isinstance(target, str) else target
    merged = {**cfg, **kwargs}
    if recursive:
        merged = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in merged.items()}
    if partial:
        return functools.partial(obj, *args, **merged)
    return obj(*args, **merged)
