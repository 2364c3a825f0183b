"""The TORCH_LIBRARY layer over the C ABI (csrc/torch_binding.cpp -> lib/libanemoi_torch.so): the hot forward ops as
``torch.ops.anemoi_hip.*``, taking tensors and the current HIP stream in C++ instead of ~10 us of ctypes marshalling per launch.

Same kernels, same C ABI underneath, bit-identical results (tests/test_torch_ext_gpu.py).  ``ANEMOI_TORCH_EXT=0`` keeps every
call on the ctypes binding (for A/B runs).  Built by ``python -m anemoi_core_amd.build`` together with libanemoi_hip.so; a
missing library is an error, like a missing libanemoi_hip.so - there is no silent fallback."""
from __future__ import annotations

import os
import threading

import torch

from . import _lib

EXT_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libanemoi_torch.so")
# an alternative build of the kernel library (ANEMOI_HIP_LIB, same-box A/Bs) is only reachable through ctypes: the extension is
# linked against THE library next to it
ENABLED = os.environ.get("ANEMOI_TORCH_EXT", "1") != "0" and not os.environ.get("ANEMOI_HIP_LIB")

_lock = threading.Lock()
_ops = None


def ops():
    """``torch.ops.anemoi_hip`` (loading the library on first use), or None when switched off."""
    global _ops
    if not ENABLED:
        return None
    if _ops is not None:
        return _ops
    with _lock:
        if _ops is None:
            _lib.load()  # libanemoi_hip.so first: the extension links against it
            if not os.path.exists(EXT_PATH):
                raise _lib.HipLibraryError(f"{EXT_PATH} not found: build it with `python -m anemoi_core_amd.build` "
                                           "(or set ANEMOI_TORCH_EXT=0 to stay on the ctypes binding). anemoi_core_amd has no CPU / eager fallback.")
            torch.ops.load_library(EXT_PATH)
            _ops = torch.ops.anemoi_hip
    return _ops
