"""Build libanemoi_hip.so (gfx950 only) in-tree with hipcc.

``python -m anemoi_core_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
The .so is git-ignored (history stays source-only) but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIBPATH = os.path.join(LIBDIR, "libanemoi_hip.so")
EXT_PATH = os.path.join(LIBDIR, "libanemoi_torch.so")
INCLUDE = os.path.join(REPO, "include")

SOURCES = ["lib.cpp", "gt_attention.hip", "gt_attention_bwd.hip", "rowwise.hip", "rowwise_bwd.hip", "linear.hip", "wgrad.hip", "peer.hip", "gt_chain2.hip", "gt_rowchain.hip", "gt_cluster_chain.hip", "gnn_chain.hip"]
# `--experiments`: the same sources compiled with -DANEMOI_EXPERIMENTS (timing switches that change what a kernel computes, in-kernel
# timeline instantiations) plus the measured-and-superseded kernels under csrc/experiments/ -> lib/libanemoi_hip_exp.so, selected with
# ANEMOI_HIP_LIB=<that path> by the A/B scripts in tools/.  Never the default: the product library carries none of it.
EXPERIMENT_SOURCES = ["experiments/gt_chain.hip", "experiments/gnn_chain2.hip"]
EXP_LIBPATH = os.path.join(LIBDIR, "libanemoi_hip_exp.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _deps(path: str, dirs: list[str], seen: set | None = None) -> list[str]:
    """The source and every project header it includes (recursively): a header edit rebuilds only the translation units that see it."""
    import re

    seen = set() if seen is None else seen
    if path in seen:
        return []
    seen.add(path)
    out = [path]
    for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), flags=re.M):
        for d in [os.path.dirname(path), *dirs]:
            cand = os.path.join(d, inc)
            if os.path.exists(cand):
                out += _deps(cand, dirs, seen)
                break
    return out


def build_library(force: bool = False, verbose: bool = True, extra_flags: tuple[str, ...] = (), experiments: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj_exp" if experiments else "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "chain_core.h"), os.path.join(CSRC, "chain2_core.h"), os.path.join(CSRC, "gnn_chain_args.h"), os.path.join(INCLUDE, "anemoi_hip.h")]
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", INCLUDE, "-I", CSRC, "-Wall", "-Wno-unused-function", *extra_flags]
    sources, libpath = SOURCES, LIBPATH
    inc_dirs = [CSRC, INCLUDE] + ([os.path.join(CSRC, "experiments")] if experiments else [])
    if experiments:
        flags += ["-DANEMOI_EXPERIMENTS", "-I", os.path.join(CSRC, "experiments")]
        headers.append(os.path.join(CSRC, "experiments", "anemoi_hip_experiments.h"))
        sources, libpath = SOURCES + EXPERIMENT_SOURCES, EXP_LIBPATH

    def compile_one(src: str) -> str:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        if not force and _newer(obj, _deps(path, inc_dirs)):
            return obj
        cmd = [hipcc, *flags, "-x", "hip", "-c", path, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(sources))) as ex:
        objs = list(ex.map(compile_one, sources))
    if force or not _newer(libpath, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", libpath, *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if not experiments:
        build_torch_extension(hipcc, force=force, verbose=verbose)
    return libpath


def build_torch_extension(hipcc: str, force: bool = False, verbose: bool = True) -> str:
    """lib/libanemoi_torch.so: the TORCH_LIBRARY layer over the C ABI (csrc/torch_binding.cpp; host code only, links against
    libanemoi_hip.so through $ORIGIN and against the libtorch of the running interpreter)."""
    import torch

    src = os.path.join(CSRC, "torch_binding.cpp")
    # the extension links against the libtorch of the interpreter that built it: a stamp with torch's version next to the .so makes
    # a torch upgrade a rebuild instead of an undefined-symbol error at load time
    stamp = EXT_PATH + ".torch_version"
    stamped = open(stamp).read().strip() if os.path.exists(stamp) else None
    # (a .so without a stamp is one built before the stamp existed, by this image's torch: adopt it)
    if not force and stamped in (None, torch.__version__) and _newer(EXT_PATH, [src, os.path.join(INCLUDE, "anemoi_hip.h"), LIBPATH]):
        if stamped is None:
            open(stamp, "w").write(torch.__version__)
        return EXT_PATH
    troot = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-I", INCLUDE, "-I", os.path.join(troot, "include"),
           "-I", os.path.join(troot, "include", "torch", "csrc", "api", "include"), "-I", os.path.join(rocm, "include"), src, "-o", EXT_PATH,
           "-L", LIBDIR, "-lanemoi_hip", "-L", os.path.join(troot, "lib"), "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    open(stamp, "w").write(torch.__version__)
    return EXT_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
