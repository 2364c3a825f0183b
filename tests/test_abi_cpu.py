"""CPU checks of the C-ABI boundary: the library builds/loads and exports every symbol include/anemoi_hip.h declares
(no compute calls: there is no GPU here), and the ctypes signatures agree with the header."""
import os
import re

import pytest
import torch

from anemoi_core_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "anemoi_hip.h")


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"^(?:int|int64_t|const char\*)\s+(anemoi_\w+)\s*\((.*?)\);", text, flags=re.S | re.M):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args == "void" else len([a for a in args.split(",") if a.strip()])
    return out


def test_header_and_binding_agree():
    decl = header_functions()
    assert len(decl) >= 9
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, nargs in decl.items():
        assert len(_lib.SIGNATURES[name][0]) == nargs, name


def test_library_loads_and_exports_all_symbols():
    if not os.path.exists(_lib.LIB_PATH):
        from anemoi_core_amd.build import build_library

        build_library(verbose=False)
    lib = _lib.load()
    assert lib.anemoi_hip_abi_version() == _lib.ABI_VERSION
    for name in header_functions():
        assert hasattr(lib, name), name


def test_product_library_carries_no_experiment_switches():
    """switches that make a kernel skip work (timing experiments: ANEMOI_*_DBG, wave-priority / delay knobs, the
    two-group GraphConv chain, the round-4 layer chain, in-kernel timelines) are compiled only into the experiments build
    (-DANEMOI_EXPERIMENTS -> lib/libanemoi_hip_exp.so).  The product .so must not even contain their names, nor the experimental entry
    points: no environment variable can change what a shipped kernel computes."""
    if not os.path.exists(_lib.LIB_PATH):
        from anemoi_core_amd.build import build_library

        build_library(verbose=False)
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"_DBG", b"ANEMOI_CHAIN2_PRIO", b"ANEMOI_CHAIN2_B_DELAY", b"ANEMOI_CHAIN2_WARM", b"ANEMOI_GNN_CHAIN_V2", b"ANEMOI_CHAIN_PRIO_YOUNG",
                 b"anemoi_gt_chain_fwd", b"anemoi_gnn_edge_chain_timeline", b"gnn_edge_chain2_kernel", b"gt_chain_kernel"):
        assert name not in blob, name
    lib = _lib.load()
    for name in _lib.EXPERIMENT_SIGNATURES:
        assert not hasattr(lib, name), name


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("ANEMOI_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError, match="no CPU / eager fallback"):
        _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)


def test_cpu_tensors_are_rejected():
    from anemoi_core_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8))


def test_build_csc_host_logic():
    from anemoi_core_amd import ops

    ei = torch.tensor([[3, 1, 2, 0, 4], [2, 0, 2, 1, 0]])
    csc = ops.build_csc(ei, (5, 4), edges_are_dst_sorted=False)
    assert csc.colptr.tolist() == [0, 2, 3, 5, 5]
    assert csc.row.tolist() == [1, 4, 0, 3, 2]  # stable within a destination
    assert csc.dst.tolist() == [0, 0, 1, 2, 2]
    assert csc.perm.tolist() == [1, 4, 3, 0, 2]
    with pytest.raises(ValueError):
        ops.build_csc(ei, (5, 4), edges_are_dst_sorted=True, check=True)
    empty = ops.build_csc(torch.zeros(2, 0, dtype=torch.long), (3, 2))
    assert empty.colptr.tolist() == [0, 0, 0] and empty.num_edges == 0


def test_processing_order_is_attached_to_the_cached_csc_of_a_large_square_graph():
    """Host logic of the fused attention's work order (ops.processing_order through layers/graphcache.get_csc): a permutation
    that keeps every XCD's eighth of the index range, attached to the (frozen) CSC of the res-6 mesh; small and bipartite graphs
    keep the identity."""
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.layers.graphcache import get_csc

    g = build_synthetic_graph("o8", 6)
    n = g.num_hidden
    csc = get_csc(torch.from_numpy(g.proc_edge_index), (n, n), True)
    assert csc.order is not None and csc.order.dtype == torch.int32
    assert torch.equal(torch.sort(csc.order.long())[0], torch.arange(n))
    per = (n + 7) // 8
    assert torch.equal(csc.order.long() // per, torch.arange(n) // per)
    assert get_csc(torch.from_numpy(g.enc_edge_index), (g.num_data, n), True).order is None
    g3 = build_synthetic_graph("o8", 3)
    assert get_csc(torch.from_numpy(g3.proc_edge_index), (g3.num_hidden, g3.num_hidden), True).order is None


def test_torch_extension_loads_and_registers_its_ops():
    """lib/libanemoi_torch.so (csrc/torch_binding.cpp) is built with the library, loads into this interpreter and registers the
    five hot forward ops; CPU tensors are refused as loudly as on the ctypes path."""
    from anemoi_core_amd import _ext
    from anemoi_core_amd.build import build_library

    build_library(verbose=False)
    ext = _ext.ops()
    assert ext is not None
    for name in ("linear", "linear_out", "layer_norm", "layer_norm_out", "gt_attention_fused_edge", "linear_with_row_stats", "linear_ln_folded"):
        assert hasattr(ext, name), name
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.layer_norm(torch.randn(4, 64), torch.ones(64), None, 1e-5, None)


def test_torch_library_ops_have_fake_kernels():
    """torch.ops.anemoi_hip.* carry Meta / FakeTensor kernels (anemoi_core_amd/_ext.py: _register_fakes), so the ops are opaque but
    traceable leaves - the reference registers the same for its Triton op (triton/gt.py:431-447, 553-556).  Shapes / dtypes must be
    those csrc/torch_binding.cpp produces (checked against the real kernels in tests/test_torch_ext_gpu.py)."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    from anemoi_core_amd import _ext

    o = _ext.ops()
    assert o is not None
    bf = torch.bfloat16
    with FakeTensorMode():
        x, w = torch.empty(100, 512, dtype=bf, device="cuda"), torch.empty(2048, 512, dtype=bf, device="cuda")
        y = o.linear(x, w, None, 1, None, None, None, None, None, None)
        assert y.shape == (100, 2048) and y.dtype == bf and y.device.type == "cuda"
        assert o.layer_norm(x, torch.empty(512, dtype=bf, device="cuda"), None, 1e-5, None).shape == (100, 512)
        y2, st = o.linear_with_row_stats(y, torch.empty(512, 2048, dtype=bf, device="cuda"), None, x)
        assert y2.shape == (100, 512) and st.shape == (100, 8, 2) and st.dtype == torch.float32
        ne = o.linear_with_row_stats(x.float(), w.float(), None, None)  # fp32: the 'not eligible' sentinel, as the kernel side
        assert ne[0].dim() == 1 and ne[0].numel() == 0
        c = torch.empty(2048, device="cuda")
        assert o.linear_ln_folded(y2, w, c, c, st, 1e-5, 0).shape == (100, 2048)
        q = torch.empty(10, 512, dtype=bf, device="cuda")
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device="cuda")  # noqa: E731
        out, lse = o.gt_attention_fused_edge(q, q, q, torch.empty(30, 12, device="cuda"), torch.empty(512, 12, device="cuda"), i32(30), i32(11), None,
                                             10, 16, None, True)
        assert out.shape == (10, 512) and out.dtype == bf and lse.shape == (10, 16) and lse.dtype == torch.float32
