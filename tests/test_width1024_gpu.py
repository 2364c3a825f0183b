"""Parity at the reference's DEFAULT GraphTransformer width: 1024 channels x 16 heads, i.e. 64 channels per head
(training/src/anemoi/training/config/model/graphtransformer.yaml:1,27-60) - the width the 512-channel benchmark model does not
exercise inside a block: K = 1024 / O = 4096 GEMMs, the LayerNorm fold over sixteen 64-column strips, the fused [q|k|v|self] buffer of
4096 columns, the wave-per-destination attention with 16 channels per lane.  One GraphTransformerProcessorBlock and one
GraphTransformerMapperBlock at the O96 hidden-mesh size (10 242 nodes) and a 2-layer AnemoiModelEncProcDec (O96 -> res 5), fp32 and bf16,
LayerNorm fold on and off, against the CPU oracle.  (The row-resident chain kernels are built for 512 channels: at 1024 the blocks
take the GEMM launches - asserted here, measured in DESIGN.md section 5.)

Tolerances as in tests/test_fullsize_parity_gpu.py (s = max(1, max |ref|)): fp32 max |err| <= 2e-5 s per block and 5e-5 s for the model;
bf16 (weights and inputs rounded first, fp32 oracle on the rounded values) max |err| <= 2e-2 s, mean |err| <= 5e-3 max(1, mean |ref|)."""
import copy

import pytest
import torch

from anemoi_core_amd.distributed.shapes import BipartiteGraphShardInfo, GraphShardInfo
from tests.helpers import lk
from tests.test_fullsize_parity_gpu import _check

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, D = 16, 1024


@pytest.fixture(scope="module")
def mesh():
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph

    return build_synthetic_graph("o8", 5)  # the hidden mesh of the benchmark (10 242 nodes, 81 840 edges) next to a small data grid


def _params(m, dtype):
    if dtype != torch.float32:
        m = m.to(dtype)  # the oracle sees the ROUNDED parameters
    return m.to(DEV), {"b." + k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_processor_block_1024(mesh, dtype, fold, monkeypatch):
    from anemoi_core_amd.layers import block as B
    from oracle import gt_oracle as O

    monkeypatch.setattr(B, "_LN_FOLD", fold)
    torch.manual_seed(1)
    blk = B.GraphTransformerProcessorBlock(in_channels=D, hidden_dim=4 * D, out_channels=D, num_heads=H, edge_dim=3, layer_kernels=lk()).eval()
    blk, params = _params(blk, dtype)
    assert not blk._chain_ok(blk.layer_norm_mlp_dst, torch.empty(mesh.num_hidden, D, device=DEV, dtype=torch.bfloat16))  # 512-channel kernels
    gen = torch.Generator().manual_seed(2)
    n = mesh.num_hidden
    x = (torch.randn(n, D, generator=gen) * 1.5 + 0.25).to(dtype).float()
    ea, ei = torch.from_numpy(mesh.proc_edge_attr).float(), torch.from_numpy(mesh.proc_edge_index)
    with torch.no_grad():
        got, _ = blk(x.to(DEV).to(dtype), ea.to(DEV), ei.to(DEV), GraphShardInfo(nodes=[n], edges=[ea.shape[0]]), 1, n)
        got2, _ = blk(x.to(DEV).to(dtype), ea.to(DEV), ei.to(DEV), GraphShardInfo(nodes=[n], edges=[ea.shape[0]]), 1, n)
        want = O.gt_processor_block(params, "b", x, ea, ei, H)
    assert torch.equal(got, got2)
    _check(f"1024-channel processor block {dtype} fold={fold}", got.float().cpu(), want, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mapper_block_1024(mesh, dtype):
    """hidden (10 242) -> data: the decoder side's bipartite block (k|v from the hidden rows, q|self from the data rows)"""
    from anemoi_core_amd.layers import block as B
    from oracle import gt_oracle as O

    torch.manual_seed(3)
    blk = B.GraphTransformerMapperBlock(in_channels=D, hidden_dim=4 * D, out_channels=D, num_heads=H, edge_dim=3, layer_kernels=lk()).eval()
    blk, params = _params(blk, dtype)
    gen = torch.Generator().manual_seed(4)
    ns, nd = mesh.num_hidden, mesh.num_data
    x_src = torch.randn(ns, D, generator=gen).to(dtype).float()
    x_dst = torch.randn(nd, D, generator=gen).to(dtype).float()
    ea, ei = torch.from_numpy(mesh.dec_edge_attr).float(), torch.from_numpy(mesh.dec_edge_index)
    with torch.no_grad():
        (_, got), _ = blk((x_src.to(DEV).to(dtype), x_dst.to(DEV).to(dtype)), ea.to(DEV), ei.to(DEV), BipartiteGraphShardInfo(), 1, (ns, nd))
        _, want = O.gt_mapper_block(params, "b", x_src, x_dst, ea, ei, H)
    _check(f"1024-channel mapper block {dtype}", got.float().cpu(), want, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_layer_model_1024(dtype):
    """bench.py's model at --channels 1024 --layers 2 (O96 -> res 5, 84 variables x 2 steps) against oracle.enc_proc_dec_forward"""
    import argparse

    import bench
    from oracle import gt_oracle as O

    args = argparse.Namespace(data_grid="o96", hidden_res=5, kind="gt", channels=D, layers=2, heads=H, vars=84)
    g, model, x = bench.build(args, DEV)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    m = copy.deepcopy(model)
    if dtype != torch.float32:
        m = m.to(dtype)
        params = {k: (v.to(dtype).float() if v.is_floating_point() else v) for k, v in params.items()}
        x = x.to(dtype).float()
    m = m.to(DEV)
    with torch.inference_mode():
        got = m({"data": x.to(DEV).to(dtype)})["data"]
        got2 = m({"data": x.to(DEV).to(dtype)})["data"]
    assert torch.equal(got, got2)
    with torch.no_grad():
        want = O.enc_proc_dec_forward({k: v.cpu() for k, v in params.items()}, dict(kind="gt", num_heads=H, num_layers=2, num_channels=D), g, x.cpu())
    assert got.shape == want.shape == (1, 1, 1, g.num_data, 84)
    _check(f"2-layer 1024-channel model {dtype}", got.float().cpu(), want, dtype, fp32_tol=5e-5)
