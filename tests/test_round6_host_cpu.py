"""Host logic of the round-6 launches (no GPU): the folds and weight images their callers prepare, the gates, the loud failures on CPU tensors."""
import pytest
import torch
import torch.nn.functional as F


def test_fold_layer_norm_identity():
    """LN(x; gamma, beta) W^T + b == ((x - mean) rstd) (W diag gamma)^T + (W beta + b): what lets the chain kernels apply LayerNorm without its affine part"""
    from anemoi_core_amd import ops

    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(7, 32, generator=g), torch.randn(12, 32, generator=g), torch.randn(12, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(32, generator=g), 0.2 * torch.randn(32, generator=g)
    wg, d = ops.fold_layer_norm(w, b, gamma, beta)
    want = F.linear(F.layer_norm(x, (32,), gamma, beta, 1e-5), w, b)
    got = F.linear(F.layer_norm(x, (32,), None, None, 1e-5), wg, d)
    assert torch.allclose(got, want, atol=1e-5)
    wg0, d0 = ops.fold_layer_norm(w, None, gamma, None)  # no biases at all
    assert torch.allclose(d0, torch.zeros(12)) and torch.equal(wg0, wg)


def test_embedding_image_is_the_zero_padded_weight():
    """pack_embedding_frag = pack_weight_frag of the weight with its columns zero-padded to a multiple of 128 (the row chain's first GEMM walks
    K in groups of 128); the fragment-major re-ordering is a pure permutation"""
    from anemoi_core_amd import ops

    w = torch.arange(512 * 192, dtype=torch.float32).reshape(512, 192)
    img = ops.pack_embedding_frag(w)
    assert img.numel() == 512 * 256
    ref = ops.pack_weight_frag(F.pad(w, (0, 64)))
    assert torch.equal(img, ref)
    assert torch.equal(torch.sort(img)[0][-(512 * 192 - 1):], torch.sort(w.flatten())[0][1:])  # every weight once, the rest zeros
    assert ops.pack_embedding_frag(torch.ones(512, 128)).numel() == 512 * 128  # already a multiple of 128: unchanged size


def test_new_launches_refuse_cpu_tensors_and_report_unsupported_shapes():
    from anemoi_core_amd import ops

    x = torch.randn(10, 512).to(torch.bfloat16)
    assert not ops.gt_row_chain_supported(x, 1024) and not ops.gt_cluster_chain_supported(x, 2048, 2048) and not ops.gt_layer_chain2_supported(x, 2048, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gt_row_chain(x, x.flatten(), x.flatten(), x.flatten(), 1024, 1e-5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gt_cluster_chain(x, x, x.flatten(), x.flatten(), x.flatten(), x.flatten(), 2048, 1e-5)


def test_gates_and_their_defaults():
    """the row counts at which the launches of round 6 take over (INTEGRATION.md switch table)"""
    import anemoi_core_amd.layers.block as B
    import anemoi_core_amd.layers.mapper as M

    assert B._LAYER_CHAIN and B._LAYER_CHAIN_MIN_ROWS == 4096 and B._CLUSTER_CHAIN and B._CLUSTER_HALO
    assert M._ROW_CHAIN and M._ROW_CHAIN_MIN_ROWS == 4096 and M._ROW_CHAIN_GEMM_BAND == (16384, 262144) and M._TAIL_PROJ


def test_decoder_offers_its_extractor_only_at_inference():
    from anemoi_core_amd.layers.mapper import GraphTransformerBackwardMapper, GraphTransformerForwardMapper

    kw = dict(in_channels_src=16, in_channels_dst=8, hidden_dim=32, num_chunks=1, num_heads=4, mlp_hidden_ratio=2, edge_dim=3)
    dec = GraphTransformerBackwardMapper(out_channels_dst=5, **kw)
    enc = GraphTransformerForwardMapper(**kw)
    assert enc._tail_projection() is None
    assert dec._tail_projection() is None  # grad mode on: the differentiable path
    with torch.no_grad():
        ln, lin = dec._tail_projection()
    assert ln is dec.node_data_extractor[0] and lin is dec.node_data_extractor[1]
