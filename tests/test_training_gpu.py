"""Training path of the GraphTransformer family (scope row f1): loss.backward() through the nn.Module mirror runs the HIP
backward kernels (GEMM dX/dW through the MFMA kernels, LayerNorm / GELU / bias backward, attention dst/src passes) and must
reproduce torch autograd of the fp32 oracle: input gradient and EVERY parameter gradient of the tiny reference model.

Tolerance (fp32): |got - want| <= 2e-4 * max|want| + 1e-6 per tensor — the accumulated fp32 re-association of a
two-layer encoder-processor-decoder backward; the reference's own gradient checks use atol 1e-3..1e-2
(models/tests/integration/triton/test_triton_gt.py:168-184)."""
import pytest
import torch

from oracle import gt_oracle as O
from tests.conftest import load_golden
from tests.helpers import build_model_from_fixture, lk

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, want, what, rtol=2e-4, atol=1e-6):
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = float((got - want).abs().max())
    assert err <= rtol * float(want.abs().max()) + atol, f"{what}: max err {err:.3e} vs max |want| {float(want.abs().max()):.3e}"


def test_linear_and_layernorm_autograd_vs_torch():
    from anemoi_core_amd import ops

    gen = torch.Generator().manual_seed(0)
    for dtype, rtol in ((torch.float32, 2e-5), (torch.bfloat16, 3e-2)):
        for N, K, Oo in ((300, 64, 128), (1030, 512, 512), (77, 20, 36)):
            x = torch.randn(N, K, generator=gen).to(dtype)
            w = (torch.randn(Oo, K, generator=gen) / K**0.5).to(dtype)
            b = (0.1 * torch.randn(Oo, generator=gen)).to(dtype)
            r = torch.randn(N, Oo, generator=gen).to(dtype)
            g = torch.randn(N, Oo, generator=gen).to(dtype)
            for act in (None, "gelu"):
                leaves = [t.detach().clone().to(DEV).requires_grad_(True) for t in (x, w, b, r)]
                y = ops.linear(leaves[0], leaves[1], leaves[2], act=act, residual=leaves[3])
                y.backward(g.to(DEV))
                ref = [t.detach().clone().float().requires_grad_(True) for t in (x, w, b, r)]
                z = torch.nn.functional.linear(ref[0], ref[1], ref[2])
                z = torch.nn.functional.gelu(z) if act else z
                (z + ref[3]).backward(g.float())
                for name, a, c in zip(("dx", "dw", "db", "dres"), leaves, ref):
                    _close(a.grad, c.grad, f"linear {name} {dtype} {act} {(N, K, Oo)}", rtol)
        for N, D in ((300, 512), (50, 100), (1029, 64)):
            x = (1.5 * torch.randn(N, D, generator=gen) + 0.3).to(dtype)
            w, b = (1 + 0.1 * torch.randn(D, generator=gen)).to(dtype), (0.1 * torch.randn(D, generator=gen)).to(dtype)
            g = torch.randn(N, D, generator=gen).to(dtype)
            leaves = [t.detach().clone().to(DEV).requires_grad_(True) for t in (x, w, b)]
            ops.layer_norm(*leaves).backward(g.to(DEV))
            ref = [t.detach().clone().float().requires_grad_(True) for t in (x, w, b)]
            torch.nn.functional.layer_norm(ref[0], (D,), ref[1], ref[2]).backward(g.float())
            for name, a, c in zip(("dx", "dgamma", "dbeta"), leaves, ref):
                _close(a.grad, c.grad, f"layer_norm {name} {dtype} {(N, D)}", rtol)


def test_processor_block_gradients_match_oracle():
    from anemoi_core_amd.distributed.shapes import GraphShardInfo
    from anemoi_core_amd.layers.block import GraphTransformerProcessorBlock

    for tag in ("proc", "proc_qknorm"):
        c = load_golden("blocks.pt")[tag]
        blk = GraphTransformerProcessorBlock(layer_kernels=lk(), **c["cfg"]).to(DEV)
        blk.load_state_dict(c["params"], strict=True)
        x = c["x"].to(DEV).requires_grad_(True)
        w = torch.randn(c["x"].shape, generator=torch.Generator().manual_seed(1))
        out, _ = blk(x, c["edge_attr"].to(DEV), c["edge_index"].to(DEV), GraphShardInfo(), 1, c["x"].shape[0])
        (out * w.to(DEV)).sum().backward()
        p = {"b." + k: v.clone().requires_grad_(True) for k, v in c["params"].items()}
        xo = c["x"].clone().requires_grad_(True)
        want = O.gt_processor_block(p, "b", xo, c["edge_attr"], c["edge_index"], c["cfg"]["num_heads"])
        (want * w).sum().backward()
        _close(out.detach(), want.detach(), f"{tag} forward", 1e-5)
        _close(x.grad, xo.grad, f"{tag} dx")
        for name, prm in blk.named_parameters():
            if p["b." + name].grad is None:
                assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
                continue
            _close(prm.grad, p["b." + name].grad, f"{tag} d{name}")


def test_full_model_gradients_match_oracle():
    c = load_golden("model_tiny.pt")["gt"]
    model, g = build_model_from_fixture(c)
    model.load_state_dict(c["params"], strict=True)
    model = model.to(DEV).train()
    x = c["x"].to(DEV).requires_grad_(True)
    w = torch.randn(c["out"].shape, generator=torch.Generator().manual_seed(2))
    out = model({"data": x})["data"]
    (out * w.to(DEV)).sum().backward()
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in c["params"].items()}
    xo = c["x"].clone().requires_grad_(True)
    want = O.enc_proc_dec_forward(p, c["cfg"], g, xo)
    (want * w).sum().backward()
    _close(out.detach(), want.detach(), "forward", 2e-5)
    _close(x.grad, xo.grad, "d input")
    n_checked = 0
    for name, prm in model.named_parameters():
        ref = p[name].grad
        if ref is None:
            continue
        assert prm.grad is not None, f"no gradient for {name}"
        _close(prm.grad, ref, f"d {name}")
        n_checked += 1
    assert n_checked >= 60, n_checked  # every weight/bias/trainable tensor of encoder, 2 processor layers, decoder


def test_graphconv_blocks_gradients_match_oracle():
    from anemoi_core_amd.distributed.shapes import GraphShardInfo
    from anemoi_core_amd.layers.block import GraphConvProcessorBlock

    for tag in ("gconv_proc", "gconv_proc_emb"):
        c = load_golden("blocks.pt")[tag]
        blk = GraphConvProcessorBlock(layer_kernels=lk(), **c["cfg"]).to(DEV)
        blk.load_state_dict(c["params"], strict=True)
        x = c["x"].to(DEV).requires_grad_(True)
        ea = c["edge_attr"].to(DEV).requires_grad_(True)
        gen = torch.Generator().manual_seed(4)
        nodes, edges = blk(x, ea, c["edge_index"].to(DEV), GraphShardInfo(), size=(c["x"].shape[0], c["x"].shape[0]))
        wn, we = torch.randn(nodes.shape, generator=gen), torch.randn(edges.shape, generator=gen)
        ((nodes * wn.to(DEV)).sum() + (edges * we.to(DEV)).sum()).backward()
        p = {"b." + k: v.clone().requires_grad_(True) for k, v in c["params"].items()}
        xo, eo = c["x"].clone().requires_grad_(True), c["edge_attr"].clone().requires_grad_(True)
        n_ref, e_ref = O.gconv_processor_block(p, "b", xo, eo, c["edge_index"])
        ((n_ref * wn).sum() + (e_ref * we).sum()).backward()
        _close(nodes.detach(), n_ref.detach(), f"{tag} nodes", 2e-5)
        _close(edges.detach(), e_ref.detach(), f"{tag} edges", 2e-5)
        _close(x.grad, xo.grad, f"{tag} dx")
        _close(ea.grad, eo.grad, f"{tag} d edge_attr")
        for name, prm in blk.named_parameters():
            _close(prm.grad, p["b." + name].grad, f"{tag} d{name}")


def test_graphconv_blocks_with_gated_mlps_match_the_reference_forward_and_autograd():
    """GraphConv processor blocks with swiglu / geglu edge MLP, node MLP and edge embedding (the first gated edge layer runs as the
    gather-add GEMM on the fused [gate; value] weight + the gating kernel), and a swiglu mapper block: outputs and ALL gradients
    against the REFERENCE's own forward / autograd (fixture gnn_gated.pt)."""
    from anemoi_core_amd.distributed.shapes import BipartiteGraphShardInfo, GraphShardInfo
    from anemoi_core_amd.layers.block import GraphConvMapperBlock, GraphConvProcessorBlock

    g = load_golden("gnn_gated.pt")
    for kind in ("swiglu", "geglu"):
        c = g[f"proc_{kind}"]
        blk = GraphConvProcessorBlock(layer_kernels=lk(), **c["cfg"]).to(DEV)
        blk.load_state_dict(c["params"], strict=True)
        n = c["x"].shape[0]
        with torch.no_grad():  # inference path (stacked node-level weight is not used for gated layers)
            y0, e0 = blk(c["x"].to(DEV), c["edge_attr"].to(DEV), c["edge_index"].to(DEV), GraphShardInfo(), size=(n, n))
        _close(y0, c["out"], f"{kind} nodes (no grad)", 2e-5)
        _close(e0, c["edges_out"], f"{kind} edges (no grad)", 2e-5)
        x, ea = c["x"].to(DEV).requires_grad_(True), c["edge_attr"].to(DEV).requires_grad_(True)
        y, e = blk(x, ea, c["edge_index"].to(DEV), GraphShardInfo(), size=(n, n))
        ((y * c["w_out"].to(DEV)).sum() + (e * c["w_edges"].to(DEV)).sum()).backward()
        _close(y.detach(), c["out"], f"{kind} nodes", 2e-5)
        _close(e.detach(), c["edges_out"], f"{kind} edges", 2e-5)
        _close(x.grad, c["grad_x"], f"{kind} dx")
        _close(ea.grad, c["grad_edge_attr"], f"{kind} d edge_attr")
        for name, prm in blk.named_parameters():
            _close(prm.grad, c["grads"][name], f"{kind} d{name}")
    c = g["map_swiglu"]
    blk = GraphConvMapperBlock(layer_kernels=lk(), **c["cfg"]).to(DEV)
    blk.load_state_dict(c["params"], strict=True)
    ns, nd = c["x_src"].shape[0], c["x_dst"].shape[0]
    with torch.no_grad():
        (ys, yd), e = blk((c["x_src"].to(DEV), c["x_dst"].to(DEV)), c["edge_attr"].to(DEV), c["edge_index"].to(DEV), BipartiteGraphShardInfo(),
                          None, size=(ns, nd))
    _close(ys, c["out_src"], "mapper src", 2e-5)
    _close(yd, c["out_dst"], "mapper dst", 2e-5)
    _close(e, c["edges_out"], "mapper edges", 2e-5)


def test_full_gnn_model_gradients_match_oracle():
    c = load_golden("model_tiny.pt")["gnn"]
    model, g = build_model_from_fixture(c)
    model.load_state_dict(c["params"], strict=True)
    model = model.to(DEV).train()
    x = c["x"].to(DEV).requires_grad_(True)
    w = torch.randn(c["out"].shape, generator=torch.Generator().manual_seed(2))
    out = model({"data": x})["data"]
    (out * w.to(DEV)).sum().backward()
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in c["params"].items()}
    xo = c["x"].clone().requires_grad_(True)
    want = O.enc_proc_dec_forward(p, c["cfg"], g, xo)
    (want * w).sum().backward()
    _close(out.detach(), want.detach(), "forward", 2e-5)
    _close(x.grad, xo.grad, "d input")
    n_checked = 0
    for name, prm in model.named_parameters():
        ref = p[name].grad
        if ref is None:
            continue
        assert prm.grad is not None, f"no gradient for {name}"
        _close(prm.grad, ref, f"d {name}")
        n_checked += 1
    assert n_checked >= 60, n_checked


def test_conditional_layernorm_block_gradients_match_oracle():
    """ConditionalLayerNorm (scope row f3) inside a GraphTransformer processor block: gradients w.r.t. x, the conditioning
    and every parameter (incl. the scale / bias maps of the conditioning) == oracle autograd."""
    from anemoi_core_amd.distributed.shapes import GraphShardInfo
    from anemoi_core_amd.layers.block import GraphTransformerProcessorBlock
    from anemoi_core_amd.layers.utils import load_layer_kernels

    c = load_golden("variants.pt")["cond"]["block"]
    lk_c = load_layer_kernels({"LayerNorm": {"_target_": "anemoi.models.layers.normalization.ConditionalLayerNorm", "condition_shape": 16,
                                             "zero_init": False}})
    blk = GraphTransformerProcessorBlock(layer_kernels=lk_c, **c["cfg"]).to(DEV)
    blk.load_state_dict(c["params"], strict=True)
    x, cond = c["x"].to(DEV).requires_grad_(True), c["cond"].to(DEV).requires_grad_(True)
    w = torch.randn(c["out"].shape, generator=torch.Generator().manual_seed(1))
    out, _ = blk(x, c["edge_attr"].to(DEV), c["edge_index"].to(DEV), GraphShardInfo(), 1, c["x"].shape[0], cond=cond)
    (out * w.to(DEV)).sum().backward()
    p = {"b." + k: v.clone().requires_grad_(True) for k, v in c["params"].items()}
    xo, co = c["x"].clone().requires_grad_(True), c["cond"].clone().requires_grad_(True)
    want = O.gt_processor_block(p, "b", xo, c["edge_attr"], c["edge_index"], c["cfg"]["num_heads"], cond=co)
    (want * w).sum().backward()
    _close(out.detach(), want.detach(), "forward", 2e-5)
    _close(x.grad, xo.grad, "dx")
    _close(cond.grad, co.grad, "d cond")
    for name, prm in blk.named_parameters():
        # lin_key.bias has a ZERO true gradient (softmax is shift-invariant per destination): both sides are round-off there
        _close(prm.grad, p["b." + name].grad, f"d{name}", atol=2e-5)


@pytest.mark.parametrize("kind", ["gt", "gnn"])
def test_full_model_gradients_match_reference_autograd(kind):
    """The HIP backward against the REFERENCE's own autograd (fixture model_tiny_grads.pt generated by importing the reference:
    input gradient and every parameter gradient, loss = sum(out * w) with the seeded w)."""
    c, r = load_golden("model_tiny.pt")[kind], load_golden("model_tiny_grads.pt")[kind]
    model, _ = build_model_from_fixture(c)
    model.load_state_dict(c["params"], strict=True)
    model = model.to(DEV).train()
    x = c["x"].to(DEV).requires_grad_(True)
    w = torch.randn(c["out"].shape, generator=torch.Generator().manual_seed(r["loss_weight_seed"]))
    (model({"data": x})["data"] * w.to(DEV)).sum().backward()
    _close(x.grad, r["dx"], "d input")
    got = dict(model.named_parameters())
    assert set(r["grads"]) <= set(got)
    for name, ref in r["grads"].items():
        assert got[name].grad is not None, f"no gradient for {name}"
        _close(got[name].grad, ref, f"d {name}")
    assert len(r["grads"]) >= 60


def test_training_step_captured_as_one_hip_graph_equals_eager():
    """Forward + backward of the tiny GraphTransformer model captured into ONE hipGraph (static input buffer, gradients allocated
    from the graph's pool): replaying it on new input values gives bit-identical gradients to an eager step on the same values."""
    c = load_golden("model_tiny.pt")["gt"]
    model, _ = build_model_from_fixture(c)
    model.load_state_dict(c["params"], strict=True)
    model = model.to(DEV).train()
    x_static = c["x"].to(DEV).clone()
    w = torch.randn(c["out"].shape, generator=torch.Generator().manual_seed(2)).to(DEV)

    def step():
        model.zero_grad(set_to_none=True)
        (model({"data": x_static})["data"] * w).sum().backward()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):  # static caches (CSC, reverse CSR, plans) are built outside the capture
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    captured = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}  # static tensors of the graph's pool
    x_new = 0.5 * c["x"].to(DEV) + 0.25
    x_static.copy_(x_new)
    graph.replay()
    torch.cuda.synchronize()
    got = {k: g.clone() for k, g in captured.items()}
    step()  # eager, same values
    torch.cuda.synchronize()
    assert len(got) >= 60
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.equal(got[k], p.grad), k
    x_static.copy_(c["x"].to(DEV))
    graph.replay()
    torch.cuda.synchronize()
    assert any(not torch.equal(captured[k], got[k]) for k in got)  # the replay really depends on the input buffer


@pytest.mark.parametrize("tag", ["proc_edgepre_qknorm", "proc_edgepre", "map_edgepre_qknorm"])
def test_edge_pre_mlp_and_qk_norm_blocks_train_on_the_fused_edge_path(tag, monkeypatch):
    """blocks with qk_norm and / or edge_pre_mlp (block.py:585-586, 637-687) train through the FUSED-edge
    attention (neither E nor dE materialised): output and every gradient - input, edge attributes, lin_edge, edge_pre_mlp,
    q_norm / k_norm, all other parameters - equal the REFERENCE's own autograd (fixture blocks_train.pt), and the fused Function
    really is what ran."""
    from anemoi_core_amd import autograd as ag
    from anemoi_core_amd.distributed.shapes import BipartiteGraphShardInfo, GraphShardInfo
    from anemoi_core_amd.layers.block import GraphTransformerMapperBlock, GraphTransformerProcessorBlock

    c = load_golden("blocks_train.pt")[tag]
    calls = []
    real = ag.fused_edge_attention
    monkeypatch.setattr(ag, "fused_edge_attention", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    cls = GraphTransformerProcessorBlock if tag.startswith("proc") else GraphTransformerMapperBlock
    blk = cls(layer_kernels=lk(), **c["cfg"]).to(DEV).train()
    blk.load_state_dict(c["params"], strict=True)
    ea = c["edge_attr"].to(DEV).requires_grad_(True)
    w = c["w"].to(DEV)
    if tag.startswith("proc"):
        x = c["x"].to(DEV).requires_grad_(True)
        out, _ = blk(x, ea, c["edge_index"].to(DEV), GraphShardInfo(), 1, c["x"].shape[0])
        (out * w).sum().backward()
        _close(out.detach(), c["out"], f"{tag} forward", 1e-5)
        _close(x.grad, c["dx"], f"{tag} dx")
    else:
        xs, xd = c["x_src"].to(DEV).requires_grad_(True), c["x_dst"].to(DEV).requires_grad_(True)
        (_, out), _ = blk((xs, xd), ea, c["edge_index"].to(DEV), BipartiteGraphShardInfo(), 1, (xs.shape[0], xd.shape[0]))
        (out * w).sum().backward()
        _close(out.detach(), c["out_dst"], f"{tag} forward", 1e-5)
        _close(xs.grad, c["dx_src"], f"{tag} dx_src")
        _close(xd.grad, c["dx_dst"], f"{tag} dx_dst")
    assert calls, "the fused-edge attention Function did not run"
    _close(ea.grad, c["d_edge_attr"], f"{tag} d edge_attr")
    got = dict(blk.named_parameters())
    for name, g in c["grads"].items():
        _close(got[name].grad, g, f"{tag} d{name}", atol=2e-5)
