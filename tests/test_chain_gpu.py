"""Row-resident layer chain (anemoi_gt_chain2_fwd, csrc/gt_chain2.hip): one launch for a GraphTransformer block's projection + skip,
LayerNorm, MLP + skip and the NEXT block's LayerNorm + fused q|k|v|self projection (reference layers/block.py:1237-1273).

Checked against (a) a torch fp32 restatement with the reference's rounding points (x1, LayerNorm output, hidden, x2 in the model
dtype - what autocast produces), (b) the launch-per-GEMM path of this package on the same inputs, and (c) under OFFSET rows
(|row mean| / sigma up to 64: a trained residual stream need not be centred), next to the LayerNorm-fold path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
D, HD = 512, 2048


def _params(gen, dtype, q_out=2048, beta=True):
    r = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    p = dict(
        wp=(r(D, D) / 22).to(dtype), bp=(0.1 * r(D)).to(dtype),
        g1=(1 + 0.2 * r(D)).to(dtype), be1=(0.1 * r(D)).to(dtype) if beta else None,
        w1=(r(HD, D) / 22).to(dtype), b1=(0.1 * r(HD)).to(dtype),
        w2=(r(D, HD) / 45).to(dtype), b2=(0.1 * r(D)).to(dtype),
        gq=(1 + 0.2 * r(D)).to(dtype), beq=(0.1 * r(D)).to(dtype) if beta else None,
        wq=(r(q_out, D) / 22).to(dtype) if q_out else None, bq=(0.1 * r(q_out)).to(dtype) if q_out else None,
    )
    return p


def _reference(attn, x, p, dtype, extra=None, eps=1e-5):
    """fp32 arithmetic on the 16-bit operands, rounded to the model dtype where the reference's autocast forward rounds."""
    f = lambda t: None if t is None else t.float()  # noqa: E731
    rnd = lambda t: t.to(dtype).float()  # noqa: E731
    x1 = rnd(F.linear(f(attn), f(p["wp"]), f(p["bp"])) + f(x))
    n1 = rnd(F.layer_norm(x1, (D,), f(p["g1"]), f(p["be1"]), eps))
    h = rnd(F.gelu(F.linear(n1, f(p["w1"]), f(p["b1"]))))
    x2 = rnd(F.linear(h, f(p["w2"]), f(p["b2"])) + x1)
    if extra is not None:
        x2 = rnd(x2 + f(extra))
    q = None
    if p["wq"] is not None:
        nq = rnd(F.layer_norm(x2, (D,), f(p["gq"]), f(p["beq"]), eps))
        q = F.linear(nq, f(p["wq"]), f(p["bq"]))
    return x2, q


def _close(got, want, what, tol=2e-2):
    got, want = got.float().cpu(), want.float().cpu()
    scale = float(want.abs().max())
    err = (got - want).abs()
    bound = tol * max(scale, 1e-3) + tol * want.abs()
    assert bool((err <= bound).all()), f"{what}: max err {float(err.max()):.3e} at scale {scale:.3g}, mean {float(err.mean()):.3e}"
    return float(err.max()), float(err.mean())


def test_pack_weight_frag_layout():
    """element (slab, ks, ni, kslot, row, e) of the image = W[slab*64 + ni*16 + row][ks*32 + kslot*8 + e] (include/anemoi_hip.h)."""
    from anemoi_core_amd import ops

    O, K = 128, 64
    w = torch.arange(O * K, dtype=torch.float32).reshape(O, K).to(DEV)
    img = ops.pack_weight_frag(w).reshape(O // 64, K // 32, 4, 4, 16, 8).cpu()
    for slab, ks, ni, kslot, row, e in [(0, 0, 0, 0, 0, 0), (1, 1, 3, 2, 15, 7), (0, 1, 2, 3, 5, 4), (1, 0, 1, 1, 9, 2)]:
        assert float(img[slab, ks, ni, kslot, row, e]) == float((slab * 64 + ni * 16 + row) * K + ks * 32 + kslot * 8 + e)


def test_chain_equals_launch_per_gemm_path():
    """the same block through this package's four-launch path (plain LayerNorm kernels between the GEMMs): equal to bf16 rounding."""
    from anemoi_core_amd import ops

    dtype, N = torch.bfloat16, 10242
    gen = torch.Generator().manual_seed(3)
    p = _params(gen, dtype)
    d = lambda t: t.to(DEV)  # noqa: E731
    attn, x = d(torch.randn(N, D, generator=gen).to(dtype)), d(torch.randn(N, D, generator=gen).to(dtype))
    x1 = ops.linear(attn, d(p["wp"]), d(p["bp"]), residual=x)
    h = ops.linear(ops.layer_norm(x1, d(p["g1"]), d(p["be1"]), 1e-5), d(p["w1"]), d(p["b1"]), act="gelu")
    x2 = ops.linear(h, d(p["w2"]), d(p["b2"]), residual=x1)
    q = ops.linear(ops.layer_norm(x2, d(p["gq"]), d(p["beq"]), 1e-5), d(p["wq"]), d(p["bq"]))
    c2, cq = _run_chain2(ops, attn.cpu(), x.cpu(), p)
    e2 = (c2.float() - x2.float()).abs()
    eq = (cq.float() - q.float()).abs()
    # different accumulation order + one-ulp flips of intermediate roundings: a few ulps of the output scale, mean far below one
    assert float(e2.max()) <= 2e-2 * float(x2.float().abs().max()) and float(e2.mean()) <= 2e-3 * float(x2.float().abs().mean() + 1)
    assert float(eq.max()) <= 3e-2 * float(q.float().abs().max()) and float(eq.mean()) <= 4e-3 * float(q.float().abs().mean() + 1)


@pytest.mark.parametrize("offset_over_sigma", [0.0, 4.0, 16.0, 64.0])
def test_layernorm_under_offset_rows(offset_over_sigma):
    """A residual stream whose rows sit at mu0 = k sigma.  The chain kernel's LayerNorm merges per-wave (mean, M2)
    partials (no E[x^2] - mean^2), so its error against the exact-LayerNorm restatement must not grow with the offset beyond the
    16-bit representation of the rows themselves (the reference's own operand: ulp(x) ~ |mu0| 2^-8).  The LayerNorm-FOLD path
    (stats producer + folding consumer, still used by the mappers' source / destination sides) is measured beside it: its
    c = row sums of the ROUNDED W gamma make rstd (x (W gamma)^T - mean c) algebraically exact in the offset; what is left is
    the fp32 cancellation of var = E[x^2] - mean^2 (relative 2^-23 (1 + mu0^2 / sigma^2))."""
    from anemoi_core_amd import ops

    dtype, N = torch.bfloat16, 4000
    gen = torch.Generator().manual_seed(11)
    p = _params(gen, dtype)
    attn = torch.randn(N, D, generator=gen).to(dtype)
    mu0 = offset_over_sigma * (torch.rand(N, 1, generator=gen) * 2 - 1)  # per-row offsets in [-k, k] sigma
    x = (torch.randn(N, D, generator=gen) + mu0).to(dtype)
    ref2, refq = _reference(attn, x, p, dtype)
    # x2 carries the offset (its scale grows with it); the projections see LayerNorm'd rows: their scale does not.  The kernel takes the
    # statistics from registers, normalises the row WITHOUT the affine part and rounds it, the affine part sits in the rounded weights - no
    # term that grows with the offset
    x2, q = _run_chain2(ops, attn, x, p)
    e2 = _close(x2, ref2, f"x2 offset {offset_over_sigma}")
    eq = _close(q, refq, f"qkvs offset {offset_over_sigma}", tol=2.5e-2)
    # the fold path on the same rows: x1 with statistics, then LN folded into the MLP-1 GEMM
    d = lambda t: t.to(DEV)  # noqa: E731
    r = ops.linear_with_row_stats(d(attn), d(p["wp"]), d(p["bp"]), d(x))
    assert r is not None
    x1, stats = r
    ws = (p["w1"].float() * p["g1"].float()).to(dtype)
    c, dd = ws.float().sum(1).contiguous(), (p["w1"].float() @ p["be1"].float() + p["b1"].float()).contiguous()
    hf = ops.linear_ln_folded(x1, d(ws), d(c), d(dd), stats, 1e-5, "gelu")
    hu = ops.linear(ops.layer_norm(x1, d(p["g1"]), d(p["be1"]), 1e-5), d(p["w1"]), d(p["b1"]), act="gelu")
    x1f = x1.float().cpu()
    href = F.gelu(F.linear(F.layer_norm(x1f, (D,), p["g1"].float(), p["be1"].float(), 1e-5), p["w1"].float(), p["b1"].float()))
    scale = float(href.abs().max())
    ef, eu = float((hf.float().cpu() - href).abs().max()) / scale, float((hu.float().cpu() - href).abs().max()) / scale
    print(f"offset {offset_over_sigma:5.1f} sigma: chain x2 max {e2[0]:.3e} qkvs max {eq[0]:.3e} | MLP-1 fold {ef:.3e} unfused {eu:.3e} (of scale)")
    # bound, as a function of |mean| / sigma: both paths within 2e-2 of the output scale up to 64 sigma (the fold's extra term,
    # 2^-23 (mu0 / sigma)^2 relative in rstd, is 5e-4 at 64 sigma: below one bf16 ulp)
    assert ef <= 2e-2 and eu <= 2e-2


def test_model_with_chain_equals_model_without():
    """AnemoiModelEncProcDec at 512 channels: chain launches (encoder -> processor hand-over of the first block's projections, block
    -> block, last block with the latent skip, decoder) against the launch-per-GEMM path of the same model, and both against the
    fp32 CPU oracle within the full-model bf16 bound (tests/test_fullsize_parity_gpu.py)."""
    import anemoi_core_amd.layers.block as B
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec
    from anemoi_core_amd.models.configs import make_data_indices, model_config
    from oracle import gt_oracle as O

    g = build_synthetic_graph("o16", 3)
    torch.manual_seed(0)
    cfg = dict(kind="gt", num_channels=512, num_layers=3, num_heads=16, trainable=8, n_vars=6, n_step_input=2)
    model = AnemoiModelEncProcDec(model_config=model_config("gt", 512, 3, 16, 8), data_indices=make_data_indices(6, 6),
                                  statistics={"data": None}, n_step_input=2, n_step_output=1, graph_data=g).eval()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(1, 2, 1, g.num_data, 6)
    want = O.enc_proc_dec_forward(params, cfg, g, x)
    m = model.to(DEV).to(torch.bfloat16)
    xb = x.to(DEV).to(torch.bfloat16)
    outs = {}
    saved = (B._LAYER_CHAIN, B._LAYER_CHAIN_MIN_ROWS)
    for flag in ("chain", "off"):
        # (by default only blocks of >= 4 096 rows take the chain; this mesh has 642 hidden nodes)
        B._LAYER_CHAIN, B._LAYER_CHAIN_MIN_ROWS = flag != "off", 0
        try:
            with torch.no_grad():
                outs[flag] = m({"data": xb})["data"].float().cpu()
        finally:
            B._LAYER_CHAIN, B._LAYER_CHAIN_MIN_ROWS = saved
    a, b = outs["chain"], outs["off"]
    scale = float(want.abs().max())
    assert not torch.equal(a, b)  # two different paths really ran
    assert float((a - b).abs().max()) <= 3e-2 * scale and float((a - b).abs().mean()) <= 4e-3 * scale, (float((a - b).abs().max()), scale)
    for name, y in (("layer chain", a), ("launch-per-GEMM", b)):
        err = (y - want).abs()
        assert float(err.max()) <= 6e-2 * max(scale, 1.0) and float(err.mean()) <= 1e-2 * max(scale, 1.0), (name, float(err.max()), float(err.mean()), scale)


# ------------------------------------------------------------------------------------------ GraphConv (GNN) chains
def _gnn_params(gen, dtype):
    r = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    return dict(w0=(r(D, D) / 22).to(dtype), b0=(0.1 * r(D)).to(dtype), w1=(r(D, D) / 22).to(dtype), b1=(0.1 * r(D)).to(dtype),
                w2=(r(D, D) / 22).to(dtype), b2=(0.1 * r(D)).to(dtype), g=(1 + 0.2 * r(D)).to(dtype), be=(0.1 * r(D)).to(dtype),
                wa=(r(D, 2 * D) / 32).to(dtype), ba=(0.1 * r(D)).to(dtype), wt=(r(2 * D, D) / 22).to(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N", [(37, 11), (5000, 700), (81840, 10242)])
def test_gnn_edge_chain_vs_fp32_restatement(dtype, M, N):
    """GraphConv's edge MLP in gather-add form + LayerNorm + residual (csrc/gnn_chain.hip) against fp32 torch with the rounding
    points of the launch-per-GEMM path, and against that path itself (ops.linear with the gather-add epilogue, ops.linear x 2,
    edge_ln_residual_segment_sum) on the same operands."""
    from anemoi_core_amd import ops

    gen = torch.Generator().manual_seed(M)
    p = _gnn_params(gen, dtype)
    e = torch.randn(M, D, generator=gen).to(dtype)
    g12 = torch.randn(N, 2 * D, generator=gen).to(dtype)
    dst = torch.sort(torch.randint(0, N, (M,), generator=gen)).values.to(torch.int32)
    src = torch.randint(0, N, (M,), generator=gen).to(torch.int32)
    d = lambda t: t.to(DEV)  # noqa: E731
    P = ops.pack_weight_frag
    got = ops.gnn_edge_chain(d(e), d(g12)[:, :D], d(dst), d(g12)[:, D:], d(src), P(d(p["w0"])), d(p["b0"]), P(d(p["w1"])), d(p["b1"]), P(d(p["w2"])),
                             d(p["b2"]), d(p["g"]), d(p["be"]), 1e-5)
    f = lambda t: t.float()  # noqa: E731
    rnd = lambda t: t.to(dtype).float()  # noqa: E731
    h1 = rnd(F.gelu(F.linear(f(e), f(p["w0"]), f(p["b0"])) + f(g12)[dst.long(), :D] + f(g12)[src.long(), D:]))
    h2 = rnd(F.gelu(F.linear(h1, f(p["w1"]), f(p["b1"]))))
    z = rnd(F.linear(h2, f(p["w2"]), f(p["b2"])))
    want = rnd(F.layer_norm(z, (D,), f(p["g"]), f(p["be"]), 1e-5) + f(e))
    _close(got, want, f"edge chain M={M}", tol=2.5e-2)
    assert torch.equal(got, ops.gnn_edge_chain(d(e), d(g12)[:, :D], d(dst), d(g12)[:, D:], d(src), P(d(p["w0"])), d(p["b0"]), P(d(p["w1"])), d(p["b1"]),
                                               P(d(p["w2"])), d(p["b2"]), d(p["g"]), d(p["be"]), 1e-5))
    if M >= 5000:  # the launch-per-GEMM path of this package on the same operands (needs a CSC over the sorted destinations)
        colptr = torch.zeros(N + 1, dtype=torch.int64)
        colptr[1:] = torch.bincount(dst.long(), minlength=N).cumsum(0)
        csc = ops.CSC(row=d(src), dst=d(dst), colptr=d(colptr.to(torch.int32)), n_src=N, n_dst=N)
        h = ops.linear(d(e), d(p["w0"]), d(p["b0"]), act="gelu", g1=d(g12)[:, :D], idx1=d(dst), g2=d(g12)[:, D:], idx2=d(src))
        zz = ops.linear(ops.linear(h, d(p["w1"]), d(p["b1"]), act="gelu"), d(p["w2"]), d(p["b2"]))
        e_old_path, agg_path = ops.edge_ln_residual_segment_sum(zz, d(e), d(p["g"]), d(p["be"]), 1e-5, csc)
        err = (got.float() - e_old_path.float()).abs()
        assert float(err.max()) <= 4e-2 * float(want.abs().max()) and float(err.mean()) <= 3e-3 * float(want.abs().mean() + 1)
        agg = ops.segment_sum_rows(got, csc.colptr)
        ea = (agg.float() - agg_path.float()).abs()
        assert float(ea.max()) <= 4e-2 * float(agg_path.float().abs().max()) + 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,trailing", [(9, True), (3000, False), (10242, True), (40962, True)])
def test_gnn_node_chain_vs_fp32_restatement(dtype, N, trailing):
    from anemoi_core_amd import ops

    gen = torch.Generator().manual_seed(N)
    p = _gnn_params(gen, dtype)
    x = torch.randn(N, D, generator=gen).to(dtype)
    agg = (3.0 * torch.randn(N, D, generator=gen)).to(dtype)
    d = lambda t: t.to(DEV)  # noqa: E731
    P = ops.pack_weight_frag
    kw = dict(wt=P(d(p["wt"])), t_out_features=2 * D) if trailing else {}
    res = ops.gnn_node_chain(d(x), d(agg), P(d(p["wa"])), d(p["ba"]), P(d(p["w1"])), d(p["b1"]), P(d(p["w2"])), d(p["b2"]), d(p["g"]), d(p["be"]), 1e-5, **kw)
    f = lambda t: t.float()  # noqa: E731
    rnd = lambda t: t.to(dtype).float()  # noqa: E731
    h1 = rnd(F.gelu(F.linear(torch.cat([f(x), f(agg)], 1), f(p["wa"]), f(p["ba"]))))
    h2 = rnd(F.gelu(F.linear(h1, f(p["w1"]), f(p["b1"]))))
    y = rnd(F.linear(h2, f(p["w2"]), f(p["b2"])))
    want = rnd(F.layer_norm(y, (D,), f(p["g"]), f(p["be"]), 1e-5) + f(x))
    if trailing:
        _close(res[0], want, f"node chain N={N}", tol=2.5e-2)
        _close(res[1], F.linear(want, f(p["wt"])), f"node chain trailing N={N}", tol=2.5e-2)
    else:
        _close(res, want, f"node chain N={N}", tol=2.5e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,deg,trailing", [(9, 3, True), (1000, 8, False), (10242, 8, True), (5000, 40, False)])
def test_gnn_node_chain_with_the_scatter_sum_inside(dtype, N, deg, trailing):
    """ops.gnn_node_chain(seg_ptr=...): the kernel sums a panel's in-edge rows itself (fp32, edge order, one rounding - the arithmetic of
    ops.segment_sum_rows) - bit-equal to the launch fed with the materialised segment sums, incl. destinations without edges and degrees
    beyond one batch of loads."""
    from anemoi_core_amd import ops

    gen = torch.Generator().manual_seed(N + deg)
    p = _gnn_params(gen, dtype)
    x = torch.randn(N, D, generator=gen).to(dtype)
    degs = torch.randint(0, 2 * deg + 1, (N,), generator=gen)
    degs[::7] = 0
    ptr = torch.zeros(N + 1, dtype=torch.int32)
    ptr[1:] = degs.cumsum(0).to(torch.int32)
    M = int(ptr[-1])
    e = torch.randn(M, D, generator=gen).to(dtype)
    d = lambda t: t.to(DEV)  # noqa: E731
    P = ops.pack_weight_frag
    kw = dict(wt=P(d(p["wt"])), t_out_features=2 * D) if trailing else {}
    w = (P(d(p["wa"])), d(p["ba"]), P(d(p["w1"])), d(p["b1"]), P(d(p["w2"])), d(p["b2"]), d(p["g"]), d(p["be"]), 1e-5)
    agg = ops.segment_sum_rows(d(e), d(ptr))
    want = ops.gnn_node_chain(d(x), agg, *w, **kw)
    got = ops.gnn_node_chain(d(x), d(e), *w, seg_ptr=d(ptr), **kw)
    if trailing:
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    else:
        assert torch.equal(got, want)
    ref = torch.zeros(N, D).index_add_(0, torch.repeat_interleave(torch.arange(N), degs), e.float())
    assert float((agg.float().cpu() - ref).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K,with_res", [(1500, 11, False), (5000, 100, True), (40320, 200, False), (81840, 3, False), (2049, 512, True), (3000, 384, False)])
def test_embedding_mlp_chain(dtype, N, K, with_res):
    """An embedding MLP (Linear-GELU-Linear-GELU-Linear-LayerNorm into 512 channels, raw input width K) through MLP.forward: the
    row-resident launch (csrc/gnn_chain.hip, MLP instantiation) against fp32 torch with the rounding points of the launch-per-GEMM path
    and against that path itself (ANEMOI_GNN_CHAIN=0) on the same module."""
    from anemoi_core_amd.layers import conv as C
    from anemoi_core_amd.layers.mlp import MLP
    from anemoi_core_amd.layers.utils import load_layer_kernels

    torch.manual_seed(N + K)
    m = MLP(K, D, D, layer_kernels=load_layer_kernels(None), n_extra_layers=1).to(DEV).to(dtype).eval()
    with torch.no_grad():
        m.layer_norm.weight.uniform_(0.5, 1.5)
        m.layer_norm.bias.uniform_(-0.3, 0.3)
    gen = torch.Generator().manual_seed(K)
    x32 = torch.randn(N, K, generator=gen)  # fp32 geometric attributes entering a 16-bit model (cast while padding)
    res = torch.randn(N, D, generator=gen).to(dtype).to(DEV) if with_res else None
    saved = C._GNN_CHAIN
    try:
        with torch.no_grad():
            C._GNN_CHAIN = True
            assert m._embedding_chain_ok(x32.to(DEV), list(m.mlp))
            got = m(x32.to(DEV), residual=res)
            got16 = m(x32.to(dtype).to(DEV), residual=res)  # input already in the model dtype
            C._GNN_CHAIN = False
            path = m(x32.to(DEV), residual=res)
    finally:
        C._GNN_CHAIN = saved
    f = lambda t: t.detach().float().cpu()  # noqa: E731
    rnd = lambda t: t.to(dtype).float()  # noqa: E731
    l0, l1, l2, ln = m.mlp[0], m.mlp[2], m.mlp[4], m.layer_norm
    h1 = rnd(F.gelu(F.linear(rnd(x32), f(l0.weight), f(l0.bias))))
    h2 = rnd(F.gelu(F.linear(h1, f(l1.weight), f(l1.bias))))
    z = rnd(F.linear(h2, f(l2.weight), f(l2.bias)))
    want = F.layer_norm(z, (D,), f(ln.weight), f(ln.bias), ln.eps)
    want = rnd(want + f(res)) if with_res else rnd(want)
    _close(got, want, f"embedding chain N={N} K={K}", tol=2.5e-2)
    assert torch.equal(got, got16)
    err = (got.float() - path.float()).abs()
    assert float(err.max()) <= 4e-2 * float(want.abs().max()) and float(err.mean()) <= 3e-3 * float(want.abs().mean() + 1)


def test_gnn_model_with_chains_equals_model_without():
    """The 512-channel GNN model: chain launches (edge chain + segment sum + node chain with the next block's stacked projection) against
    the launch-per-GEMM path and the fp32 CPU oracle."""
    import anemoi_core_amd.layers.conv as B
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec
    from anemoi_core_amd.models.configs import make_data_indices, model_config
    from oracle import gt_oracle as O

    g = build_synthetic_graph("o16", 3)
    torch.manual_seed(0)
    cfg = dict(kind="gnn", num_channels=512, num_layers=3, num_heads=16, trainable=8, n_vars=6, n_step_input=2)
    model = AnemoiModelEncProcDec(model_config=model_config("gnn", 512, 3, 16, 8), data_indices=make_data_indices(6, 6),
                                  statistics={"data": None}, n_step_input=2, n_step_output=1, graph_data=g).eval()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(1, 2, 1, g.num_data, 6)
    want = O.enc_proc_dec_forward(params, cfg, g, x)
    m = model.to(DEV).to(torch.bfloat16)
    xb = x.to(DEV).to(torch.bfloat16)
    outs, saved = {}, B._GNN_CHAIN
    for flag in (True, False):
        B._GNN_CHAIN = flag
        try:
            with torch.no_grad():
                outs[flag] = m({"data": xb})["data"].float().cpu()
        finally:
            B._GNN_CHAIN = saved
    a, b = outs[True], outs[False]
    scale = float(want.abs().max())
    assert not torch.equal(a, b)  # the chain kernels really ran
    assert float((a - b).abs().max()) <= 4e-2 * scale and float((a - b).abs().mean()) <= 5e-3 * scale, (float((a - b).abs().max()), scale)
    for name, y in (("chain", a), ("launch-per-GEMM", b)):
        err = (y - want).abs()
        assert float(err.max()) <= 6e-2 * max(scale, 1.0) and float(err.mean()) <= 1e-2 * max(scale, 1.0), (name, float(err.max()), float(err.mean()), scale)


# ------------------------------------------------------------------------------------------ the layer chain (csrc/gt_chain2.hip)
def _run_chain2(ops, attn, x, p, extra=None, rows_per_tile=0, **kw):
    """the caller's side of anemoi_gt_chain2_fwd: the LayerNorms' affine parts folded into the Linears that follow them"""
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    dt = attn.dtype
    w1g, d1 = ops.fold_layer_norm(d(p["w1"]), d(p["b1"]), d(p["g1"]), d(p["be1"]))
    parts = [d(p["bp"]).float(), d1, d(p["b2"]).float()]
    wqg, qf = None, 0
    if p["wq"] is not None:
        wq_, dq = ops.fold_layer_norm(d(p["wq"]), d(p["bq"]), d(p["gq"]), d(p["beq"]))
        wqg, qf = ops.pack_weight_frag(wq_), p["wq"].shape[0]
        parts.append(dq)
    vec = torch.cat(parts).to(dt).contiguous()
    return ops.gt_layer_chain2(d(attn), d(x), ops.pack_weight_frag(d(p["wp"])), ops.pack_weight_frag(w1g), ops.pack_weight_frag(d(p["w2"])), vec,
                               p["w1"].shape[0], 1e-5, extra=d(extra), wqg=wqg, q_out_features=qf, lnq_eps=1e-5, rows_per_tile=rows_per_tile, **kw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,rows_per_tile", [(41, 0), (7, 0), (1, 0), (1000, 0), (10242, 0), (10242, 41), (12345, 17), (40962, 0)])
def test_chain2_vs_fp32_restatement(dtype, N, rows_per_tile):
    """the role-split kernel against the same fp32 restatement (the reference's rounding points; its LayerNorm output is rounded WITH the
    affine part, the kernel's without - the affine part rides in the rounded weights): every row of x2 and of the trailing projection,
    ragged last panels, several panels per workgroup, panel heights that are no multiple of the 16-row MFMA band."""
    from anemoi_core_amd import ops

    gen = torch.Generator().manual_seed(N + rows_per_tile)
    p = _params(gen, dtype)
    attn = torch.randn(N, D, generator=gen).to(dtype)
    x = (2.0 * torch.randn(N, D, generator=gen) + 0.5).to(dtype)
    x2, q = _run_chain2(ops, attn, x, p, rows_per_tile=rows_per_tile)
    ref2, refq = _reference(attn, x, p, dtype)
    _close(x2, ref2, f"x2 N={N}")
    _close(q, refq, f"qkvs N={N}")
    x2b, qb = _run_chain2(ops, attn, x, p, rows_per_tile=rows_per_tile)
    assert torch.equal(x2, x2b) and torch.equal(q, qb)  # deterministic


@pytest.mark.parametrize("variant", ["no_q", "extra", "no_beta", "q1024", "q512", "q1536", "hidden1024", "hidden512", "hidden1536", "extra_q1024", "extra_q2048"])
def test_chain2_variants(variant):
    """no trailing projection, the latent skip as a second residual, LayerNorms without bias, odd / single chunk counts of the trailing
    projection (group B has one chunk fewer, or none) and of the hidden width (x2 lands in the other h buffer)."""
    import tests.test_chain_gpu as me
    from anemoi_core_amd import ops

    dtype, N = torch.bfloat16, 3000
    gen = torch.Generator().manual_seed(7)
    hd = {"hidden1024": 1024, "hidden512": 512, "hidden1536": 1536}.get(variant, 2048)
    old = me.HD
    me.HD = hd
    try:
        p = _params(gen, dtype, q_out={"no_q": 0, "extra": 0, "q1024": 1024, "q512": 512, "q1536": 1536, "extra_q1024": 1024}.get(variant, 2048), beta=variant != "no_beta")
    finally:
        me.HD = old
    attn = torch.randn(N, D, generator=gen).to(dtype)
    x = torch.randn(N, D, generator=gen).to(dtype)
    # (extra + a trailing projection: the decoder's k | v behind the last processor block - the projection reads LayerNorm(x2 + extra))
    extra = (3.0 * torch.randn(N, D, generator=gen)).to(dtype) if variant.startswith("extra") else None
    res = _run_chain2(ops, attn, x, p, extra=extra)
    ref2, refq = _reference(attn, x, p, dtype, extra=extra)
    if p["wq"] is None:
        assert isinstance(res, torch.Tensor)
        _close(res, ref2, variant)
    else:
        _close(res[0], ref2, variant)
        _close(res[1], refq, variant + " q")


@pytest.mark.parametrize("N,q_out,want_x", [(3000, 128, True), (40320, 128, False), (10242, 256, False), (5000, 384, True), (50001, 128, False)])
def test_chain2_narrow_trailing_projection(N, q_out, want_x):
    """a NARROW trailing projection (the decoder's node_data_extractor: LayerNorm + Linear(512, 84) zero-padded to 128 columns): only the first
    q_out / 128 waves of group A have a chunk, x2 is optionally not written at all; one round and several rounds of panels"""
    from anemoi_core_amd import ops

    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(N + q_out)
    p = _params(gen, dtype, q_out=q_out)
    attn = torch.randn(N, D, generator=gen).to(dtype)
    x = torch.randn(N, D, generator=gen).to(dtype)
    x2, q = _run_chain2(ops, attn, x, p, want_x_out=want_x)
    ref2, refq = _reference(attn, x, p, dtype)
    assert (x2 is not None) == want_x and tuple(q.shape) == (N, q_out)
    if want_x:
        _close(x2, ref2, f"x2 N={N}")
    _close(q, refq, f"narrow q N={N} q_out={q_out}")
    x2b, qb = _run_chain2(ops, attn, x, p, want_x_out=want_x)
    assert torch.equal(q, qb)


@pytest.mark.parametrize("N,hidden,with_extra", [(30000, 2048, False), (50001, 2048, True), (30000, 1024, False), (30000, 1536, False)])
def test_chain2_without_trailing_projection_over_several_panel_rounds(N, hidden, with_extra):
    """a mapper-style tail (no trailing projection) with several panels per workgroup: from the second panel on the attention rows come
    in during the step in which group A idles; odd and even chunk counts (x2 lands in either h buffer), with and without the latent skip"""
    import tests.test_chain_gpu as me
    from anemoi_core_amd import ops

    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(N + hidden)
    old = me.HD
    me.HD = hidden
    try:
        p = _params(gen, dtype, q_out=0)
    finally:
        me.HD = old
    attn = torch.randn(N, D, generator=gen).to(dtype)
    x = torch.randn(N, D, generator=gen).to(dtype)
    extra = (2.0 * torch.randn(N, D, generator=gen)).to(dtype) if with_extra else None
    got = _run_chain2(ops, attn, x, p, extra=extra)
    ref2, _ = _reference(attn, x, p, dtype, extra=extra)
    _close(got, ref2, f"x2 N={N} hidden={hidden}")
    assert torch.equal(got, _run_chain2(ops, attn, x, p, extra=extra))


# ------------------------------------------------------------------------------------------ mapper-side row chain (round 6, csrc/gt_rowchain.hip)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K,q_out,want_x", [(37, 64, 1024, True), (642, 192, 1024, False), (10242, 64, 1024, True), (40320, 192, 1024, False),
                                              (40320, 184, 1024, True), (5000, 512, 2048, True), (3000, 128, 512, True), (2049, 8, 1536, False),
                                              (1, 320, 1024, True), (30011, 96, 1024, True), (13000, 64, 512, True), (20000, 200, 1536, False)])
def test_row_chain_vs_fp32_restatement(dtype, N, K, q_out, want_x):
    """embedding -> LayerNorm -> fused projection of one mapper side in one launch against the fp32 restatement with the reference's rounding
    points (the embedded rows and the LayerNorm output in the model dtype): every row, ragged last panels, several panels per workgroup,
    input widths that are no multiple of the 128-column K group (zero-padded image), one / two / three / four projection chunks; above 12 288
    rows the pipelined kernel (two chunks: the schedule that splits the projection 3 : 1 between the wave groups; else the plain one)."""
    from anemoi_core_amd import ops

    gen = torch.Generator().manual_seed(N + K + q_out)
    r = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    x = r(N, K).to(dtype)
    we, be = (r(D, K) / max(K, 16) ** 0.5).to(dtype), (0.1 * r(D)).to(dtype)
    g, b = (1 + 0.2 * r(D)).to(dtype), (0.1 * r(D)).to(dtype)
    wq, bq = (r(q_out, D) / 22).to(dtype), (0.1 * r(q_out)).to(dtype)
    f = lambda t: t.float()  # noqa: E731
    rnd = lambda t: t.to(dtype).float()  # noqa: E731
    y_ref = rnd(F.linear(f(x), f(we), f(be)))
    q_ref = F.linear(rnd(F.layer_norm(y_ref, (D,), f(g), f(b), 1e-5)), f(wq), f(bq))
    d = lambda t: t.to(DEV)  # noqa: E731
    wqg, dq = ops.fold_layer_norm(d(wq), d(bq), d(g), d(b))
    vec = torch.cat([d(be).float(), dq]).to(dtype).contiguous()
    args = (d(x), ops.pack_embedding_frag(d(we)), ops.pack_weight_frag(wqg), vec, q_out, 1e-5)
    y, q = ops.gt_row_chain(*args, want_x_out=want_x)
    assert (y is not None) == want_x
    if want_x:
        _close(y, y_ref, f"y N={N} K={K}")
    _close(q, q_ref, f"q N={N} K={K}", tol=2.5e-2)
    y2, q2 = ops.gt_row_chain(*args, want_x_out=want_x)
    assert torch.equal(q, q2) and (not want_x or torch.equal(y, y2))  # deterministic


def test_decoder_with_tail_projection_equals_decoder_without():
    """node_data_extractor as the narrow trailing projection of the decoder's chain launch against its LayerNorm launch + GEMM (same model, same input)"""
    import anemoi_core_amd.layers.block as B
    import anemoi_core_amd.layers.mapper as M
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec
    from anemoi_core_amd.models.configs import make_data_indices, model_config

    g = build_synthetic_graph("o32", 4)
    torch.manual_seed(0)
    model = AnemoiModelEncProcDec(model_config=model_config("gt", 512, 2, 16, 8), data_indices=make_data_indices(6, 6),
                                  statistics={"data": None}, n_step_input=2, n_step_output=1, graph_data=g).eval().to(DEV).to(torch.bfloat16)
    xb = torch.randn(1, 2, 1, g.num_data, 6).to(DEV).to(torch.bfloat16)
    outs = {}
    saved = (M._TAIL_PROJ, B._LAYER_CHAIN_MIN_ROWS)
    for flag in (True, False):
        M._TAIL_PROJ, B._LAYER_CHAIN_MIN_ROWS = flag, 0  # (every block tail on the row-resident chain: the tail projection rides on it)
        try:
            with torch.no_grad():
                outs[flag] = model({"data": xb})["data"].float().cpu()
        finally:
            M._TAIL_PROJ, B._LAYER_CHAIN_MIN_ROWS = saved
    a, b = outs[True], outs[False]
    scale = float(b.abs().max())
    assert not torch.equal(a, b)  # two different paths really ran
    assert float((a - b).abs().max()) <= 3e-2 * scale and float((a - b).abs().mean()) <= 4e-3 * scale, (float((a - b).abs().max()), scale)


def test_mapper_with_row_chain_equals_mapper_without():
    """The O96 encoder / decoder mappers (GraphTransformerForwardMapper / BackwardMapper at 512 channels) with the row-chain launches against
    the embedding GEMM + LayerNorm-fold GEMM path of the same modules."""
    import anemoi_core_amd.layers.mapper as M
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec
    from anemoi_core_amd.models.configs import make_data_indices, model_config

    g = build_synthetic_graph("o32", 4)
    torch.manual_seed(0)
    model = AnemoiModelEncProcDec(model_config=model_config("gt", 512, 2, 16, 8), data_indices=make_data_indices(6, 6),
                                  statistics={"data": None}, n_step_input=2, n_step_output=1, graph_data=g).eval().to(DEV).to(torch.bfloat16)
    xb = torch.randn(1, 2, 1, g.num_data, 6).to(DEV).to(torch.bfloat16)
    outs = {}
    saved = (M._ROW_CHAIN, M._ROW_CHAIN_MIN_ROWS, M._ROW_CHAIN_GEMM_BAND)
    for flag in (True, False):
        M._ROW_CHAIN, M._ROW_CHAIN_MIN_ROWS, M._ROW_CHAIN_GEMM_BAND = flag, 0, (0, 0)
        try:
            with torch.no_grad():
                outs[flag] = model({"data": xb})["data"].float().cpu()
        finally:
            M._ROW_CHAIN, M._ROW_CHAIN_MIN_ROWS, M._ROW_CHAIN_GEMM_BAND = saved
    a, b = outs[True], outs[False]
    scale = float(b.abs().max())
    assert not torch.equal(a, b)  # two different paths really ran
    assert float((a - b).abs().max()) <= 3e-2 * scale and float((a - b).abs().mean()) <= 4e-3 * scale, (float((a - b).abs().max()), scale)


# ------------------------------------------------------------------------------------------ cluster chain (round 6, csrc/gt_cluster_chain.hip)
def _run_cluster(ops, attn, x, p, extra=None, ln_out=None, **kw):
    """the caller's side of anemoi_gt_cluster_chain_fwd: the operands of the layer chain (hidden = 2048)"""
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    dt = attn.dtype
    w1g, d1 = ops.fold_layer_norm(d(p["w1"]), d(p["b1"]), d(p["g1"]), d(p["be1"]))
    parts = [d(p["bp"]).float(), d1, d(p["b2"]).float()]
    wqg, qf = None, 0
    if p["wq"] is not None:
        wq_, dq = ops.fold_layer_norm(d(p["wq"]), d(p["bq"]), d(p["gq"]), d(p["beq"]))
        wqg, qf = ops.pack_weight_frag(wq_), p["wq"].shape[0]
        parts.append(dq)
    vec = torch.cat(parts).to(dt).contiguous()
    return ops.gt_cluster_chain(d(attn), d(x), ops.pack_weight_frag(d(p["wp"])), ops.pack_weight_frag(w1g), ops.pack_weight_frag(d(p["w2"])), vec,
                                p["w1"].shape[0], 1e-5, extra=d(extra), wqg=wqg, q_out_features=qf, lnq_eps=1e-5, ln_out=ln_out, **kw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N", [37, 642, 1281, 2561, 4095, 1, 12500])
def test_cluster_chain_vs_fp32_restatement(dtype, N):
    """the cluster chain (four CUs per 48-row panel, ONE exchange of fp32 partial sums) against the fp32 restatement of the block tail with the
    reference's rounding points: every row of x2 and of the trailing projection, ragged last panels, fewer panels than clusters, several
    panels per cluster (12 500 rows = 261 panels on 64 clusters: the exchange slots and counters are reused), repeated launches."""
    from anemoi_core_amd import ops

    gen = torch.Generator().manual_seed(N)
    p = _params(gen, dtype)
    attn = torch.randn(N, D, generator=gen).to(dtype)
    x = (2.0 * torch.randn(N, D, generator=gen) + 0.5).to(dtype)
    x2, q = _run_cluster(ops, attn, x, p)
    ref2, refq = _reference(attn, x, p, dtype)
    _close(x2, ref2, f"x2 N={N}")
    _close(q, refq, f"qkvs N={N}")
    for _ in range(3):  # the counters are monotonic across launches; results are deterministic (partials added in member order)
        x2b, qb = _run_cluster(ops, attn, x, p)
        assert torch.equal(x2, x2b) and torch.equal(q, qb)
    # next to the row-resident chain on the same operands: a few ulps of the output scale (different accumulation order)
    c2, cq = _run_chain2(ops, attn, x, p)
    e2, eq = (c2.float() - x2.float()).abs(), (cq.float() - q.float()).abs()
    assert float(e2.max()) <= 2e-2 * float(c2.float().abs().max()) and float(eq.max()) <= 3e-2 * float(cq.float().abs().max())


@pytest.mark.parametrize("variant", ["no_q", "extra", "no_beta", "q1024", "q512", "q1536", "ln_out", "ln_out_no_q"])
def test_cluster_chain_variants(variant):
    """no trailing projection, the latent skip as a second residual, LayerNorms without bias, fewer projection chunks than members, and the
    LayerNorm'd rows of x2 (without the affine part) as an extra output - what a sharded block sends to its halo peers."""
    from anemoi_core_amd import ops

    dtype, N = torch.bfloat16, 3000
    gen = torch.Generator().manual_seed(7)
    q_out = {"no_q": 0, "extra": 0, "q1024": 1024, "q512": 512, "q1536": 1536, "ln_out_no_q": 0}.get(variant, 2048)
    p = _params(gen, dtype, q_out=q_out, beta=variant != "no_beta")
    attn = torch.randn(N, D, generator=gen).to(dtype)
    x = torch.randn(N, D, generator=gen).to(dtype)
    extra = (3.0 * torch.randn(N, D, generator=gen)).to(dtype) if variant == "extra" else None
    ln_out = torch.empty(N, D, dtype=dtype, device=DEV) if variant.startswith("ln_out") else None
    res = _run_cluster(ops, attn, x, p, extra=extra, ln_out=ln_out)
    ref2, refq = _reference(attn, x, p, dtype, extra=extra)
    if p["wq"] is None:
        assert isinstance(res, torch.Tensor)
        _close(res, ref2, variant)
    else:
        _close(res[0], ref2, variant)
        _close(res[1], refq, variant + " q")
    if ln_out is not None:
        want = F.layer_norm(ref2, (D,), None, None, 1e-5)
        _close(ln_out, want, variant + " ln_out")


@pytest.mark.parametrize("q_split", [0, 1, 2, 4])
def test_cluster_chain_second_projection_destination(q_split):
    """the trailing projection's chunks from q_split on land in the head of a wider / taller buffer (a sharded block: k | v of the local rows in
    front of the halo rows): both destinations equal the one-destination launch, rows beyond the head untouched"""
    from anemoi_core_amd import ops

    dtype, N = torch.bfloat16, 1281
    gen = torch.Generator().manual_seed(9)
    p = _params(gen, dtype)
    attn, x = torch.randn(N, D, generator=gen).to(dtype), torch.randn(N, D, generator=gen).to(dtype)
    x2, q = _run_cluster(ops, attn, x, p)
    w2 = 2048 - 512 * q_split
    big = torch.full((N + 200, w2), 7.0, dtype=dtype, device=DEV)
    y2, q1 = _run_cluster(ops, attn, x, p, q_out2=big[:N], q_split=q_split)
    assert torch.equal(x2, y2)
    assert (q1 is None) == (q_split == 0) and (q1 is None or torch.equal(q1, q[:, :512 * q_split]))
    assert torch.equal(big[:N], q[:, 512 * q_split:]) and bool((big[N:] == 7.0).all())


def test_cluster_chain_in_a_hipgraph():
    """captured and replayed: the counters advance by device atomics, so a replay needs no host-side epoch"""
    from anemoi_core_amd import ops

    dtype, N = torch.bfloat16, 1281
    gen = torch.Generator().manual_seed(5)
    p = _params(gen, dtype)
    attn, x = torch.randn(N, D, generator=gen).to(dtype), torch.randn(N, D, generator=gen).to(dtype)
    want = _run_cluster(ops, attn, x, p)
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    w1g, d1 = ops.fold_layer_norm(d(p["w1"]), d(p["b1"]), d(p["g1"]), d(p["be1"]))
    wq_, dq = ops.fold_layer_norm(d(p["wq"]), d(p["bq"]), d(p["gq"]), d(p["beq"]))
    vec = torch.cat([d(p["bp"]).float(), d1, d(p["b2"]).float(), dq]).to(dtype).contiguous()
    ops_in = (d(attn), d(x), ops.pack_weight_frag(d(p["wp"])), ops.pack_weight_frag(w1g), ops.pack_weight_frag(d(p["w2"])), vec)
    wqg = ops.pack_weight_frag(wq_)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.gt_cluster_chain(*ops_in, HD, 1e-5, wqg=wqg, q_out_features=2048)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = [ops.gt_cluster_chain(*ops_in, HD, 1e-5, wqg=wqg, q_out_features=2048) for _ in range(4)]  # four dependent-in-order launches
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    for x2, q in out:
        assert torch.equal(x2, want[0]) and torch.equal(q, want[1])


def test_small_mesh_model_with_cluster_chain_equals_model_without():
    """AnemoiModelEncProcDec at 512 channels on a 642-node hidden mesh: the cluster chain (encoder tail -> processor blocks -> latent skip)
    against the LayerNorm-fold GEMM launches of round 5, and both against the fp32 CPU oracle within the full-model bf16 bound."""
    import anemoi_core_amd.layers.block as B
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec
    from anemoi_core_amd.models.configs import make_data_indices, model_config
    from oracle import gt_oracle as O

    g = build_synthetic_graph("o16", 3)
    torch.manual_seed(0)
    cfg = dict(kind="gt", num_channels=512, num_layers=3, num_heads=16, trainable=8, n_vars=6, n_step_input=2)
    model = AnemoiModelEncProcDec(model_config=model_config("gt", 512, 3, 16, 8), data_indices=make_data_indices(6, 6),
                                  statistics={"data": None}, n_step_input=2, n_step_output=1, graph_data=g).eval()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(1, 2, 1, g.num_data, 6)
    want = O.enc_proc_dec_forward(params, cfg, g, x)
    m = model.to(DEV).to(torch.bfloat16)
    xb = x.to(DEV).to(torch.bfloat16)
    outs = {}
    saved = B._CLUSTER_CHAIN
    for flag in (True, False):
        B._CLUSTER_CHAIN = flag
        try:
            with torch.no_grad():
                outs[flag] = m({"data": xb})["data"].float().cpu()
        finally:
            B._CLUSTER_CHAIN = saved
    a, b = outs[True], outs[False]
    scale = float(want.abs().max())
    assert not torch.equal(a, b)  # two different paths really ran
    assert float((a - b).abs().max()) <= 3e-2 * scale and float((a - b).abs().mean()) <= 4e-3 * scale, (float((a - b).abs().max()), scale)
    for name, y in (("cluster chain", a), ("launch-per-GEMM", b)):
        err = (y - want).abs()
        assert float(err.max()) <= 6e-2 * max(scale, 1.0) and float(err.mean()) <= 1e-2 * max(scale, 1.0), (name, float(err.max()), float(err.mean()), scale)
