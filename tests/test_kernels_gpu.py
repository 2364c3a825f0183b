"""GPU parity tests of the HIP kernels (through the C ABI) against the oracle.  Run with -m gpu on an MI355X.

Tolerances (stated per the north star):
  fp32 : atol 1e-4 (+ rtol 1e-5 for outputs much larger than 1) against the fp32 oracle — atol 1e-4, rtol 0 is the
         reference's own precedent for its fused kernel on O(1) data
         (models/tests/integration/triton/test_triton_gt.py:135-136); most checks pass at 2e-5.
  bf16 : inputs are rounded to bf16 first and the oracle is evaluated in fp32 on the rounded inputs; the
         kernel's result (fp32 accumulation, one final rounding to bf16) must match within
         atol 2e-2 * scale + rtol 2e-2, where scale = max |oracle| (bf16 has 8 mantissa bits: 2^-8 = 3.9e-3 per
         rounding; GEMMs with K<=2048 keep a few roundings of slack).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import gt_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from anemoi_core_amd import ops as _ops

    return _ops


def assert_close(got, want, dtype, what=""):
    got = got.float().cpu()
    want = want.float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    if dtype == torch.float32:
        atol, rtol = 1e-4, 1e-5  # the relative part only matters for |values| >> 1 (fp32 round-off of large sums)
    else:
        scale = float(want.abs().max()) if want.numel() else 1.0
        atol, rtol = 2e-2 * max(scale, 1e-3), 2e-2
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    assert bool((err <= bound).all()), f"{what}: max err {float(err.max()):.3e}, worst excess {float((err - bound).max()):.3e} ({dtype})"


def rand_graph(gen, n_src, n_dst, m, empty=()):
    src = torch.randint(0, n_src, (m,), generator=gen)
    dst = torch.randint(0, n_dst, (m,), generator=gen)
    for d in empty:
        dst = torch.where(dst == d, (dst + 1) % n_dst, dst)
    ei = torch.stack([src, dst])
    return ei[:, torch.sort(ei[1], stable=True)[1]].contiguous()


# ------------------------------------------------------------------------------------------ attention
def test_attention_golden_conv_cases(ops, golden):
    """The reference-generated conv vectors, incl. the reference kernel test's non power-of-two shapes."""
    for i, c in enumerate(golden("conv.pt")):
        n_src, n_dst = c["size"]
        H, C = c["q"].shape[1:]
        csc = ops.build_csc(c["edge_index"].to(DEV), (n_src, n_dst), edges_are_dst_sorted=False)
        e = c["e"][csc.perm.cpu()] if csc.perm is not None else c["e"]
        q, k, v, e = (t.reshape(t.shape[0], H * C).to(DEV) for t in (c["q"], c["k"], c["v"], e))
        out, lse = ops.gt_attention(q, k, v, e, csc, H, return_lse=True)
        assert_close(out.view(n_dst, H, C), c["out"], torch.float32, f"conv case {i}")
        assert_close(lse, O.gt_conv_lse(c["q"], c["k"], c["e"], c["edge_index"], c["size"]), torch.float32, f"lse case {i}")
        deg = torch.bincount(c["edge_index"][1], minlength=n_dst)
        if (deg == 0).any():
            assert float(out.cpu()[deg == 0].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,C", [(16, 32), (4, 16), (8, 16), (16, 64), (2, 32), (6, 6)])
def test_attention_materialised_vs_oracle(ops, dtype, H, C):
    gen = torch.Generator().manual_seed(H * 100 + C)
    n_src, n_dst, m = 300, 200, 1700
    ei = rand_graph(gen, n_src, n_dst, m, (0, 77))
    D = H * C
    q, k, v, e, add = (torch.randn(n, D, generator=gen).to(dtype) for n in (n_dst, n_src, n_src, m, n_dst))
    csc = ops.build_csc(ei.to(DEV), (n_src, n_dst))
    out, lse = ops.gt_attention(q.to(DEV), k.to(DEV), v.to(DEV), e.to(DEV), csc, H, addend=add.to(DEV), return_lse=True)
    f = lambda t, n: t.float().view(n, H, C)  # noqa: E731
    want = O.gt_conv(f(q, n_dst), f(k, n_src), f(v, n_src), f(e, m), ei, (n_src, n_dst)).reshape(n_dst, D) + add.float()
    assert_close(out, want, dtype, "attention")
    assert_close(lse, O.gt_conv_lse(f(q, n_dst), f(k, n_src), f(e, m), ei, (n_src, n_dst)), torch.float32 if dtype == torch.float32 else dtype, "lse")
    # no edge term at all (e = None)
    out0 = ops.gt_attention(q.to(DEV), k.to(DEV), v.to(DEV), None, csc, H)
    want0 = O.gt_conv(f(q, n_dst), f(k, n_src), f(v, n_src), torch.zeros(m, H, C), ei, (n_src, n_dst)).reshape(n_dst, D)
    assert_close(out0, want0, dtype, "attention without edges")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,C,fe", [(16, 32, 11), (4, 16, 11), (8, 16, 3), (16, 32, 15), (16, 64, 7), (6, 6, 5)])
def test_attention_fused_edge_vs_oracle(ops, dtype, H, C, fe):
    gen = torch.Generator().manual_seed(7 * H + C + fe)
    n_src, n_dst, m = 260, 310, 2100
    ei = rand_graph(gen, n_src, n_dst, m, (5,))
    D = H * C
    q, k, v, add = (torch.randn(n, D, generator=gen).to(dtype) for n in (n_dst, n_src, n_src, n_dst))
    ea = torch.randn(m, fe, generator=gen).to(dtype)
    w = (torch.randn(D, fe, generator=gen) / math.sqrt(fe)).to(dtype)
    b = (0.1 * torch.randn(D, generator=gen)).to(dtype)
    csc = ops.build_csc(ei.to(DEV), (n_src, n_dst))
    feat = ops.pack_edge_features(ea.to(DEV))
    assert feat.shape == (m, ops.edge_feature_pad(fe)) and float(feat[:, fe].min()) == 1.0
    wp = ops.pack_edge_weights(w.to(DEV), b.to(DEV))
    assert wp.shape == (D, ops.edge_feature_pad(fe)) and wp.dtype == torch.float32
    out, lse = ops.gt_attention_fused_edge(q.to(DEV), k.to(DEV), v.to(DEV), feat, wp, csc, H, addend=add.to(DEV), return_lse=True)
    e = F.linear(ea.float(), w.float(), b.float()).view(m, H, C)  # fp32 E from the rounded inputs (never rounded itself)
    f = lambda t, n: t.float().view(n, H, C)  # noqa: E731
    want = O.gt_conv(f(q, n_dst), f(k, n_src), f(v, n_src), e, ei, (n_src, n_dst)).reshape(n_dst, D) + add.float()
    assert_close(out, want, dtype, "fused-edge attention")
    assert_close(lse, O.gt_conv_lse(f(q, n_dst), f(k, n_src), e, ei, (n_src, n_dst)), dtype, "fused-edge lse")
    # no bias
    out_nb = ops.gt_attention_fused_edge(q.to(DEV), k.to(DEV), v.to(DEV), feat, ops.pack_edge_weights(w.to(DEV), None), csc, H)
    e_nb = F.linear(ea.float(), w.float()).view(m, H, C)
    assert_close(out_nb, O.gt_conv(f(q, n_dst), f(k, n_src), f(v, n_src), e_nb, ei, (n_src, n_dst)).reshape(n_dst, D), dtype, "fused-edge, no bias")


def test_attention_online_softmax_rescale(ops):
    """Large, growing scores force the running-max rescale branch on every edge (guide §5.4 rule 26)."""
    H, C, n = 16, 32, 64
    D = H * C
    deg = 40
    src = torch.arange(deg).repeat(n) % n
    dst = torch.arange(n).repeat_interleave(deg)
    ei = torch.stack([src, dst])
    gen = torch.Generator().manual_seed(3)
    q = 4.0 * torch.randn(n, D, generator=gen)
    k = 4.0 * torch.randn(n, D, generator=gen) * torch.linspace(0.1, 3.0, n).view(n, 1)  # later sources score higher
    v = torch.randn(n, D, generator=gen)
    e = torch.randn(n * deg, D, generator=gen)
    csc = ops.build_csc(ei.to(DEV), (n, n))
    out = ops.gt_attention(q.to(DEV), k.to(DEV), v.to(DEV), e.to(DEV), csc, H)
    want = O.gt_conv(q.view(n, H, C), k.view(n, H, C), v.view(n, H, C), e.view(-1, H, C), ei, (n, n)).reshape(n, D)
    assert_close(out, want, torch.float32, "rescale")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_fused_edge_rescale_branches(ops, dtype):
    """Force BOTH branches of the deferred running-max update of the fused kernel (guide §5.4 rule 26): scores that
    grow steeply along the edge list (rescale taken repeatedly), scores that stay within the threshold (never taken
    after the first edge), and one spiked edge in the middle of a long list."""
    H, C, n, fe = 16, 32, 48, 11
    D = H * C
    deg = 70  # > 64: also crosses the 64-edge chunk boundary
    src = (torch.arange(deg).repeat(n) * 7 + torch.arange(n).repeat_interleave(deg)) % n
    dst = torch.arange(n).repeat_interleave(deg)
    ei = torch.stack([src, dst])
    gen = torch.Generator().manual_seed(11)
    for mode in ("growing", "flat", "spike"):
        q = torch.randn(n, D, generator=gen)
        k = torch.randn(n, D, generator=gen)
        if mode == "growing":
            q, k = 3.0 * q, 3.0 * k * torch.linspace(0.05, 4.0, n).view(n, 1)
        elif mode == "flat":
            q, k = 0.1 * q, 0.1 * k
        v = torch.randn(n, D, generator=gen)
        ea = torch.randn(n * deg, fe, generator=gen)
        if mode == "spike":
            ea[deg // 2::deg] *= 25.0
        w = torch.randn(D, fe, generator=gen) / math.sqrt(fe)
        b = 0.1 * torch.randn(D, generator=gen)
        q, k, v, ea, w, b = (t.to(dtype) for t in (q, k, v, ea, w, b))
        csc = ops.build_csc(ei.to(DEV), (n, n))
        out, lse = ops.gt_attention_fused_edge(q.to(DEV), k.to(DEV), v.to(DEV), ops.pack_edge_features(ea.to(DEV)),
                                               ops.pack_edge_weights(w.to(DEV), b.to(DEV)), csc, H, return_lse=True)
        e = F.linear(ea.float(), w.float(), b.float()).view(-1, H, C)
        f = lambda t: t.float().view(n, H, C)  # noqa: E731
        assert_close(out, O.gt_conv(f(q), f(k), f(v), e, ei, (n, n)).reshape(n, D), dtype, f"fused rescale [{mode}]")
        assert_close(lse, O.gt_conv_lse(f(q), f(k), e, ei, (n, n)), dtype, f"fused rescale lse [{mode}]")


def test_attention_processing_order_changes_nothing(ops, monkeypatch):
    """The locality-preserving WORK order of the fused attention (ops.processing_order: BFS over the short edges inside every
    XCD's index range, res-6 mesh) is a permutation, differs from the identity, and the kernel's output and log-sum-exp are
    bit-identical with and without it, also with the write-through output stores (ANEMOI_ATTN_OUT_WT=1 is read at first launch:
    covered by the alternative-build A/B runs, here the default)."""
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph

    g = build_synthetic_graph("o8", 6)
    n, H, D, fe = g.num_hidden, 16, 512, 11
    ei = torch.from_numpy(g.proc_edge_index).to(DEV)
    csc = ops.build_csc(ei, (n, n))
    order = ops.processing_order(csc)
    assert order is not None and order.dtype == torch.int32 and order.shape == (n,)
    assert torch.equal(torch.sort(order.long())[0], torch.arange(n, device=DEV)) and not torch.equal(order.long(), torch.arange(n, device=DEV))
    per = (n + 7) // 8  # every XCD keeps its own index range
    assert torch.equal(order.long() // per, torch.arange(n, device=DEV) // per)
    gen = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(n, D, generator=gen).to(torch.bfloat16).to(DEV) for _ in range(3))
    feat = ops.pack_edge_features(torch.randn(ei.shape[1], fe, generator=gen).to(DEV))
    w = ops.pack_edge_weights((torch.randn(D, fe, generator=gen) / 3).to(torch.bfloat16).to(DEV), torch.zeros(D, dtype=torch.bfloat16, device=DEV))
    plain = ops.gt_attention_fused_edge(q, k, v, feat, w, csc, H, return_lse=True)
    import dataclasses

    ordered = ops.gt_attention_fused_edge(q, k, v, feat, w, dataclasses.replace(csc, order=order), H, return_lse=True)
    assert torch.equal(plain[0], ordered[0]) and torch.equal(plain[1], ordered[1])
    small = ops.build_csc(ei[:, (ei[0] < 1000) & (ei[1] < 1000)], (1000, 1000))
    assert ops.processing_order(small) is None  # nothing to gain: the rows fit the L2s
    bip = ops.build_csc(torch.from_numpy(g.enc_edge_index).to(DEV), (g.num_data, n))
    assert ops.processing_order(bip) is None  # bipartite: no destination-destination adjacency to walk


def test_attention_strided_views(ops):
    """q/k/v as column slices of one fused [N, 4D] projection buffer (leading dimension 4D)."""
    gen = torch.Generator().manual_seed(5)
    H, C, n, m = 16, 32, 150, 900
    D = H * C
    ei = rand_graph(gen, n, n, m)
    buf = torch.randn(n, 4 * D, generator=gen).to(torch.bfloat16).to(DEV)
    e = torch.randn(m, D, generator=gen).to(torch.bfloat16).to(DEV)
    csc = ops.build_csc(ei.to(DEV), (n, n))
    q, k, v = buf[:, :D], buf[:, D:2 * D], buf[:, 2 * D:3 * D]
    out = ops.gt_attention(q, k, v, e, csc, H, addend=buf[:, 3 * D:])
    f = lambda t: t.float().cpu().contiguous().view(-1, H, C)  # noqa: E731
    want = O.gt_conv(f(q), f(k), f(v), f(e), ei, (n, n)).reshape(n, D) + buf[:, 3 * D:].float().cpu()
    assert_close(out, want, torch.bfloat16, "strided")


# ------------------------------------------------------------------------------------------ layer norm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [512, 64, 100, 33, 1024, 2048, 32])
def test_layernorm(ops, dtype, D):
    gen = torch.Generator().manual_seed(D)
    x = (2.0 * torch.randn(257, D, generator=gen) + 0.5).to(dtype)
    g = (1 + 0.3 * torch.randn(D, generator=gen)).to(dtype)
    b = (0.2 * torch.randn(D, generator=gen)).to(dtype)
    y = ops.layer_norm(x.to(DEV), g.to(DEV), b.to(DEV))
    assert_close(y, F.layer_norm(x.float(), (D,), g.float(), b.float()), dtype, "layernorm")
    y_nb = ops.layer_norm(x.to(DEV), g.to(DEV), None)
    assert_close(y_nb, F.layer_norm(x.float(), (D,), g.float(), None), dtype, "layernorm, no bias")
    # residual fused, row-strided input (a column slab of a wider buffer) and out= into the head of a larger buffer
    res = torch.randn(257, D, generator=gen).to(dtype)
    assert_close(ops.layer_norm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, res.to(DEV)),
                 F.layer_norm(x.float(), (D,), g.float(), b.float()) + res.float(), dtype, "layernorm + residual")
    if D % 8 == 0:
        wide = torch.randn(257, D + 16, generator=gen).to(dtype).to(DEV)
        assert_close(ops.layer_norm(wide[:, 8:8 + D], g.to(DEV), b.to(DEV)),
                     F.layer_norm(wide[:, 8:8 + D].float().cpu(), (D,), g.float(), b.float()), dtype, "layernorm, strided rows")
    buf = torch.zeros(300, D, dtype=dtype, device=DEV)
    got = ops.layer_norm(x.to(DEV), g.to(DEV), b.to(DEV), out=buf[:257])
    assert got.data_ptr() == buf.data_ptr() and torch.equal(got, y) and float(buf[257:].abs().max()) == 0.0
    # per-head norm over C (qk_norm): 3-D input
    if D <= 64:
        x3 = x.view(257, 1, D).expand(257, 4, D).contiguous()
        assert_close(ops.layer_norm(x3.to(DEV), g.to(DEV), None), F.layer_norm(x3.float(), (D,), g.float(), None), dtype, "3-D")


# ------------------------------------------------------------------------------------------ linear
def _lin_ref(x, w, b=None, act=None, res=None, x2=None, g1=None, i1=None, g2=None, i2=None):
    a = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
    y = F.linear(a, w.float(), None if b is None else b.float())
    if g1 is not None:
        y = y + g1.float()[i1.long()]
    if g2 is not None:
        y = y + g2.float()[i2.long()]
    if act == "gelu":
        y = F.gelu(y)
    if res is not None:
        y = y + res.float()
    return y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K,O", [(300, 512, 512), (1000, 512, 2048), (257, 2048, 512), (129, 64, 128), (70, 20, 64), (50, 11, 7), (333, 512, 100), (5, 64, 64),
                                   (1300, 512, 2048), (2100, 2048, 512), (1111, 512, 512), (1030, 64, 100), (1500, 72, 512),
                                   (4200, 512, 2048), (3000, 192, 3072),  # > 256 tiles: several tiles per persistent workgroup
                                   (10242, 512, 512),  # 192-row tile variant (216 tiles instead of 164 of 256 rows)
                                   (10242, 512, 2048), (10242, 192, 2048)])  # 320x256 big tile, one per CU, + 2 tail rows on the VALU
def test_linear_epilogues(ops, dtype, N, K, O):
    gen = torch.Generator().manual_seed(N + K + O)
    x = torch.randn(N, K, generator=gen).to(dtype)
    w = (torch.randn(O, K, generator=gen) / math.sqrt(K)).to(dtype)
    b = (0.1 * torch.randn(O, generator=gen)).to(dtype)
    res = torch.randn(N, O, generator=gen).to(dtype)
    d = lambda t: t.to(DEV)  # noqa: E731
    assert_close(ops.linear(d(x), d(w)), _lin_ref(x, w), dtype, "plain")
    assert_close(ops.linear(d(x), d(w), d(b)), _lin_ref(x, w, b), dtype, "bias")
    assert_close(ops.linear(d(x), d(w), d(b), act="gelu"), _lin_ref(x, w, b, "gelu"), dtype, "bias+gelu")
    assert_close(ops.linear(d(x), d(w), d(b), residual=d(res)), _lin_ref(x, w, b, None, res), dtype, "bias+residual")
    assert_close(ops.linear(d(x), d(w), d(b), act="gelu", residual=d(res)), _lin_ref(x, w, b, "gelu", res), dtype, "bias+gelu+residual")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,K1,K2,O", [(400, 512, 512, 512), (90, 32, 32, 32), (200, 128, 64, 256), (1400, 512, 512, 512), (1200, 128, 64, 256),
                                       (10242, 256, 256, 2048)])  # big tile + tail rows with the K-concat / gather-add epilogue
def test_linear_concat_and_gather(ops, dtype, N, K1, K2, O):
    gen = torch.Generator().manual_seed(N + K1)
    x, x2 = torch.randn(N, K1, generator=gen).to(dtype), torch.randn(N, K2, generator=gen).to(dtype)
    w = (torch.randn(O, K1 + K2, generator=gen) / math.sqrt(K1 + K2)).to(dtype)
    b = (0.1 * torch.randn(O, generator=gen)).to(dtype)
    g1, g2 = torch.randn(37, O, generator=gen).to(dtype), torch.randn(53, O, generator=gen).to(dtype)
    i1 = torch.randint(0, 37, (N,), generator=gen, dtype=torch.int32)
    i2 = torch.randint(0, 53, (N,), generator=gen, dtype=torch.int32)
    d = lambda t: t.to(DEV)  # noqa: E731
    assert_close(ops.linear(d(x), d(w), d(b), x2=d(x2)), _lin_ref(x, w, b, x2=x2), dtype, "concat")
    got = ops.linear(d(x), d(w[:, :K1].contiguous()), d(b), act="gelu", g1=d(g1), idx1=d(i1), g2=d(g2), idx2=d(i2))
    assert_close(got, _lin_ref(x, w[:, :K1], b, "gelu", None, None, g1, i1, g2, i2), dtype, "gather-add")


def test_linear_transpose_detecting(ops):
    """A = I with an ASYMMETRIC weight: catches a transposed MFMA C/D mapping (guide §3)."""
    K = O = 128
    x = torch.eye(K, dtype=torch.bfloat16)
    w = (torch.arange(O * K, dtype=torch.float32).view(O, K) % 61 - 30).to(torch.bfloat16)
    y = ops.linear(x.to(DEV), w.to(DEV))
    assert torch.equal(y.float().cpu(), w.float().t())


# ------------------------------------------------------------------------------------------ GraphConv pieces
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [512, 32, 100])
def test_edge_ln_residual_segment_sum(ops, dtype, D):
    gen = torch.Generator().manual_seed(D)
    n_dst, m = 90, 700
    ei = rand_graph(gen, 50, n_dst, m, (3, 4))
    z, e_old = torch.randn(m, D, generator=gen).to(dtype), torch.randn(m, D, generator=gen).to(dtype)
    g, b = (1 + 0.2 * torch.randn(D, generator=gen)).to(dtype), (0.1 * torch.randn(D, generator=gen)).to(dtype)
    csc = ops.build_csc(ei.to(DEV), (50, n_dst))
    e_new, agg = ops.edge_ln_residual_segment_sum(z.to(DEV), e_old.to(DEV), g.to(DEV), b.to(DEV), 1e-5, csc)
    want_e = F.layer_norm(z.float(), (D,), g.float(), b.float()) + e_old.float()
    assert_close(e_new, want_e, dtype, "e_new")
    want_agg = torch.zeros(n_dst, D).index_add_(0, ei[1], e_new.float().cpu())  # sum of what was stored
    assert_close(agg, want_agg, dtype, "agg")
    e2, agg2 = ops.edge_ln_residual_segment_sum(z.to(DEV), e_old.to(DEV), None, None, 1e-5, csc)
    assert_close(e2, z.float() + e_old.float(), dtype, "no LN")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_beta", [True, False])
def test_edge_ln_residual_segment_sum_ragged_512(ops, dtype, with_beta):
    """The 512-channel 16-bit kernel works on two in-edges of a destination at a time (32 lanes each) and prefetches the next two:
    segments of 0, 1, 2, 3 (odd tails), 5, 8 and 37 edges, the first and the last destination empty."""
    D = 512
    gen = torch.Generator().manual_seed(7)
    deg = torch.tensor([0, 1, 2, 3, 5, 8, 37, 0, 4, 1, 0, 7, 2, 0])
    n_dst, m = deg.numel(), int(deg.sum())
    dst = torch.repeat_interleave(torch.arange(n_dst), deg)
    ei = torch.stack([torch.randint(0, 23, (m,), generator=gen), dst])
    z, e_old = (3 * torch.randn(m, D, generator=gen) + 1).to(dtype), torch.randn(m, D, generator=gen).to(dtype)
    g = (1 + 0.2 * torch.randn(D, generator=gen)).to(dtype)
    b = (0.1 * torch.randn(D, generator=gen)).to(dtype) if with_beta else None
    csc = ops.build_csc(ei.to(DEV), (23, n_dst))
    e_new, agg = ops.edge_ln_residual_segment_sum(z.to(DEV), e_old.to(DEV), g.to(DEV), None if b is None else b.to(DEV), 1e-5, csc)
    want_e = F.layer_norm(z.float(), (D,), g.float(), None if b is None else b.float()) + e_old.float()
    assert_close(e_new, want_e, dtype, "e_new")
    want_agg = torch.zeros(n_dst, D).index_add_(0, dst, e_new.float().cpu())  # sum of what was stored, fp32
    assert torch.equal(agg.float().cpu()[deg == 0], torch.zeros(int((deg == 0).sum()), D)), "empty segments must give zero rows"
    assert torch.equal(agg.float().cpu()[deg == 1], e_new.float().cpu()[(deg[dst] == 1)]), "a one-edge segment is that edge's row"
    assert_close(agg, want_agg, dtype, "agg")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_rows(ops, dtype):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(500, 512, generator=gen).to(dtype)
    idx = torch.randint(0, 500, (333,), generator=gen, dtype=torch.int32)
    assert torch.equal(ops.gather_rows(x.to(DEV), idx.to(DEV)).cpu(), x[idx.long()])
    xs = torch.randn(40, 33, generator=gen).to(dtype)
    assert torch.equal(ops.gather_rows(xs.to(DEV), idx.to(DEV) % 40).cpu(), xs[(idx % 40).long()])


# ------------------------------------------------------------------------------------------ reference-op mirror
def test_custom_op_matches_reference_signature(ops):
    gen = torch.Generator().manual_seed(9)
    n_src, n_dst, H, C, m = 40, 30, 4, 16, 200
    ei = rand_graph(gen, n_src, n_dst, m)
    q, k, v, e = (torch.randn(n, H, C, generator=gen) for n in (n_dst, n_src, n_src, m))
    colptr = torch.zeros(n_dst + 1, dtype=torch.long)
    colptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n_dst), 0)
    row = ei[0]
    rowptr = torch.zeros(n_src + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n_src), 0)
    edge_ids = torch.argsort(ei[0], stable=True)
    d = lambda t: t.to(DEV)  # noqa: E731
    out = ops.graph_transformer_attention_conv(d(q), d(k), d(v), d(e), (d(row), d(colptr)), (d(rowptr), d(edge_ids), d(ei[1])))
    assert out.shape == (n_dst, H, C)
    assert_close(out, O.gt_conv(q, k, v, e, ei, (n_src, n_dst)), torch.float32, "custom op")
    torch.library.opcheck(torch.ops.anemoi_amd.graph_transformer_attention.default,
                          (d(q), d(k), d(v), d(e), d(row), d(colptr), d(rowptr), d(edge_ids), d(ei[1])),
                          test_utils=("test_schema", "test_faketensor"))


def test_cpu_tensor_fails_loudly(ops):
    x = torch.randn(4, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layer_norm(x, torch.ones(64), torch.zeros(64))


# ------------------------------------------------------------------------------------------ LayerNorm fold
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
# 320-row tiles + 2 tail rows / whole tiles / partial last tile / tail of 10 / 320 k + 162 and 320 k + 192: <= 32 rows beyond a multiple
# of 160 but NOT of 320 (the producer used to peel them without strip sums while the consumer expected sums)
@pytest.mark.parametrize("N", [10242, 640, 4000, 330, 5282, 5312, 642, 1469, 2562])  # the last three: small-tile consumers (< 4096 rows)
def test_layernorm_folded_into_neighbouring_gemms(ops, dtype, N):
    """anemoi_linear_stats_fwd + anemoi_linear_lnfold_fwd: y = h W2^T + b2 + res with row statistics, then
    act(LN(y) W1^T + b1) from the raw y — against fp32 torch, and the producer's y equal to the plain GEMM's to rounding."""
    gen = torch.Generator().manual_seed(N)
    D, Hd = 512, 2048
    h = torch.randn(N, Hd, generator=gen).to(dtype)
    w2, b2 = (torch.randn(D, Hd, generator=gen) / 45).to(dtype), (0.1 * torch.randn(D, generator=gen)).to(dtype)
    res = (2.0 * torch.randn(N, D, generator=gen) + 0.5).to(dtype)
    gamma, beta = (1 + 0.2 * torch.randn(D, generator=gen)).to(dtype), (0.1 * torch.randn(D, generator=gen)).to(dtype)
    w1, b1 = (torch.randn(Hd, D, generator=gen) / 22).to(dtype), (0.1 * torch.randn(Hd, generator=gen)).to(dtype)
    d = lambda t: t.to(DEV)  # noqa: E731
    r = ops.linear_with_row_stats(d(h), d(w2), d(b2), d(res))
    assert r is not None
    y, stats = r
    plain = ops.linear(d(h), d(w2), d(b2), residual=d(res)).float()  # may be another kernel (K summed in another order): 1 ulp
    assert float((y.float() - plain).abs().max()) <= 1e-2 * float(plain.abs().max())
    yf = y.float()
    # a few rows beyond a multiple of the 320-row tile (the icosphere's "+ 2") are computed a column per wave and carry NO strip
    # sums (include/anemoi_hip.h): the folding consumer takes the statistics of such rows from the rows themselves - ONE rule
    # (n_rows % 320 <= 32) on both sides
    nst = N - N % 320 if 0 < N % 320 <= 32 and N > 320 else N
    s = stats[:nst].sum(1)
    assert float((s[:, 0] - yf[:nst].sum(1)).abs().max()) < 1e-3
    assert float((s[:, 1] - (yf[:nst] * yf[:nst]).sum(1)).abs().max()) < 1e-3 * float((yf * yf).sum(1).max())
    ws = (w1.float() * gamma.float()).to(dtype)
    c, dd = ws.float().sum(1).contiguous(), (w1.float() @ beta.float() + b1.float()).contiguous()
    for act in (None, "gelu"):
        out = ops.linear_ln_folded(y, d(ws), d(c), d(dd), stats, 1e-5, act)
        assert out is not None
        ref = F.linear(F.layer_norm(yf.cpu(), (D,), gamma.float(), beta.float()), w1.float(), b1.float())
        ref = F.gelu(ref) if act else ref
        assert_close(out, ref, dtype, f"fold act={act} N={N}")
        assert torch.equal(out, ops.linear_ln_folded(y, d(ws), d(c), d(dd), stats, 1e-5, act))  # deterministic


@pytest.mark.parametrize("n,o,i", [(10242, 2048, 512), (10242, 512, 2048), (1000, 24, 16), (4099, 520, 136), (63, 128, 128),
                                   (257, 8, 8), (40320, 512, 184), (81840, 512, 16)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wgrad_transpose_read_kernel(n, o, i, dtype):
    """dW = dZ^T X through the LDS transpose-read kernel == fp32 matmul of the same 16-bit operands; non-symmetric random
    operands (an operand / output transpose cannot pass), ragged row counts and widths, run-to-run bit-identical."""
    from anemoi_core_amd import ops

    g = torch.Generator(device="cuda").manual_seed(n + o + i)
    dz = (torch.randn(n, o, device="cuda", generator=g) * 0.5).to(dtype)
    x = torch.randn(n, i, device="cuda", generator=g).to(dtype)
    got, db = ops.linear_wgrad(dz, x, with_bias_grad=True)
    ref = dz.float().t() @ x.float()
    ref_b = dz.float().sum(0)
    assert db.shape == (o,) and db.dtype == dtype
    assert float((db.float() - ref_b).abs().max()) <= 6e-3 * float(ref_b.abs().max()) + 1e-3
    scale = float(ref.abs().max())
    assert got.shape == (o, i) and got.dtype == dtype
    assert float((got.float() - ref).abs().max()) <= 6e-3 * scale + 1e-3
    assert torch.equal(got, ops.linear_wgrad(dz, x))  # and without the bias-gradient column sums
    # row-strided views (a column slab of a wider buffer) are taken in place
    wide = torch.randn(n, o + 16, device="cuda", generator=g).to(dtype)
    got2 = ops.linear_wgrad(wide[:, 8:8 + o], x)
    ref2 = wide[:, 8:8 + o].float().t() @ x.float()
    assert float((got2.float() - ref2).abs().max()) <= 6e-3 * float(ref2.abs().max()) + 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V_out", [37, 84])  # scalar path / four columns per thread
def test_assemble_output_residual_columns(ops, dtype, V_out):
    """out[n, v] = x_out[n, v] + x_skip[n, col_map[v]] where col_map[v] >= 0 (reference _assemble_output: index_add_ of the
    skip connection onto the prognostic columns, encoder_processor_decoder.py:145-163); x_skip may be a row-strided view."""
    g = torch.Generator().manual_seed(3)
    N, V_in = 1000, 53
    x_out = torch.randn(N, V_out, generator=g).to(dtype)
    wide = torch.randn(N, V_in + 5, generator=g).to(dtype)
    x_skip = wide[:, 2:2 + V_in]
    col_map = torch.full((V_out,), -1, dtype=torch.int32)
    picks = torch.randperm(V_out, generator=g)[:20]
    col_map[picks] = torch.randint(0, V_in, (20,), generator=g, dtype=torch.int32)
    got = ops.assemble_output(x_out.to(DEV), wide.to(DEV)[:, 2:2 + V_in], col_map.to(DEV))
    ref = x_out.clone()  # the reference adds in the output dtype: cast(x_out) + skip, rounded once more
    ref[:, picks] = (x_out[:, picks].float() + x_skip[:, col_map[picks].long()].float()).to(dtype)
    assert got.dtype == dtype and torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,N,V,A,W", [(2, 1000, 84, 12, 192), (2, 333, 84, 12, 180), (3, 257, 7, 5, 32), (1, 64, 4, 0, 8)])
def test_assemble_input_matches_permute_and_cat(ops, dtype, T, N, V, A, W):
    """[x[0] | ... | x[T-1] | attrs | zeros] per node == the reference's "(batch ensemble grid) (time vars)" rearrangement + cat
    (encoder_processor_decoder.py:98-143), bit-exact (pure data movement), with 8-byte and scalar paths."""
    g = torch.Generator().manual_seed(T * N + V)
    x5 = torch.randn(1, T, 1, N, V, generator=g).to(dtype)
    attrs = torch.randn(N, A, generator=g).to(dtype) if A else None
    flat = x5.permute(0, 2, 3, 1, 4).reshape(N, T * V)
    ref = torch.cat([flat] + ([attrs] if A else []) + [torch.zeros(N, W - T * V - A, dtype=dtype)], 1)
    got = ops.assemble_input(x5.to(DEV)[0, :, 0], None if attrs is None else attrs.to(DEV), W)
    assert got.shape == (N, W) and torch.equal(got.cpu(), ref)
