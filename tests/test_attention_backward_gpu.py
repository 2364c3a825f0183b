"""Backward of the graph-transformer attention op (scope row f1: reference triton/gt.py:182-376, 447-556), through the C ABI
and through the autograd registration of the op mirror, against torch autograd of the fp32 oracle.

Tolerances: fp32 atol 1e-4 (+ rtol 1e-5), the reference's own precedent for its fused kernels
(models/tests/integration/triton/test_triton_gt.py:135-136, 168-184 uses atol 1e-3..1e-2 for the gradients); bf16: the
inputs are rounded first, the oracle differentiates in fp32 on the rounded inputs, the result must match within
2e-2 * max|grad| + 2e-2 |grad|."""
import pytest
import torch

from oracle import gt_oracle as O
from tests.test_kernels_gpu import DEV, assert_close, rand_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from anemoi_core_amd import ops as _ops

    return _ops


def _oracle_grads(q, k, v, e, ei, size, g):
    qs, ks, vs, es = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v, e))
    out = O.gt_conv(qs, ks, vs, es, ei, size)
    out.backward(g.float())
    return out.detach(), qs.grad, ks.grad, vs.grad, es.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,C", [(16, 32), (4, 16), (8, 64), (2, 32), (3, 7), (2, 5), (1, 64)])
def test_backward_c_abi_vs_oracle_autograd(ops, dtype, H, C):
    gen = torch.Generator().manual_seed(100 * H + C)
    n_src, n_dst, m = 70, 50, 400
    ei = rand_graph(gen, n_src, n_dst, m, empty=(3, 7, 20))  # zero-in-degree destinations; some sources unused
    q, k, v, e, g = (torch.randn(n, H, C, generator=gen).to(dtype) for n in (n_dst, n_src, n_src, m, n_dst))
    out_ref, dq, dk, dv, de = _oracle_grads(q, k, v, e, ei, (n_src, n_dst), g)
    csc = ops.build_csc(ei.to(DEV), (n_src, n_dst))
    flat = lambda t: t.reshape(t.shape[0], H * C).to(DEV)  # noqa: E731
    out, lse = ops.gt_attention(flat(q), flat(k), flat(v), flat(e), csc, H, return_lse=True)
    got = ops.gt_attention_backward(flat(g), flat(q), flat(k), flat(v), flat(e), out, lse, csc, ops.build_reverse_csr(csc), H)
    for name, a, b in zip(("dq", "dk", "dv", "de"), got, (dq, dk, dv, de)):
        assert_close(a.view(-1, H, C), b, dtype, f"{name} H={H} C={C}")
    # rows without edges get exact zeros
    assert float(got[0].view(n_dst, -1)[[3, 7, 20]].abs().max()) == 0.0
    unused = torch.ones(n_src, dtype=torch.bool)
    unused[ei[0]] = False
    if bool(unused.any()):
        assert float(got[1][unused.to(DEV)].abs().max()) == 0.0 and float(got[2][unused.to(DEV)].abs().max()) == 0.0


def test_backward_strided_inputs_and_determinism(ops):
    """q|k|v as column slices of one fused projection output (what the blocks produce) and run-to-run bit equality."""
    gen = torch.Generator().manual_seed(5)
    n, H, C, m = 300, 16, 32, 2400
    D = H * C
    ei = rand_graph(gen, n, n, m)
    qkv = torch.randn(n, 3 * D, generator=gen).to(torch.bfloat16).to(DEV)
    e = torch.randn(m, D, generator=gen).to(torch.bfloat16).to(DEV)
    g = torch.randn(n, D, generator=gen).to(torch.bfloat16).to(DEV)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    csc = ops.build_csc(ei.to(DEV), (n, n))
    rev = ops.build_reverse_csr(csc)
    out, lse = ops.gt_attention(q, k, v, e, csc, H, return_lse=True)
    a = ops.gt_attention_backward(g, q, k, v, e, out, lse, csc, rev, H)
    b = ops.gt_attention_backward(g, q.contiguous(), k.contiguous(), v.contiguous(), e, out, lse, csc, rev, H)
    c = ops.gt_attention_backward(g, q, k, v, e, out, lse, csc, rev, H)
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)


def test_autograd_through_the_op_mirror(ops):
    """loss.backward() through anemoi_amd::graph_transformer_attention == oracle autograd; opcheck of both ops."""
    gen = torch.Generator().manual_seed(11)
    n_src, n_dst, H, C, m = 40, 30, 4, 16, 200
    ei = rand_graph(gen, n_src, n_dst, m)
    q, k, v, e = (torch.randn(n, H, C, generator=gen) for n in (n_dst, n_src, n_src, m))
    w = torch.randn(n_dst, H, C, generator=gen)
    colptr = torch.zeros(n_dst + 1, dtype=torch.long)
    colptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n_dst), 0)
    rowptr = torch.zeros(n_src + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n_src), 0)
    edge_ids = torch.argsort(ei[0], stable=True)
    d = lambda t: t.to(DEV)  # noqa: E731
    leaves = [d(t).requires_grad_(True) for t in (q, k, v, e)]
    out = ops.graph_transformer_attention_conv(*leaves, (d(ei[0]), d(colptr)), (d(rowptr), d(edge_ids), d(ei[1])))
    (out * d(w)).sum().backward()
    _, dq, dk, dv, de = _oracle_grads(q, k, v, e, ei, (n_src, n_dst), w)
    for name, a, b in zip(("dq", "dk", "dv", "de"), leaves, (dq, dk, dv, de)):
        assert_close(a.grad, b, torch.float32, name)
    args = (d(q).requires_grad_(True), d(k).requires_grad_(True), d(v).requires_grad_(True), d(e).requires_grad_(True),
            d(ei[0]), d(colptr), d(rowptr), d(edge_ids), d(ei[1]))
    torch.library.opcheck(torch.ops.anemoi_amd.graph_transformer_attention.default, args,
                          test_utils=("test_schema", "test_autograd_registration", "test_faketensor"))


def test_backward_fullsize_properties(ops):
    """O96 processor graph size (10 242 nodes, 81 840 edges, 16 x 32, bf16): linearity in d_out and the identity
    sum_e dE_e = scatter of (dK_edge + dV_edge) i.e. column sums: sum_s dk_s + dv_s == sum_e dE_e (both reduce the same
    per-edge terms), plus rows of zero-degree nodes are zero."""
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph

    gr = build_synthetic_graph("o8", 5)
    ei = torch.from_numpy(gr.proc_edge_index).long()
    n, H, C = gr.num_hidden, 16, 32
    D, m = H * C, ei.shape[1]
    gen = torch.Generator().manual_seed(3)
    dt = torch.bfloat16
    q, k, v = (torch.randn(n, D, generator=gen).to(dt).to(DEV) for _ in range(3))
    e = (0.5 * torch.randn(m, D, generator=gen)).to(dt).to(DEV)
    g1, g2 = (torch.randn(n, D, generator=gen).to(dt).to(DEV) for _ in range(2))
    csc = ops.build_csc(ei.to(DEV), (n, n))
    rev = ops.build_reverse_csr(csc)
    out, lse = ops.gt_attention(q, k, v, e, csc, H, return_lse=True)
    a = ops.gt_attention_backward(g1, q, k, v, e, out, lse, csc, rev, H)
    b = ops.gt_attention_backward(g2, q, k, v, e, out, lse, csc, rev, H)
    c = ops.gt_attention_backward((g1.float() + 2.0 * g2.float()).to(dt), q, k, v, e, out, lse, csc, rev, H)
    for x, y, z in zip(a, b, c):
        ref = x.float() + 2.0 * y.float()
        assert float((z.float() - ref).abs().max()) <= 6e-2 * float(ref.abs().max())
    dq, dk, dv, de = a
    lhs = (dk.float() + dv.float()).sum(0)
    rhs = de.float().sum(0)
    assert float((lhs - rhs).abs().max()) <= 2e-2 * float(rhs.abs().max()) + 0.5  # bf16 rounding of 8e4 / 1e4 summands


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,C,fe", [(16, 32, 11), (4, 16, 11), (8, 64, 3), (8, 32, 7), (8, 8, 15)])
def test_fused_edge_backward_vs_oracle_autograd(ops, dtype, H, C, fe):
    """anemoi_gt_attention_fused_edge_bwd (lin_edge fused, E / dE never materialised): dq, dk, dv, the gradients of lin_edge's
    weight and bias and of the edge attributes == torch autograd of the fp32 oracle applied to E = edge_attr W^T + b."""
    gen = torch.Generator().manual_seed(7 * H + C + fe)
    n_src, n_dst, m = 70, 50, 400
    D = H * C
    assert ops.fused_edge_backward_supported(D, H, fe)
    ei = rand_graph(gen, n_src, n_dst, m, empty=(3, 7, 20))
    q, k, v, g = (torch.randn(n, H, C, generator=gen).to(dtype) for n in (n_dst, n_src, n_src, n_dst))
    ea = torch.randn(m, fe, generator=gen)
    w, b = (torch.randn(D, fe, generator=gen) / fe**0.5).to(dtype), (0.1 * torch.randn(D, generator=gen)).to(dtype)
    # oracle: differentiable E in fp32
    qs, ks, vs, eas, ws, bs = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v, ea, w, b))
    csc = ops.build_csc(ei.to(DEV), (n_src, n_dst))
    perm = csc.perm.cpu() if csc.perm is not None else torch.arange(m)
    e_ref = (eas @ ws.t() + bs).view(m, H, C)
    O.gt_conv(qs, ks, vs, e_ref, ei, (n_src, n_dst)).backward(g.float())
    flat = lambda t: t.reshape(t.shape[0], D).to(DEV)  # noqa: E731
    feat = ops.pack_edge_features(ea[perm].to(DEV))
    wp = ops.pack_edge_weights(w.to(DEV), b.to(DEV))
    out, lse = ops.gt_attention_fused_edge(flat(q), flat(k), flat(v), feat, wp, csc, H, return_lse=True)
    dq, dk, dv, d_wp, d_feat = ops.gt_attention_fused_edge_backward(flat(g), flat(q), flat(k), flat(v), feat, wp, out, lse, csc,
                                                                    ops.build_reverse_csr(csc), H, need_feat_grad=True)
    for name, a, r in (("dq", dq, qs.grad), ("dk", dk, ks.grad), ("dv", dv, vs.grad)):
        assert_close(a.view(-1, H, C), r, dtype, f"{name} H={H} C={C}")
    # parameter / attribute gradients are fp32 sums over all edges: compare relative to their scale
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    for name, a, r in (("d_weight", d_wp[:, :fe].cpu(), ws.grad), ("d_bias", d_wp[:, fe].cpu(), bs.grad)):
        assert float((a - r).abs().max()) <= tol * float(r.abs().max()) + 1e-5, name
    d_ea = torch.empty(m, fe)
    d_ea[perm] = d_feat[:, :fe].cpu()
    assert float((d_ea - eas.grad).abs().max()) <= tol * float(eas.grad.abs().max()) + 1e-5
    assert float(dq.view(n_dst, -1)[[3, 7, 20]].abs().max()) == 0.0  # rows without edges
    again = ops.gt_attention_fused_edge_backward(flat(g), flat(q), flat(k), flat(v), feat, wp, out, lse, csc,
                                                 ops.build_reverse_csr(csc), H, need_feat_grad=True)
    assert all(torch.equal(x, y) for x, y in zip((dq, dk, dv, d_wp, d_feat), again))  # deterministic
    # forward with an addend (the blocks' self term): the backward takes out + addend and hands back the addend's gradient
    add = torch.randn(n_dst, D, generator=gen).to(dtype).to(DEV)
    y = ops.gt_attention_fused_edge(flat(q), flat(k), flat(v), feat, wp, csc, H, addend=add)
    d_add = torch.empty_like(add)
    withadd = ops.gt_attention_fused_edge_backward(flat(g), flat(q), flat(k), flat(v), feat, wp, y, lse, csc, ops.build_reverse_csr(csc), H,
                                                   need_feat_grad=True, addend=add, d_addend=d_add)
    assert torch.equal(d_add, flat(g))
    for name, a, r2 in zip(("dq", "dk", "dv", "d_wp", "d_feat"), withadd, (dq, dk, dv, d_wp, d_feat)):
        assert float((a.float() - r2.float()).abs().max()) <= (1e-4 if dtype == torch.float32 else 3e-2) * float(r2.float().abs().max()) + 1e-5, name
