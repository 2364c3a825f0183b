"""Attention dropout in training mode (reference layers/conv.py:145: ``alpha = dropout(alpha, p, training)`` after the segment
softmax).  The reference draws its mask from torch's generator, so a fixture with a fixed seed cannot pin OUR mask; what is
checked instead: the mask's invariants, p = 0 equality with the plain op, and forward + gradients against the oracle (autograd of
the fp32 restatement) given the SAME mask."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gt_oracle as O  # noqa: E402  (the checker; tests only)


def _graph(n_src, n_dst, deg, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.randint(0, deg + 1, (n_dst,), generator=g)
    d[0] = 0  # a destination without edges
    dst = torch.repeat_interleave(torch.arange(n_dst), d)
    src = torch.randint(0, n_src, (dst.numel(),), generator=g)
    return torch.stack([src, dst])  # dst-sorted: CSC order == this order


def test_mask_invariants():
    from anemoi_core_amd import ops

    M, H, p = 20000, 8, 0.3
    a = ops.attention_dropout_mask(M, H, p, 1234, "cuda")
    b = ops.attention_dropout_mask(M, H, p, 1234, "cuda")
    c = ops.attention_dropout_mask(M, H, p, 1235, "cuda")
    assert torch.equal(a, b), "same seed, same mask"
    vals = torch.unique(a).cpu().tolist()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p)) < 1e-6
    keep = (a > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / (M * H)) ** 0.5, keep
    assert (a != c).float().mean().item() > 0.3, "another seed, another mask"
    # no structure along either axis: per-head and per-edge-parity keep rates are all near 1 - p
    per_head = (a > 0).float().mean(0).cpu().numpy()
    assert np.all(np.abs(per_head - (1 - p)) < 5 * (p * (1 - p) / M) ** 0.5), per_head
    assert torch.equal(ops.attention_dropout_mask(M, H, 0.0, 7, "cuda"), torch.ones(M, H, device="cuda"))
    with pytest.raises(Exception):
        ops.attention_dropout_mask(M, H, 1.0, 7, "cuda")


@pytest.mark.parametrize("H,C", [(4, 16), (8, 64), (3, 5)])  # wave-per-destination kernels (VEC 1 / 8) and the generic path
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_and_gradients_match_the_oracle_given_the_same_mask(H, C, dtype):
    from anemoi_core_amd import ops
    from anemoi_core_amd.autograd import attention_conv

    n_src, n_dst, p, seed = 150, 120, 0.25, 99
    ei = _graph(n_src, n_dst, 9, 3)
    M, D = ei.shape[1], H * C
    g = torch.Generator().manual_seed(5)
    q, k, v, e = (torch.randn(n, D, generator=g).to(dtype) for n in (n_dst, n_src, n_src, M))
    go = torch.randn(n_dst, D, generator=g).to(dtype)
    csc = ops.build_csc(ei.cuda(), (n_src, n_dst), True)
    mask = ops.attention_dropout_mask(M, H, p, seed, "cuda").cpu()
    assert 0.6 < (mask > 0).float().mean() < 0.9

    leaves = [t.cuda().requires_grad_() for t in (q, k, v, e)]
    out = attention_conv(*leaves, csc, H, None, p, seed)
    out.backward(go.cuda())

    ref_leaves = [t.float().requires_grad_() for t in (q, k, v, e)]
    rq, rk, rv, re_ = (t.view(-1, H, C) for t in ref_leaves)
    ref = O.gt_conv(rq, rk, rv, re_, ei, (n_src, n_dst), alpha_scale=mask).reshape(n_dst, D)
    ref.backward(go.float())

    tol = 2e-5 if dtype == torch.float32 else 3e-2
    def close(a, b, what):
        err = (a.detach().float().cpu() - b.detach()).abs().max().item()
        assert err <= tol * max(1.0, b.detach().abs().max().item()), (what, err)
    close(out, ref, "out")
    for name, a, b in zip("qkve", leaves, ref_leaves):
        close(a.grad, b.grad, "d" + name)
    # and the dropout did something: the same call without it differs
    plain = ops.gt_attention(*[t.detach() for t in leaves], csc, H)
    assert (plain.float() - out.detach().float()).abs().max().item() > 1e-2
    # p = 0 through the dropout entry point is the plain op, bit for bit
    assert torch.equal(attention_conv(*[t.detach() for t in leaves], csc, H, None, 0.0, seed), plain)


def test_conv_module_training_mode():
    """GraphTransformerConv(dropout=p): eval = identity of the plain op; train = masked, reproducible under torch.manual_seed,
    differentiable; p = 0 in training equals eval."""
    from anemoi_core_amd.layers.conv import GraphTransformerConv

    H, C, n_src, n_dst = 4, 16, 90, 70
    ei = _graph(n_src, n_dst, 7, 11)
    # scrambled edge order: the module sorts by destination (csc.perm) and the edge gradient comes back in the caller's order
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(2))
    ei = ei[:, perm].cuda()
    g = torch.Generator().manual_seed(8)
    q = torch.randn(n_dst, H, C, generator=g).cuda().requires_grad_()
    k, v = (torch.randn(n_src, H, C, generator=g).cuda().requires_grad_() for _ in range(2))
    e = torch.randn(ei.shape[1], H, C, generator=g).cuda().requires_grad_()
    conv = GraphTransformerConv(out_channels=C, dropout=0.4)
    with torch.no_grad():
        y_eval = conv.eval()(q, k, v, e, ei, (n_src, n_dst))
    conv.train()
    torch.manual_seed(42)
    y1 = conv(q, k, v, e, ei, (n_src, n_dst))
    torch.manual_seed(42)
    y2 = conv(q, k, v, e, ei, (n_src, n_dst))
    y3 = conv(q, k, v, e, ei, (n_src, n_dst))
    assert torch.equal(y1, y2) and not torch.equal(y1, y3) and not torch.allclose(y1, y_eval)
    y1.sum().backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in (q, k, v, e))
    assert e.grad.shape == e.shape
    # expectation over masks ~ the eval output (64 draws; generous bound)
    with torch.no_grad():
        mean = sum(conv(q, k, v, e, ei, (n_src, n_dst)) for _ in range(64)) / 64
    assert (mean - y_eval).abs().mean().item() < 0.35 * y_eval.abs().mean().item()
    conv0 = GraphTransformerConv(out_channels=C, dropout=0.0).train()
    assert torch.equal(conv0(q, k, v, e, ei, (n_src, n_dst)).detach(), y_eval)
    # without an edge tensor the module still trains (zeros are materialised for the backward kernels)
    conv(q, k, v, None, ei, (n_src, n_dst)).sum().backward()
