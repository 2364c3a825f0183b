"""Tensor-level entry points of the MI355X kernels (thin wrappers over the C ABI, include/anemoi_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every function checks its
arguments, allocates the output with torch, and enqueues one kernel through ctypes.  All tensors must live
on a ROCm device — there is no CPU / eager fallback (a CPU tensor raises).

The op-level mirror of the reference boundary is at the bottom:
``anemoi_amd::graph_transformer_attention`` and ``graph_transformer_attention_conv`` follow
``anemoi::graph_transformer_attention`` / ``graph_transformer_attention_conv``
(reference models/src/anemoi/models/triton/gt.py:390-428, 564-576).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import os

import torch
from torch import Tensor

from . import _ext, _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


def _dt(t: Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {t.dtype}; supported: float32, bfloat16, float16") from None


try:  # raw handle of torch's current stream without constructing a Stream object (host launch overhead matters at N > 1)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    _raw_stream = None


def _stream() -> int:
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts: Optional[Tensor]) -> None:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "anemoi_core_amd kernels run on an MI355X (ROCm) device only; got a tensor on "
                f"'{t.device}'. There is no CPU fallback in the product path."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    # kernels are enqueued on the CURRENT device's current stream (_stream): tensors of another GPU would be launched on the
    # wrong device / stream.  Fail loudly instead (multi-GPU processes: torch.cuda.set_device / `with torch.cuda.device(d)`).
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}; make {dev} current "
                           "(torch.cuda.set_device or `with torch.cuda.device(...)`) before calling anemoi_core_amd ops")


def _rows(t: Optional[Tensor], name: str, dtype=None) -> tuple[int, int]:
    """(data_ptr, leading dimension) of a 2-D row-major view whose last dim is contiguous."""
    if t is None:
        return 0, 0
    shape, stride = t.shape, t.stride()
    if len(shape) != 2:
        raise ValueError(f"{name}: expected a 2-D tensor, got shape {tuple(shape)}")
    if shape[1] > 1 and stride[1] != 1:
        raise ValueError(f"{name}: last dimension must be contiguous (strides {stride})")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype} does not match {dtype}")
    ld = stride[0] if shape[0] > 1 else max(shape[1], stride[0])
    return t.data_ptr(), ld


def _vec(t: Optional[Tensor], name: str, n: int, dtype) -> int:
    if t is None:
        return 0
    if t.dim() != 1 or t.shape[0] != n or not t.is_contiguous() or t.dtype != dtype:
        raise ValueError(f"{name}: expected contiguous [{n}] {dtype}, got {tuple(t.shape)} {t.dtype}")
    return t.data_ptr()


# ------------------------------------------------------------------------------------------ static graph structure
@dataclass(frozen=True)
class CSC:
    """dst-sorted (CSC) adjacency of a static graph, int32, built once (the reference rebuilds it on every
    call: layers/block.py:779-785, triton/utils.py:25-70)."""

    row: Tensor  # [M] int32 source node of every edge, CSC order
    dst: Tensor  # [M] int32 destination node of every edge, CSC order
    colptr: Tensor  # [n_dst + 1] int32
    n_src: int
    n_dst: int
    perm: Optional[Tensor] = None  # original -> CSC edge order when the input was not dst-sorted
    order: Optional[Tensor] = None  # [n_dst] int32: the order the fused attention WORKS on the destinations (processing_order)

    @property
    def num_edges(self) -> int:
        return self.row.shape[0]


def build_csc(edge_index: Tensor, size: tuple[int, int], edges_are_dst_sorted: bool = True, check: bool = False) -> CSC:
    """edge_index [2, M] (src, dst) any integer dtype -> CSC.  Pure index bookkeeping (torch ops; works on any
    device so that the host logic is testable without a GPU)."""
    n_src, n_dst = int(size[0]), int(size[1])
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be [2, M], got {tuple(edge_index.shape)}")
    if max(n_src, n_dst, edge_index.shape[1]) >= 2**31:
        raise ValueError("graph too large for int32 indices")
    src, dst = edge_index[0].long(), edge_index[1].long()
    perm = None
    if not edges_are_dst_sorted:
        perm = torch.sort(dst, stable=True)[1]  # sort_edge_index_by_dst (distributed/khop_edges.py:37-40)
        src, dst = src[perm], dst[perm]
    elif check and dst.numel() > 1 and not bool((dst[1:] >= dst[:-1]).all()):
        raise ValueError("edge_index is not sorted by destination node")
    colptr = torch.zeros(n_dst + 1, dtype=torch.long, device=edge_index.device)
    if dst.numel():
        colptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n_dst), 0)
    return CSC(row=src.to(torch.int32).contiguous(), dst=dst.to(torch.int32).contiguous(),
               colptr=colptr.to(torch.int32).contiguous(), n_src=n_src, n_dst=n_dst, perm=perm)


def processing_order(csc: CSC, slices: int = 8) -> Optional[Tensor]:
    """A permutation of the destinations of a SQUARE graph that keeps mesh neighbours together inside each of the `slices`
    contiguous index ranges the fused attention hands to the XCDs (csrc/gt_attention.hip): breadth-first order over the
    SHORT edges of the range's own sub-graph.  Why: the hidden mesh is sorted by latitude, so ~768 consecutive destinations are
    a whole latitude ring whose K|V neighbourhood (three rings, 2 KiB per row) overflows an XCD's 4 MiB L2 from res 6 on
    (measured 1.65x the compulsory traffic); consecutive BFS levels of a strip form compact patches instead.  "Short": an edge
    with an endpoint of minimal in-degree - in a multi-scale icosphere every finest-level edge touches a vertex that exists at
    the finest level only (in-degree 6), while the long coarse-level edges join high-degree vertices; in a graph of uniform
    degree every edge qualifies.  Host-side index work, once per static graph; None when there is nothing to gain (bipartite
    graphs, graphs whose K|V rows fit the L2s anyway)."""
    import numpy as np

    n = csc.n_dst
    if csc.n_src != n or n < int(os.environ.get("ANEMOI_ATTN_ORDER_MIN_NODES", "16384")) or csc.num_edges == 0:
        return None
    src = csc.row.cpu().numpy().astype(np.int64)
    dst = csc.dst.cpu().numpy().astype(np.int64)
    deg = np.bincount(dst, minlength=n)
    dmin = deg[deg > 0].min()
    keep = np.minimum(deg[src], deg[dst]) == dmin
    src, dst = src[keep], dst[keep]
    per = (n + slices - 1) // slices
    same = (src // per) == (dst // per)  # edges inside one XCD's range
    src, dst = src[same], dst[same]
    o = np.argsort(dst, kind="stable")
    src, dst = src[o], dst[o]
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptr, dst + 1, 1)
    ptr = np.cumsum(ptr)
    seen = np.zeros(n, dtype=bool)
    out = []
    for x in range(slices):
        lo, hi = x * per, min(n, (x + 1) * per)
        nxt = lo
        while nxt < hi:
            if seen[nxt]:
                nxt += 1
                continue
            frontier = np.array([nxt], dtype=np.int64)
            seen[nxt] = True
            while frontier.size:
                out.append(frontier)
                starts, ends = ptr[frontier], ptr[frontier + 1]
                cnt = ends - starts
                if cnt.sum() == 0:
                    break
                idx = np.repeat(starts - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(cnt.sum())
                cand = np.unique(src[idx])
                cand = cand[~seen[cand]]
                seen[cand] = True
                frontier = cand
    order = np.concatenate(out) if out else np.arange(n)
    assert order.size == n and np.array_equal(np.sort(order), np.arange(n)), "processing_order: not a permutation"
    return torch.from_numpy(order.astype(np.int32)).to(csc.row.device)


# ------------------------------------------------------------------------------------------ kernels
def gt_attention(q: Tensor, k: Tensor, v: Tensor, e: Optional[Tensor], csc: CSC, num_heads: int,
                 addend: Optional[Tensor] = None, return_lse: bool = False, dropout_p: float = 0.0, dropout_seed: int = 0):
    """out[d] = softmax-attention over in-edges (+ addend).  q/out/addend [n_dst, D]; k, v [n_src, D]; e [M, D].
    ``dropout_p`` > 0: dropout on the softmax weights (conv.py:145), mask = f(dropout_seed, edge, head) (``attention_dropout_mask``)."""
    _dev(q, k, v, e, addend, csc.row)
    if not 0.0 <= dropout_p < 1.0:
        raise ValueError(f"dropout probability must be in [0, 1), got {dropout_p}")
    D = q.shape[1]
    if D % num_heads:
        raise ValueError(f"channels {D} not divisible by heads {num_heads}")
    if q.shape[0] != csc.n_dst or k.shape[0] != csc.n_src or v.shape[0] != csc.n_src:
        raise ValueError(f"node counts {q.shape[0]}/{k.shape[0]}/{v.shape[0]} do not match the graph ({csc.n_dst}, {csc.n_src})")
    if e is not None and e.shape[0] != csc.num_edges:
        raise ValueError(f"edge tensor has {e.shape[0]} rows, graph has {csc.num_edges} edges")
    out = torch.empty((csc.n_dst, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((csc.n_dst, num_heads), dtype=torch.float32, device=q.device) if return_lse else None
    (qp, ldq), (kp, ldk), (vp, ldv) = _rows(q, "q"), _rows(k, "k", q.dtype), _rows(v, "v", q.dtype)
    ep, lde = _rows(e, "e", q.dtype)
    ap, lda = _rows(addend, "addend", q.dtype)
    rc = _lib.load().anemoi_gt_attention_dropout_fwd(qp, ldq, kp, ldk, vp, ldv, ep, lde, csc.row.data_ptr(), csc.colptr.data_ptr(),
                                                     ap, lda, out.data_ptr(), D, lse.data_ptr() if return_lse else 0,
                                                     csc.n_dst, csc.n_src, num_heads, D // num_heads, float(dropout_p),
                                                     int(dropout_seed) & (2**64 - 1), _dt(q), _stream())
    _lib.check(rc, "gt_attention_fwd")
    return (out, lse) if return_lse else out


def attention_dropout_mask(num_edges: int, num_heads: int, dropout_p: float, dropout_seed: int, device) -> Tensor:
    """fp32 [M, H]: 0 where ``gt_attention(dropout_p, dropout_seed)`` drops the weight of (CSC edge, head), 1 / (1 - p) where it
    keeps it - the scale the kernels derive on the fly."""
    out = torch.empty((num_edges, num_heads), dtype=torch.float32, device=device)
    _dev(out)
    _lib.check(_lib.load().anemoi_attention_dropout_mask(out.data_ptr(), num_edges, num_heads, float(dropout_p), int(dropout_seed) & (2**64 - 1),
                                                         _stream()), "attention_dropout_mask")
    return out


def build_reverse_csr(csc: CSC) -> tuple[Tensor, Tensor, Tensor]:
    """(rowptr [n_src+1], edge_ids [M], edge_dst [M]) int32: the CSC edges grouped by SOURCE, as the backward pass walks
    them (reference triton/utils.py:25-70 returns the same triple next to the CSC).  Index bookkeeping, any device."""
    order = torch.sort(csc.row.long(), stable=True)[1]
    rowptr = torch.zeros(csc.n_src + 1, dtype=torch.long, device=csc.row.device)
    if csc.num_edges:
        rowptr[1:] = torch.cumsum(torch.bincount(csc.row.long(), minlength=csc.n_src), 0)
    return rowptr.to(torch.int32).contiguous(), order.to(torch.int32).contiguous(), csc.dst


def gt_attention_backward(d_out: Tensor, q: Tensor, k: Tensor, v: Tensor, e: Tensor, out: Tensor, lse: Tensor, csc: CSC,
                          reverse: tuple[Tensor, Tensor, Tensor], num_heads: int, grads_out=None, dropout_p: float = 0.0,
                          dropout_seed: int = 0):
    """Gradients (dq, dk, dv, de) of ``gt_attention`` with a materialised edge tensor.  All node/edge tensors [rows, D];
    ``out``/``lse`` are the forward's results; ``reverse`` = build_reverse_csr(csc).  ``grads_out`` = (dq, dk, dv): write
    the node gradients into these (row-strided) views, e.g. column slabs of one fused-projection gradient buffer."""
    rowptr, edge_ids, edge_dst = reverse
    _dev(d_out, q, k, v, e, out, lse, csc.row, rowptr, edge_ids, edge_dst)
    D = q.shape[1]
    if D % num_heads:
        raise ValueError(f"channels {D} not divisible by heads {num_heads}")
    M = csc.num_edges
    if q.shape[0] != csc.n_dst or k.shape[0] != csc.n_src or v.shape[0] != csc.n_src or e.shape[0] != M:
        raise ValueError("tensor row counts do not match the graph")
    if d_out.shape != q.shape or out.shape != q.shape or tuple(lse.shape) != (csc.n_dst, num_heads) or lse.dtype != torch.float32:
        raise ValueError("d_out/out must have q's shape and lse must be fp32 [n_dst, H]")
    if rowptr.shape[0] != csc.n_src + 1 or edge_ids.shape[0] != M or edge_dst.shape[0] != M:
        raise ValueError("reverse CSR does not match the graph")
    if grads_out is None:
        dq, dk, dv = torch.empty_like(q, memory_format=torch.contiguous_format), torch.empty((csc.n_src, D), dtype=q.dtype, device=q.device), \
            torch.empty((csc.n_src, D), dtype=q.dtype, device=q.device)
    else:
        dq, dk, dv = grads_out
        if dq.shape != q.shape or dk.shape != k.shape or dv.shape != v.shape:
            raise ValueError("grads_out shapes must equal q, k, v")
    de = torch.empty((M, D), dtype=q.dtype, device=q.device)
    (dqp, lddq), (dkp, lddk), (dvp, lddv) = _rows(dq, "dq", q.dtype), _rows(dk, "dk", q.dtype), _rows(dv, "dv", q.dtype)
    ws = torch.empty((2, M, num_heads), dtype=torch.float32, device=q.device)
    (qp, ldq), (kp, ldk), (vp, ldv), (ep, lde) = _rows(q, "q"), _rows(k, "k", q.dtype), _rows(v, "v", q.dtype), _rows(e, "e", q.dtype)
    (op, ldo), (gp, ldg) = _rows(out, "out", q.dtype), _rows(d_out, "d_out", q.dtype)
    i32 = lambda t: t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()  # noqa: E731
    rowptr, edge_ids, edge_dst = i32(rowptr), i32(edge_ids), i32(edge_dst)
    rc = _lib.load().anemoi_gt_attention_dropout_bwd(
        qp, ldq, kp, ldk, vp, ldv, ep, lde, op, ldo, lse.contiguous().data_ptr(), gp, ldg, csc.row.data_ptr(), csc.colptr.data_ptr(),
        rowptr.data_ptr(), edge_ids.data_ptr(), edge_dst.data_ptr(), dqp, lddq, dkp, lddk, dvp, lddv,
        de.data_ptr(), D, ws[0].data_ptr(), ws[1].data_ptr(), csc.n_dst, csc.n_src, M, num_heads, D // num_heads, float(dropout_p),
        int(dropout_seed) & (2**64 - 1), _dt(q), _stream())
    _lib.check(rc, "gt_attention_bwd")
    return dq, dk, dv, de


def fused_edge_backward_supported(D: int, num_heads: int, fe: int) -> bool:
    """Shapes the fused-edge backward kernels are instantiated for (csrc/gt_attention_bwd.hip: launch_fused)."""
    if D % 64 or D % num_heads:
        return False
    vec, C = D // 64, D // num_heads
    lph = C // vec if vec and C % vec == 0 else 0
    return (vec, lph) in ((8, 4), (8, 8), (4, 8), (1, 16), (1, 8)) and edge_feature_pad(fe) <= 16


def gt_attention_fused_edge_backward(d_out: Tensor, q: Tensor, k: Tensor, v: Tensor, edge_feat: Tensor, w_packed: Tensor, out: Tensor,
                                     lse: Tensor, csc: CSC, reverse: tuple[Tensor, Tensor, Tensor], num_heads: int, grads_out=None,
                                     need_feat_grad: bool = False, addend: Optional[Tensor] = None, d_addend: Optional[Tensor] = None):
    """Gradients of ``gt_attention_fused_edge``: (dq, dk, dv, d_w_packed fp32 [D, fe_pad], d_edge_feat fp32 [M, fe_pad] or
    None).  E and dE are never materialised.  ``out`` / ``lse``: the forward's results; ``addend``: the forward's addend if it
    had one (``out`` includes it); ``d_addend``: a [n_dst, D] view that receives the addend's gradient (= d_out)."""
    rowptr, edge_ids, edge_dst = reverse
    _dev(d_out, q, k, v, edge_feat, w_packed, out, lse, csc.row, rowptr, edge_ids, edge_dst)
    D = q.shape[1]
    M, fe_pad = csc.num_edges, w_packed.shape[1]
    if edge_feat.dtype != torch.float32 or tuple(edge_feat.shape) != (M, fe_pad) or not edge_feat.is_contiguous():
        raise ValueError(f"edge_feat must be contiguous fp32 [{M}, {fe_pad}]")
    if w_packed.dtype != torch.float32 or tuple(w_packed.shape) != (D, fe_pad) or not w_packed.is_contiguous():
        raise ValueError(f"w_packed must be contiguous fp32 [{D}, {fe_pad}]")
    if d_out.shape != q.shape or out.shape != q.shape or tuple(lse.shape) != (csc.n_dst, num_heads) or lse.dtype != torch.float32:
        raise ValueError("d_out/out must have q's shape and lse must be fp32 [n_dst, H]")
    dev = q.device
    if grads_out is None:
        dq = torch.empty_like(q, memory_format=torch.contiguous_format)
        dk, dv = torch.empty((csc.n_src, D), dtype=q.dtype, device=dev), torch.empty((csc.n_src, D), dtype=q.dtype, device=dev)
    else:
        dq, dk, dv = grads_out
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    lib = _lib.load()
    d_wp = f32(D, fe_pad)
    d_feat = f32(M, fe_pad) if need_feat_grad else None
    ws, sf = f32(2, max(M, 1), num_heads), f32(csc.n_dst, num_heads, 2, fe_pad)
    qg = f32(csc.n_dst, num_heads, 2, fe_pad) if need_feat_grad else None
    part = f32(int(lib.anemoi_gt_attention_fused_edge_bwd_partial_floats(num_heads, D // num_heads, fe_pad)))
    (qp, ldq), (kp, ldk), (vp, ldv) = _rows(q, "q"), _rows(k, "k", q.dtype), _rows(v, "v", q.dtype)
    (op, ldo), (gp, ldg) = _rows(out, "out", q.dtype), _rows(d_out, "d_out", q.dtype)
    (dqp, lddq), (dkp, lddk), (dvp, lddv) = _rows(dq, "dq", q.dtype), _rows(dk, "dk", q.dtype), _rows(dv, "dv", q.dtype)
    i32 = lambda t: t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()  # noqa: E731
    rowptr, edge_ids, edge_dst = i32(rowptr), i32(edge_ids), i32(edge_dst)
    rc = lib.anemoi_gt_attention_fused_edge_bwd(
        qp, ldq, kp, ldk, vp, ldv, edge_feat.data_ptr(), fe_pad, w_packed.data_ptr(), op, ldo, lse.contiguous().data_ptr(), gp, ldg,
        csc.row.data_ptr(), csc.colptr.data_ptr(), rowptr.data_ptr(), edge_ids.data_ptr(), edge_dst.data_ptr(), dqp, lddq, dkp, lddk,
        dvp, lddv, d_wp.data_ptr(), 0 if d_feat is None else d_feat.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), sf.data_ptr(),
        0 if qg is None else qg.data_ptr(), part.data_ptr(), *_rows(addend, "addend", q.dtype), *_rows(d_addend, "d_addend", q.dtype),
        csc.n_dst, csc.n_src, M, num_heads, D // num_heads, _dt(q), _stream())
    _lib.check(rc, "gt_attention_fused_edge_bwd")
    return dq, dk, dv, d_wp, d_feat


def edge_feature_pad(fe: int) -> int:
    return 4 * ((fe + 1 + 3) // 4)


def pack_edge_features(edge_attr: Tensor) -> Tensor:
    """[M, Fe] (any float dtype) -> fp32 [M, fe_pad] = [edge_attr | 1 | 0...] for the fused-edge attention."""
    _dev(edge_attr)
    M, fe = edge_attr.shape
    fe_pad = edge_feature_pad(fe)
    out = torch.empty((M, fe_pad), dtype=torch.float32, device=edge_attr.device)
    p, ld = _rows(edge_attr, "edge_attr")
    _lib.check(_lib.load().anemoi_pack_edge_features(p, ld, out.data_ptr(), M, fe, fe_pad, _dt(edge_attr), _stream()), "pack_edge_features")
    return out


def pack_edge_weights(w_edge: Tensor, b_edge: Optional[Tensor]) -> Tensor:
    """lin_edge parameters -> fp32 [D, fe_pad] = [w_edge | b_edge | 0] for the fused-edge attention."""
    _dev(w_edge, b_edge)
    D, fe = w_edge.shape
    fe_pad = edge_feature_pad(fe)
    if not w_edge.is_contiguous():
        raise ValueError("w_edge must be contiguous [D, Fe]")
    out = torch.empty((D, fe_pad), dtype=torch.float32, device=w_edge.device)
    rc = _lib.load().anemoi_pack_edge_weights(w_edge.data_ptr(), _vec(b_edge, "b_edge", D, w_edge.dtype), out.data_ptr(), D, fe, fe_pad,
                                              _dt(w_edge), _stream())
    _lib.check(rc, "pack_edge_weights")
    return out


def gt_attention_fused_edge(q: Tensor, k: Tensor, v: Tensor, edge_feat: Tensor, w_packed: Tensor, csc: CSC, num_heads: int,
                            addend: Optional[Tensor] = None, return_lse: bool = False):
    """Attention with lin_edge fused: edge_feat = pack_edge_features(edge_attr) fp32 [M, fe_pad];
    w_packed = pack_edge_weights(lin_edge.weight, lin_edge.bias) fp32 [D, fe_pad]."""
    ext = _ext.ops()
    if ext is not None:
        out, lse = ext.gt_attention_fused_edge(q, k, v, edge_feat, w_packed, csc.row, csc.colptr, csc.order, csc.n_src, num_heads, addend, return_lse)
        return (out, lse) if return_lse else out
    _dev(q, k, v, edge_feat, w_packed, addend, csc.row)
    D = q.shape[1]
    fe_pad = w_packed.shape[1]
    if D % num_heads:
        raise ValueError(f"channels {D} not divisible by heads {num_heads}")
    if edge_feat.dtype != torch.float32 or tuple(edge_feat.shape) != (csc.num_edges, fe_pad) or not edge_feat.is_contiguous():
        raise ValueError(f"edge_feat must be contiguous fp32 [{csc.num_edges}, {fe_pad}], got {tuple(edge_feat.shape)} {edge_feat.dtype}")
    if w_packed.dtype != torch.float32 or tuple(w_packed.shape) != (D, fe_pad) or not w_packed.is_contiguous():
        raise ValueError(f"w_packed must be contiguous fp32 [{D}, {fe_pad}]")
    if q.shape[0] != csc.n_dst or k.shape[0] != csc.n_src or v.shape[0] != csc.n_src:
        raise ValueError("node counts do not match the graph")
    out = torch.empty((csc.n_dst, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((csc.n_dst, num_heads), dtype=torch.float32, device=q.device) if return_lse else None
    (qp, ldq), (kp, ldk), (vp, ldv) = _rows(q, "q"), _rows(k, "k", q.dtype), _rows(v, "v", q.dtype)
    ap, lda = _rows(addend, "addend", q.dtype)
    rc = _lib.load().anemoi_gt_attention_fused_edge_fwd(
        qp, ldq, kp, ldk, vp, ldv, edge_feat.data_ptr(), fe_pad, w_packed.data_ptr(),
        csc.row.data_ptr(), csc.colptr.data_ptr(), csc.order.data_ptr() if csc.order is not None else 0, ap, lda, out.data_ptr(), D,
        lse.data_ptr() if return_lse else 0,
        csc.n_dst, csc.n_src, num_heads, D // num_heads, _dt(q), _stream())
    _lib.check(rc, "gt_attention_fused_edge_fwd")
    return (out, lse) if return_lse else out


def cond_layer_norm(x: Tensor, scale: Tensor, shift: Tensor, eps: float = 1e-5) -> Tensor:
    """y = LayerNorm(x) * (scale + 1) + shift over the last dim, per-row scale / shift [N, D] (column slices allowed)."""
    if _needs_grad(x, scale, shift):
        from .autograd import CondLayerNormFunction

        return CondLayerNormFunction.apply(x, scale, shift, float(eps))
    return _cond_layer_norm_fwd(x, scale, shift, eps)


def _cond_layer_norm_fwd(x: Tensor, scale: Tensor, shift: Tensor, eps: float = 1e-5) -> Tensor:
    _dev(x, scale, shift)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if tuple(scale.shape) != tuple(x2.shape) or tuple(shift.shape) != tuple(x2.shape):
        raise ValueError("scale / shift must be [rows(x), D]")
    y = torch.empty((x2.shape[0], D), dtype=x.dtype, device=x.device)
    (p, ld), (sp, lds), (bp, ldb) = _rows(x2, "x"), _rows(scale, "scale", x.dtype), _rows(shift, "shift", x.dtype)
    _lib.check(_lib.load().anemoi_cond_layernorm_fwd(p, ld, sp, lds, bp, ldb, y.data_ptr(), D, x2.shape[0], D, float(eps), _dt(x), _stream()),
               "cond_layernorm_fwd")
    return y.view(x.shape)


def cond_layer_norm_backward(d_y: Tensor, x: Tensor, scale: Tensor, eps: float = 1e-5):
    """(dx, d_scale) of cond_layer_norm, both [N, D]; d_shift is d_y itself."""
    _dev(d_y, x, scale)
    D = x.shape[-1]
    x2, g2 = x.reshape(-1, D), d_y.reshape(-1, D)
    dx = torch.empty((x2.shape[0], D), dtype=x.dtype, device=x.device)
    ds = torch.empty((x2.shape[0], D), dtype=x.dtype, device=x.device)
    (xp, ldx), (sp, lds), (gp, ldg) = _rows(x2, "x"), _rows(scale, "scale", x.dtype), _rows(g2, "d_y", x.dtype)
    _lib.check(_lib.load().anemoi_cond_layernorm_bwd(xp, ldx, sp, lds, gp, ldg, dx.data_ptr(), D, ds.data_ptr(), D, x2.shape[0], D, float(eps),
                                                     _dt(x), _stream()), "cond_layernorm_bwd")
    return dx.view(x.shape), ds


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def layer_norm(x: Tensor, weight: Tensor, bias: Optional[Tensor], eps: float = 1e-5, residual: Optional[Tensor] = None,
               out: Optional[Tensor] = None) -> Tensor:
    """LayerNorm over the last dim of a [..., D] tensor (fp32 statistics), optionally + residual (same shape).
    Differentiable: with autograd recording it runs as ``autograd.LayerNormFunction`` (HIP backward kernel).
    ``out`` (inference only): a contiguous [rows, D] tensor that receives the result."""
    if out is not None:
        if _needs_grad(x, weight, bias, residual):
            raise ValueError("layer_norm: out= is an inference-only argument")
        return _layer_norm_fwd(x, weight, bias, eps, residual, out)
    if _needs_grad(x, weight, bias, residual):
        from .autograd import LayerNormFunction

        y = LayerNormFunction.apply(x, weight, bias, float(eps))
        return y if residual is None else y + residual
    return _layer_norm_fwd(x, weight, bias, eps, residual)


def _layer_norm_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], eps: float = 1e-5, residual: Optional[Tensor] = None,
                    out: Optional[Tensor] = None) -> Tensor:
    """``out``: a contiguous [rows, D] tensor (e.g. the leading rows of a larger buffer) that receives the result."""
    ext = _ext.ops()
    if ext is not None:  # the same entry point through the TORCH_LIBRARY layer: checks and marshalling in C++
        if out is None:
            return ext.layer_norm(x, weight, bias, float(eps), residual)
        ext.layer_norm_out(x, weight, bias, float(eps), residual, out)
        return out.view(x.shape)
    _dev(x, weight, bias, residual, out)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if x2.shape[1] > 1 and x2.stride(1) != 1:
        x2 = x2.contiguous()
    if out is not None and (tuple(out.shape) != (x2.shape[0], D) or out.dtype != x.dtype or not out.is_contiguous()):
        raise ValueError("layer_norm: out must be a contiguous [rows, D] tensor of x's dtype")
    y = torch.empty((x2.shape[0], D), dtype=x.dtype, device=x.device) if out is None else out
    p, ld = _rows(x2, "x")
    if residual is not None and residual.shape != x.shape:
        raise ValueError("residual shape does not match x")
    rp, ldr = _rows(None if residual is None else residual.reshape(-1, D), "residual", x.dtype)
    rc = _lib.load().anemoi_layernorm_fwd(p, ld, _vec(weight, "weight", D, x.dtype), _vec(bias, "bias", D, x.dtype), rp, ldr,
                                          y.data_ptr(), D, x2.shape[0], D, float(eps), _dt(x), _stream())
    _lib.check(rc, "layernorm_fwd")
    return y.view(x.shape)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, act: Optional[str] = None,
           residual: Optional[Tensor] = None, x2: Optional[Tensor] = None, g1: Optional[Tensor] = None,
           idx1: Optional[Tensor] = None, g2: Optional[Tensor] = None, idx2: Optional[Tensor] = None,
           out: Optional[Tensor] = None, seg1=None, seg2=None) -> Tensor:
    """y = act([x | x2] @ weight^T + bias + g1[idx1] + g2[idx2]) + residual.  x [N, K1], x2 [N, K2], weight [O, K1+K2].
    Differentiable: with autograd recording it runs as ``autograd.LinearFunction``; the gather-add terms then need
    ``seg1`` / ``seg2`` = (ptr int32 [rows(g)+1], ids int32 [N] or None): the rows of the output grouped by idx value
    (for a dst-sorted graph: (colptr, None) for idx = dst and (rowptr, edge_ids) for idx = src)."""
    if _needs_grad(x, weight, bias, residual, x2, g1, g2):
        if out is not None:
            raise ValueError("out= is not supported while autograd is recording")
        from .autograd import LinearFunction

        if x2 is not None:
            x = torch.cat([x, x2.to(x.dtype)], dim=1)
        if g1 is not None or g2 is not None:
            if (g1 is not None and seg1 is None) or (g2 is not None and seg2 is None):
                raise ValueError("the backward of the gather-add terms needs seg1 / seg2 (row groups of idx1 / idx2)")
            return LinearFunction.apply(x, weight, bias, act, residual, g1, idx1, seg1, g2, idx2, seg2)
        return LinearFunction.apply(x, weight, bias, act, residual)
    return _linear_fwd(x, weight, bias, act=act, residual=residual, x2=x2, g1=g1, idx1=idx1, g2=g2, idx2=idx2, out=out)


def _linear_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, act: Optional[str] = None,
                residual: Optional[Tensor] = None, x2: Optional[Tensor] = None, g1: Optional[Tensor] = None,
                idx1: Optional[Tensor] = None, g2: Optional[Tensor] = None, idx2: Optional[Tensor] = None,
                out: Optional[Tensor] = None, want_pre: bool = False):
    """``want_pre`` (training, act = "gelu"): returns (y, pre) with pre = the pre-activation stored by the same kernel, or
    (y, None) when the shape does not run on the DMA-ring kernels (the backward then recomputes it)."""
    if not want_pre:
        ext = _ext.ops()
        if ext is not None:  # the same entry point through the TORCH_LIBRARY layer: checks and marshalling in C++
            if act not in (None, "gelu"):
                raise ValueError(f"unsupported activation {act!r}")
            actc = _lib.ACT_GELU if act == "gelu" else _lib.ACT_NONE
            if out is None:
                return ext.linear(x, weight, bias, actc, residual, x2, g1, idx1, g2, idx2)
            ext.linear_out(x, weight, bias, actc, residual, x2, g1, idx1, g2, idx2, out)
            return out
    _dev(x, weight, bias, residual, x2, g1, idx1, g2, idx2, out)
    N, K1 = x.shape
    K2 = 0 if x2 is None else x2.shape[1]
    O = weight.shape[0]
    if weight.shape[1] != K1 + K2:
        raise ValueError(f"weight is {tuple(weight.shape)}, expected [{O}, {K1 + K2}]")
    if act not in (None, "gelu"):
        raise ValueError(f"unsupported activation {act!r}")
    for name, t in (("x2", x2), ("residual", residual)):
        if t is not None and t.shape[0] != N:
            raise ValueError(f"{name} has {t.shape[0]} rows, expected {N}")
    if residual is not None and residual.shape[1] != O:
        raise ValueError("residual width does not match the output width")
    for name, g, idx in (("g1", g1, idx1), ("g2", g2, idx2)):
        if (g is None) != (idx is None):
            raise ValueError(f"{name} and its index must be given together")
        if g is not None and (g.shape[1] != O or idx.dtype != torch.int32 or idx.shape != (N,) or not idx.is_contiguous()):
            raise ValueError(f"{name}: table must be [*, {O}] and index contiguous int32 [{N}]")
    y = out if out is not None else torch.empty((N, O), dtype=x.dtype, device=x.device)
    dt = x.dtype
    (xp, ldx), (x2p, ldx2), (wp, ldw) = _rows(x, "x"), _rows(x2, "x2", dt), _rows(weight, "weight", dt)
    (g1p, ldg1), (g2p, ldg2), (rp, ldr), (yp, ldy) = _rows(g1, "g1", dt), _rows(g2, "g2", dt), _rows(residual, "residual", dt), _rows(y, "out", dt)
    lib = _lib.load()
    i1p, i2p = idx1.data_ptr() if idx1 is not None else 0, idx2.data_ptr() if idx2 is not None else 0
    actc = _lib.ACT_GELU if act == "gelu" else _lib.ACT_NONE
    if want_pre and act == "gelu" and dt != torch.float32 and N > 0:
        pre = torch.empty((N, O), dtype=dt, device=x.device)
        rc = lib.anemoi_linear_fwd_pre(xp, ldx, K1, x2p, ldx2, K2, wp, ldw, _vec(bias, "bias", O, dt), g1p, ldg1, i1p, g2p, ldg2, i2p, rp, ldr,
                                       yp, ldy, pre.data_ptr(), O, N, O, actc, _dt(x), _stream())
        if rc == 0:
            return y, pre
        if rc != _lib.E_UNSUPPORTED:
            _lib.check(rc, "linear_fwd_pre")
    rc = lib.anemoi_linear_fwd(xp, ldx, K1, x2p, ldx2, K2, wp, ldw, _vec(bias, "bias", O, dt), g1p, ldg1, i1p, g2p, ldg2, i2p, rp, ldr, yp, ldy,
                               N, O, actc, _dt(x), _stream())
    _lib.check(rc, "linear_fwd")
    return (y, None) if want_pre else y


_REDUCE_WS: dict = {}


def _reduce_workspace(D: int, device) -> Tensor:
    """fp32 scratch for the deterministic column sums (per device and width, allocated once)."""
    key = (str(device), D)
    ws = _REDUCE_WS.get(key)
    if ws is None:
        ws = torch.empty(_lib.load().anemoi_reduce_workspace_bytes(D) // 4, dtype=torch.float32, device=device)
        _REDUCE_WS[key] = ws
    return ws


def layer_norm_backward(d_y: Tensor, x: Tensor, weight: Tensor, eps: float = 1e-5, need_param_grads: bool = True):
    """(dx, dgamma fp32 [D], dbeta fp32 [D]) of LayerNorm over the last dim; statistics recomputed from x."""
    _dev(d_y, x, weight)
    D = x.shape[-1]
    x2, g2 = x.reshape(-1, D), d_y.reshape(-1, D)
    dx = torch.empty((x2.shape[0], D), dtype=x.dtype, device=x.device)
    gb = torch.empty((2, D), dtype=torch.float32, device=x.device) if need_param_grads else None  # one buffer: one cast later
    dg, db = (gb[0], gb[1]) if need_param_grads else (None, None)
    (xp, ldx), (gp, ldg) = _rows(x2, "x"), _rows(g2, "d_y", x.dtype)
    rc = _lib.load().anemoi_layernorm_bwd(xp, ldx, _vec(weight, "weight", D, x.dtype), gp, ldg, dx.data_ptr(), D,
                                          dg.data_ptr() if need_param_grads else 0, db.data_ptr() if need_param_grads else 0,
                                          _reduce_workspace(D, x.device).data_ptr() if need_param_grads else 0,
                                          x2.shape[0], D, float(eps), _dt(x), _stream())
    _lib.check(rc, "layernorm_bwd")
    return dx.view(x.shape), dg, db


def colsum(x: Tensor) -> Tensor:
    """fp32 column sums of a [N, D] tensor (bias gradient), deterministic."""
    _dev(x)
    N, D = x.shape
    out = torch.empty(D, dtype=torch.float32, device=x.device)
    p, ld = _rows(x, "x")
    _lib.check(_lib.load().anemoi_colsum(p, ld, out.data_ptr(), _reduce_workspace(D, x.device).data_ptr(), N, D, _dt(x), _stream()), "colsum")
    return out


def gelu(x: Tensor) -> Tensor:
    """Exact (erf) GELU of a [..., D] tensor as ONE stand-alone kernel (the layer_kernels ``Activation`` plug-in; the fused
    blocks apply GELU in a GEMM epilogue instead).  Differentiable (backward: ``gelu_backward``)."""
    if _needs_grad(x):
        from .autograd import GeluFunction

        return GeluFunction.apply(x)
    return _gelu_fwd(x)


def _gelu_fwd(x: Tensor) -> Tensor:
    _dev(x)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if x2.shape[1] > 1 and x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], D), dtype=x.dtype, device=x.device)
    p, ld = _rows(x2, "x")
    _lib.check(_lib.load().anemoi_gelu_fwd(p, ld, y.data_ptr(), D, x2.shape[0], D, _dt(x), _stream()), "gelu_fwd")
    return y.view(x.shape)


def gelu_backward(pre: Tensor, d_y: Tensor) -> Tensor:
    """d_pre = d_y * gelu'(pre) (exact erf form), [N, D]."""
    _dev(pre, d_y)
    N, D = pre.shape
    out = torch.empty((N, D), dtype=pre.dtype, device=pre.device)
    (pp, ldp), (gp, ldg) = _rows(pre, "pre"), _rows(d_y, "d_y", pre.dtype)
    _lib.check(_lib.load().anemoi_gelu_bwd(pp, ldp, gp, ldg, out.data_ptr(), D, N, D, _dt(pre), _stream()), "gelu_bwd")
    return out


def edge_ln_residual_segment_sum(z: Tensor, e_old: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float,
                                 csc: CSC) -> tuple[Tensor, Tensor]:
    """e_new = LayerNorm(z) + e_old;  agg[d] = sum of e_new over the in-edges of d.  Returns (e_new, agg).  Differentiable."""
    if _needs_grad(z, e_old, gamma, beta):
        from .autograd import EdgeLnResidualSegmentSumFunction

        return EdgeLnResidualSegmentSumFunction.apply(z, e_old, gamma, beta, float(eps), csc)
    return _edge_ln_residual_segment_sum_fwd(z, e_old, gamma, beta, eps, csc)


def _edge_ln_residual_segment_sum_fwd(z: Tensor, e_old: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float,
                                      csc: CSC) -> tuple[Tensor, Tensor]:
    _dev(z, e_old, gamma, beta, csc.colptr)
    M, D = z.shape
    if M != csc.num_edges or tuple(e_old.shape) != (M, D):
        raise ValueError("edge tensors do not match the graph")
    e_new = torch.empty((M, D), dtype=z.dtype, device=z.device)
    agg = torch.empty((csc.n_dst, D), dtype=z.dtype, device=z.device)
    (zp, ldz), (ep, lde) = _rows(z, "z"), _rows(e_old, "e_old", z.dtype)
    rc = _lib.load().anemoi_edge_ln_residual_segment_sum_fwd(
        zp, ldz, ep, lde, _vec(gamma, "gamma", D, z.dtype), _vec(beta, "beta", D, z.dtype), float(eps), csc.colptr.data_ptr(),
        e_new.data_ptr(), D, agg.data_ptr(), D, csc.n_dst, D, _dt(z), _stream())
    _lib.check(rc, "edge_ln_residual_segment_sum_fwd")
    return e_new, agg


def segment_sum_rows(x: Tensor, ptr: Tensor, ids: Optional[Tensor] = None) -> Tensor:
    """out[r] = sum_{i in [ptr[r], ptr[r+1])} x[ids[i] if ids is given else i]; ptr int32 [n_out + 1], ids int32."""
    _dev(x, ptr, ids)
    for t in (ptr, ids):
        if t is not None and (t.dtype != torch.int32 or t.dim() != 1 or not t.is_contiguous()):
            raise ValueError("ptr / ids must be contiguous int32 vectors")
    D, n_out = x.shape[1], ptr.shape[0] - 1
    out = torch.empty((n_out, D), dtype=x.dtype, device=x.device)
    p, ld = _rows(x, "x")
    rc = _lib.load().anemoi_segment_sum_rows(p, ld, ptr.data_ptr(), ids.data_ptr() if ids is not None else 0, out.data_ptr(), D, n_out, D,
                                             _dt(x), _stream())
    _lib.check(rc, "segment_sum_rows")
    return out


def gather_add_rows(a: Tensor, b: Tensor, idx: Tensor) -> Tensor:
    """out[i] = a[i] + b[idx[i]] (idx int32 [rows(a)])."""
    _dev(a, b, idx)
    if idx.dtype != torch.int32 or idx.shape != (a.shape[0],) or not idx.is_contiguous() or a.shape[1] != b.shape[1]:
        raise ValueError("idx must be contiguous int32 [rows(a)] and a, b must have the same width")
    D = a.shape[1]
    out = torch.empty((a.shape[0], D), dtype=a.dtype, device=a.device)
    (ap, lda), (bp, ldb) = _rows(a, "a"), _rows(b, "b", a.dtype)
    _lib.check(_lib.load().anemoi_gather_add_rows(ap, lda, bp, ldb, idx.data_ptr(), out.data_ptr(), D, a.shape[0], D, _dt(a), _stream()), "gather_add_rows")
    return out


def linear_splitk(a: Tensor, b: Tensor, splits: int) -> Tensor:
    """fp32 [rows(a), rows(b)] = a @ b^T for 16-bit a [R, K], b [C, K] with the reduction split over ``splits`` groups of
    CUs (fp32 atomics): the weight-gradient GEMM.  K must be a multiple of 64 * splits."""
    _dev(a, b)
    R, K = a.shape
    Cc = b.shape[0]
    if b.shape[1] != K or a.dtype != b.dtype or a.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("linear_splitk: a [R, K] and b [C, K] must be bf16/fp16 of the same dtype")
    y = torch.zeros((R, Cc), dtype=torch.float32, device=a.device)
    (ap, lda), (bp, ldb) = _rows(a, "a"), _rows(b, "b", a.dtype)
    _lib.check(_lib.load().anemoi_linear_splitk_f32(ap, lda, bp, ldb, y.data_ptr(), Cc, R, Cc, K, int(splits), _dt(a), _stream()), "linear_splitk_f32")
    return y


def linear_wgrad_eligible(dz: Tensor, x: Tensor) -> bool:
    return (dz.dtype in (torch.bfloat16, torch.float16) and x.dtype == dz.dtype and dz.shape[1] % 8 == 0 and x.shape[1] % 8 == 0
            and dz.shape[0] > 0)


def linear_wgrad(dz: Tensor, x: Tensor, with_bias_grad: bool = False):
    """dW [O, I] = dz^T x for 16-bit dz [N, O], x [N, I] (the reduction runs over the rows; no transposes in HBM);
    ``with_bias_grad``: returns (dW, db) with db [O] = column sums of dz from the same kernel."""
    _dev(dz, x)
    N, O = dz.shape
    I = x.shape[1]
    if x.shape[0] != N or not linear_wgrad_eligible(dz, x):
        raise ValueError("linear_wgrad: dz [N, O] and x [N, I] must be bf16/fp16 of one dtype with O, I multiples of 8")
    lib = _lib.load()
    ws = torch.empty((int(lib.anemoi_linear_wgrad_workspace_bytes(N, O, I)) // 4,), dtype=torch.float32, device=dz.device)
    dw = torch.empty((O, I), dtype=dz.dtype, device=dz.device)
    (zp, ldz), (xp, ldx) = _rows(dz, "dz"), _rows(x, "x", dz.dtype)
    if ldz % 8 or ldx % 8:
        raise ValueError("linear_wgrad: row strides must be multiples of 8 elements")
    db = torch.empty((O,), dtype=dz.dtype, device=dz.device) if with_bias_grad else None
    _lib.check(lib.anemoi_linear_wgrad(zp, ldz, xp, ldx, dw.data_ptr(), I, 0 if db is None else db.data_ptr(), ws.data_ptr(), N, O, I,
                                       _dt(dz), _stream()), "linear_wgrad")
    return (dw, db) if with_bias_grad else dw


def linear_with_row_stats(x: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Optional[Tensor] = None):
    """(y, stats) with y = x W^T + bias [+ residual] and stats [N, O/64, 2] fp32 = per 64-column strip (sum, sum of squares) of
    the stored rows of y — what ``linear_ln_folded`` needs to apply the LayerNorm of y.  None if the shape is not eligible."""
    ext = _ext.ops()
    if ext is not None:
        y, stats = ext.linear_with_row_stats(x, weight, bias, residual)
        return None if y.dim() != 2 else (y, stats)
    _dev(x, weight, bias, residual)
    N, K = x.shape
    O = weight.shape[0]
    if x.dtype == torch.float32 or O % 64 or K % 64:
        return None
    y = torch.empty((N, O), dtype=x.dtype, device=x.device)
    stats = torch.empty((N, O // 64, 2), dtype=torch.float32, device=x.device)
    dt = x.dtype
    (xp, ldx), (wp, ldw), (rp, ldr) = _rows(x, "x"), _rows(weight, "weight", dt), _rows(residual, "residual", dt)
    rc = _lib.load().anemoi_linear_stats_fwd(xp, ldx, K, wp, ldw, _vec(bias, "bias", O, dt), rp, ldr, y.data_ptr(), O, stats.data_ptr(), N, O,
                                             _dt(x), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "linear_stats_fwd")
    return y, stats


def linear_ln_folded(x: Tensor, w_scaled: Tensor, c: Tensor, d: Tensor, stats: Tensor, eps: float, act: Optional[str] = None):
    """act(LayerNorm(x) W^T + b) from raw x, w_scaled = W * gamma, c = rowsum(w_scaled) (fp32), d = W beta + b (fp32) and the
    producer's row statistics of x.  None if the shape is not eligible (caller: LayerNorm + linear)."""
    ext = _ext.ops()
    if ext is not None:
        y = ext.linear_ln_folded(x, w_scaled, c, d, stats, float(eps), _lib.ACT_GELU if act == "gelu" else _lib.ACT_NONE)
        return None if y.dim() != 2 else y
    _dev(x, w_scaled, c, d, stats)
    N, K = x.shape
    O = w_scaled.shape[0]
    if tuple(stats.shape) != (N, K // 64, 2) or stats.dtype != torch.float32 or not stats.is_contiguous():
        raise ValueError("stats must be contiguous fp32 [N, K/64, 2]")
    y = torch.empty((N, O), dtype=x.dtype, device=x.device)
    dt = x.dtype
    (xp, ldx), (wp, ldw) = _rows(x, "x"), _rows(w_scaled, "w_scaled", dt)
    rc = _lib.load().anemoi_linear_lnfold_fwd(xp, ldx, K, wp, ldw, c.data_ptr(), d.data_ptr(), stats.data_ptr(), K // 64, float(eps),
                                              _lib.ACT_GELU if act == "gelu" else _lib.ACT_NONE, y.data_ptr(), O, N, O, _dt(x), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "linear_lnfold_fwd")
    return y


# ------------------------------------------------------------------------------------------ row-resident layer chain
CHAIN_CHANNELS = 512  # the chain kernels (csrc/gt_chain2.hip, gnn_chain.hip) are built for this width: 4 waves x 128 / 8 waves x 64 columns


def pack_weight_frag(weight: Tensor) -> Tensor:
    """Fragment-major image of a Linear weight [O, K] (O % 64 == 0, K % 32 == 0) for the chain kernels: one contiguous KiB per
    MFMA B fragment, [O/64 slabs][K/32 k-steps][4 column blocks][4 k-slots][16 rows][8] (include/anemoi_hip.h).  A pure
    re-ordering (torch view + permute), made once per parameter version."""
    O, K = weight.shape
    if O % 64 or K % 32:
        raise ValueError(f"pack_weight_frag: weight [{O}, {K}] needs O % 64 == 0 and K % 32 == 0")
    w = weight.detach().reshape(O // 64, 4, 16, K // 32, 4, 8)  # (slab, ni, row, ks, kslot, e)
    return w.permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1)


def gt_layer_chain_supported(x: Tensor, hidden: int, q_out: int = 0) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.shape[1] == CHAIN_CHANNELS and x.dtype in (torch.bfloat16, torch.float16)
            and hidden > 0 and hidden % CHAIN_CHANNELS == 0 and q_out % CHAIN_CHANNELS == 0)


class _Chain2Args(_lib.C.Structure):
    _p, _i64, _i32, _f = _lib.C.c_void_p, _lib.C.c_int64, _lib.C.c_int32, _lib.C.c_float
    _fields_ = [("attn", _p), ("ld_attn", _i64), ("x_res", _p), ("ld_x", _i64), ("wp", _p), ("w1", _p), ("hidden", _i32), ("w2", _p),
                ("wq", _p), ("q_out_features", _i32), ("vec", _p), ("ln1_eps", _f), ("lnq_eps", _f), ("extra", _p), ("ld_extra", _i64),
                ("x_out", _p), ("ld_out", _i64), ("q_out", _p), ("ld_q", _i64), ("n_rows", _i32), ("channels", _i32),
                ("rows_per_tile", _i32), ("timeline", _p)]


CHAIN2_VEC_MAX = 6144  # elements of [b_p | d1 | b_2 | dq] the kernel keeps in LDS (csrc/gt_chain2.hip)


def fold_layer_norm(weight: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Optional[Tensor]) -> tuple[Tensor, Tensor]:
    """``LN(x) W^T + b = ((x - mean) rstd) (W diag(gamma))^T + (W beta + b)``: the affine part of a LayerNorm folded into the Linear
    that follows it.  Returns (W diag(gamma) rounded to the weight's dtype, d = W beta + b in fp32)."""
    w = weight.detach().float()
    b = w.new_zeros(w.shape[0]) if bias is None else bias.detach().float()
    wg = (w * gamma.detach().float()).to(weight.dtype)
    d = b if beta is None else w @ beta.detach().float() + b
    return wg, d


def gt_layer_chain2_supported(x: Tensor, hidden: int, q_out: int = 0) -> bool:
    """``q_out``: a multiple of 512, or a NARROW trailing projection of 128, 256 or 384 columns (the decoder's node_data_extractor)."""
    narrow = 0 < q_out < CHAIN_CHANNELS and q_out % 128 == 0
    return gt_layer_chain_supported(x, hidden, 0 if narrow else q_out) and 2 * CHAIN_CHANNELS + hidden + q_out <= CHAIN2_VEC_MAX


def gt_layer_chain2(attn: Tensor, x_res: Tensor, wp: Tensor, w1g: Tensor, w2: Tensor, vec: Tensor, hidden: int, ln1_eps: float, *,
                    extra: Optional[Tensor] = None, wqg: Optional[Tensor] = None, q_out_features: int = 0, lnq_eps: float = 1e-5,
                    rows_per_tile: int = 0, timeline: Optional[Tensor] = None, want_x_out: bool = True):
    """The row-local part of a GraphTransformer block in ONE launch with role-split waves (anemoi_gt_chain2_fwd, csrc/gt_chain2.hip):

        x1 = attn Wp^T + bp + x_res;  h = GELU(LN(x1; ln1) W1^T + b1);  x_out = h W2^T + b2 + x1 [+ extra]
        q_out = LN(x_out; lnq) Wq^T + bq        (optional: the NEXT block's LayerNorm + fused q|k|v|self projection)

    with the LayerNorms' affine parts folded by the caller (``fold_layer_norm``): ``w1g`` / ``wqg`` are the fragment-major images of
    ``W diag(gamma)``, ``vec = cat[bp, d1, b2, dq]`` in the model dtype with ``d = W beta + b``.  ``q_out_features`` may also be 128, 256 or
    384 (a narrow trailing projection, e.g. the decoder's ``node_data_extractor`` zero-padded to a multiple of 128 rows); with
    ``want_x_out=False`` (needs a trailing projection) x_out is not written and returned as None.  Returns ``x_out`` or ``(x_out, q_out)``.
    Inference only (no autograd)."""
    _dev(attn, x_res, wp, w1g, w2, vec, extra, wqg)
    N, D = attn.shape
    dt = attn.dtype
    if D != CHAIN_CHANNELS or dt not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"gt_layer_chain2: {D} channels / {dt} (built for {CHAIN_CHANNELS} channels, 16-bit dtypes)")
    if 2 * D + hidden + q_out_features > CHAIN2_VEC_MAX:
        raise NotImplementedError(f"gt_layer_chain2: hidden={hidden} + q_out={q_out_features} exceed the kernel's LDS vector region")
    if tuple(x_res.shape) != (N, D) or (extra is not None and tuple(extra.shape) != (N, D)):
        raise ValueError("gt_layer_chain2: attn, x_res and extra must have the same [N, channels] shape")
    if extra is not None and not want_x_out:
        raise ValueError("gt_layer_chain2: a second residual needs x_out")
    for name, w, numel in (("wp", wp, D * D), ("w1g", w1g, hidden * D), ("w2", w2, D * hidden), ("wqg", wqg, q_out_features * D)):
        if w is None and numel == 0:
            continue
        if w is None or w.dim() != 1 or w.numel() != numel or w.dtype != dt or not w.is_contiguous():
            raise ValueError(f"gt_layer_chain2: {name} must be the contiguous fragment-major image ({numel} x {dt}) made by pack_weight_frag")
    if vec.dim() != 1 or vec.numel() != 2 * D + hidden + q_out_features or vec.dtype != dt or not vec.is_contiguous():
        raise ValueError(f"gt_layer_chain2: vec must be contiguous [{2 * D + hidden + q_out_features}] {dt} = cat[bp, d1, b2, dq]")
    if not want_x_out and not q_out_features:
        raise ValueError("gt_layer_chain2: want_x_out=False needs a trailing projection")
    x_out = torch.empty((N, D), dtype=dt, device=attn.device) if want_x_out else None
    q_out = torch.empty((N, q_out_features), dtype=dt, device=attn.device) if q_out_features else None
    (ap, lda), (xp, ldx), (ep, lde) = _rows(attn, "attn", dt), _rows(x_res, "x_res", dt), _rows(extra, "extra", dt)
    a = _Chain2Args(ap, lda, xp, ldx, wp.data_ptr(), w1g.data_ptr(), hidden, w2.data_ptr(), 0 if wqg is None else wqg.data_ptr(), q_out_features,
                    vec.data_ptr(), float(ln1_eps), float(lnq_eps), ep, lde, 0 if x_out is None else x_out.data_ptr(), D, 0 if q_out is None else q_out.data_ptr(),
                    q_out_features, N, D, int(rows_per_tile), 0 if timeline is None else timeline.data_ptr())
    _lib.check(_lib.load().anemoi_gt_chain2_fwd(_lib.C.byref(a), _dt(attn), _stream()), "gt_chain2_fwd")
    return x_out if q_out is None else (x_out, q_out)


class _ClusterChainArgs(_lib.C.Structure):
    _p, _i64, _i32, _f = _lib.C.c_void_p, _lib.C.c_int64, _lib.C.c_int32, _lib.C.c_float
    _fields_ = [("attn", _p), ("ld_attn", _i64), ("x_res", _p), ("ld_x", _i64), ("wp", _p), ("w1", _p), ("hidden", _i32), ("w2", _p),
                ("wq", _p), ("q_out_features", _i32), ("vec", _p), ("ln1_eps", _f), ("lnq_eps", _f), ("extra", _p), ("ld_extra", _i64),
                ("x_out", _p), ("ld_out", _i64), ("q_out", _p), ("ld_q", _i64), ("ln_out", _p), ("ld_ln", _i64), ("q_out2", _p), ("ld_q2", _i64), ("q_split", _i32), ("workspace", _p),
                ("workspace_bytes", _i64), ("n_rows", _i32), ("channels", _i32)]


_CLUSTER_WS: dict = {}


def _cluster_workspace(device) -> Tensor:
    """The cluster chain's exchange workspace (counters + partial-sum slots), zeroed ONCE - its counters are monotonic across launches
    (csrc/gt_cluster_chain.hip).  Launches that share a workspace must be ordered: one per (device, stream) for eager launches and one
    for launches recorded into hipGraphs (a replay is ordered like one stream).  Allocated on first eager use - run a forward eagerly
    before capturing it, as every capture needs anyway."""
    capturing = torch.cuda.is_current_stream_capturing()
    key = (str(device), "graph" if capturing else torch.cuda.current_stream(device).cuda_stream)
    ws = _CLUSTER_WS.get(key)
    if ws is None:
        if capturing:
            raise RuntimeError("gt_cluster_chain: its workspace must exist before hipGraph capture (run the forward once eagerly)")
        n = _lib.load().anemoi_gt_cluster_chain_workspace_bytes()
        ws = _CLUSTER_WS[key] = torch.zeros(n // 4, dtype=torch.int32, device=device)
        _CLUSTER_WS.setdefault((str(device), "graph"), torch.zeros(n // 4, dtype=torch.int32, device=device))
    return ws


def gt_cluster_chain_supported(x: Tensor, hidden: int, q_out: int = 0) -> bool:
    return gt_layer_chain_supported(x, hidden, q_out) and hidden == 4 * CHAIN_CHANNELS and q_out <= 4 * CHAIN_CHANNELS


def gt_cluster_chain(attn: Tensor, x_res: Tensor, wp: Tensor, w1g: Tensor, w2: Tensor, vec: Tensor, hidden: int, ln1_eps: float, *,
                     extra: Optional[Tensor] = None, wqg: Optional[Tensor] = None, q_out_features: int = 0, lnq_eps: float = 1e-5,
                     ln_out: Optional[Tensor] = None, q_out2: Optional[Tensor] = None, q_split: int = 0):
    """``gt_layer_chain2`` for block tails of a few thousand rows (anemoi_gt_cluster_chain_fwd, csrc/gt_cluster_chain.hip): four CUs of one
    XCD share a 48-row panel as a tensor-parallel group over the MLP's hidden width and exchange the second Linear's partial sums once.
    Same operands (``hidden`` must be 2048); ``ln_out`` (optional, [N, 512]): receives LN'(x2) WITHOUT its affine part; ``q_out2`` (optional,
    [N, q_out_features - 512 q_split], a view with any row stride): receives the trailing projection's chunks from ``q_split`` on (the returned
    ``q_out`` then holds the first ``512 q_split`` columns).  Returns ``x_out`` or ``(x_out, q_out)``.  Inference only (no autograd)."""
    _dev(attn, x_res, wp, w1g, w2, vec, extra, wqg, ln_out, q_out2)
    N, D = attn.shape
    dt = attn.dtype
    if not gt_cluster_chain_supported(attn, hidden, q_out_features):
        raise NotImplementedError(f"gt_cluster_chain: {D} channels / {dt} / hidden={hidden} / q_out={q_out_features}")
    if tuple(x_res.shape) != (N, D) or (extra is not None and tuple(extra.shape) != (N, D)) or (ln_out is not None and tuple(ln_out.shape) != (N, D)):
        raise ValueError("gt_cluster_chain: attn, x_res, extra and ln_out must have the same [N, channels] shape")
    if extra is not None and (q_out_features or ln_out is not None):
        raise ValueError("gt_cluster_chain: a trailing projection / LayerNorm output and a second residual exclude each other")
    for name, w, numel in (("wp", wp, D * D), ("w1g", w1g, hidden * D), ("w2", w2, D * hidden), ("wqg", wqg, q_out_features * D)):
        if w is None and numel == 0:
            continue
        if w is None or w.dim() != 1 or w.numel() != numel or w.dtype != dt or not w.is_contiguous():
            raise ValueError(f"gt_cluster_chain: {name} must be the contiguous fragment-major image ({numel} x {dt}) made by pack_weight_frag")
    if vec.dim() != 1 or vec.numel() != 2 * D + hidden + q_out_features or vec.dtype != dt or not vec.is_contiguous():
        raise ValueError(f"gt_cluster_chain: vec must be contiguous [{2 * D + hidden + q_out_features}] {dt} = cat[bp, d1, b2, dq]")
    ws = _cluster_workspace(attn.device)
    x_out = torch.empty((N, D), dtype=dt, device=attn.device)
    q_cols = q_out_features if q_out2 is None else D * q_split
    if q_out2 is not None and (q_split < 0 or D * q_split > q_out_features or tuple(q_out2.shape) != (N, q_out_features - D * q_split)):
        raise ValueError(f"gt_cluster_chain: q_out2 must be [N, {q_out_features} - 512 q_split] with 0 <= 512 q_split <= q_out_features")
    q_out = torch.empty((N, q_cols), dtype=dt, device=attn.device) if q_cols else None
    (ap, lda), (xp, ldx), (ep, lde), (lp, ldl), (q2p, ldq2) = (_rows(attn, "attn", dt), _rows(x_res, "x_res", dt), _rows(extra, "extra", dt),
                                                              _rows(ln_out, "ln_out", dt), _rows(q_out2, "q_out2", dt))
    a = _ClusterChainArgs(ap, lda, xp, ldx, wp.data_ptr(), w1g.data_ptr(), hidden, w2.data_ptr(), 0 if wqg is None else wqg.data_ptr(), q_out_features,
                          vec.data_ptr(), float(ln1_eps), float(lnq_eps), ep, lde, x_out.data_ptr(), D, 0 if q_out is None else q_out.data_ptr(),
                          max(q_cols, 8), lp, ldl, q2p, ldq2, int(q_split), ws.data_ptr(), ws.numel() * 4, N, D)
    _lib.check(_lib.load().anemoi_gt_cluster_chain_fwd(_lib.C.byref(a), _dt(attn), _stream()), "gt_cluster_chain_fwd")
    return x_out if (q_out is None and q_out2 is None) else (x_out, q_out)


class _RowChainArgs(_lib.C.Structure):
    _p, _i64, _i32, _f = _lib.C.c_void_p, _lib.C.c_int64, _lib.C.c_int32, _lib.C.c_float
    _fields_ = [("x", _p), ("ld_x", _i64), ("in_features", _i32), ("we", _p), ("wq", _p), ("q_out_features", _i32), ("vec", _p), ("ln_eps", _f),
                ("x_out", _p), ("ld_out", _i64), ("q_out", _p), ("ld_q", _i64), ("n_rows", _i32), ("channels", _i32), ("rows_per_tile", _i32)]


def pack_embedding_frag(weight: Tensor) -> Tensor:
    """Fragment-major image of an embedding weight [512, in] with its columns zero-padded to a multiple of 128 (``gt_row_chain``)."""
    O, K = weight.shape
    pad = (-K) % 128
    w = weight.detach()
    return pack_weight_frag(torch.nn.functional.pad(w, (0, pad)) if pad else w)


def gt_row_chain_supported(x: Tensor, q_out: int) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and 0 < x.shape[1] <= CHAIN_CHANNELS and x.shape[1] % 8 == 0
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and q_out > 0 and q_out % CHAIN_CHANNELS == 0 and q_out <= 4 * CHAIN_CHANNELS)


def gt_row_chain(x: Tensor, we: Tensor, wqg: Tensor, vec: Tensor, q_out_features: int, ln_eps: float, *, want_x_out: bool = True,
                 rows_per_tile: int = 0):
    """One side of a GraphTransformer mapper in ONE launch (anemoi_gt_rowchain_fwd, csrc/gt_rowchain.hip):

        y = x We^T + be;   q_out = LN(y) Wq^T + bq

    ``we``: ``pack_embedding_frag(We)``; ``wqg``: fragment-major image of ``Wq diag(gamma)``; ``vec = cat[be, Wq beta + bq]`` in the
    model dtype (``fold_layer_norm``).  Returns ``(y or None, q_out)``.  Inference only (no autograd)."""
    _dev(x, we, wqg, vec)
    N, K = x.shape
    dt, D = x.dtype, CHAIN_CHANNELS
    if not gt_row_chain_supported(x, q_out_features):
        raise NotImplementedError(f"gt_row_chain: x {tuple(x.shape)} {dt} / q_out={q_out_features} (16-bit rows of <= {D} columns, a multiple of 8; q_out a multiple of {D})")
    kp = (K + 127) // 128 * 128
    for name, w, numel in (("we", we, D * kp), ("wqg", wqg, q_out_features * D)):
        if w.dim() != 1 or w.numel() != numel or w.dtype != dt or not w.is_contiguous():
            raise ValueError(f"gt_row_chain: {name} must be the contiguous fragment-major image ({numel} x {dt})")
    if vec.dim() != 1 or vec.numel() != D + q_out_features or vec.dtype != dt or not vec.is_contiguous():
        raise ValueError(f"gt_row_chain: vec must be contiguous [{D + q_out_features}] {dt} = cat[be, dq]")
    x_out = torch.empty((N, D), dtype=dt, device=x.device) if want_x_out else None
    q_out = torch.empty((N, q_out_features), dtype=dt, device=x.device)
    xp, ldx = _rows(x, "x", dt)
    a = _RowChainArgs(xp, ldx, K, we.data_ptr(), wqg.data_ptr(), q_out_features, vec.data_ptr(), float(ln_eps),
                      0 if x_out is None else x_out.data_ptr(), D, q_out.data_ptr(), q_out_features, N, D, int(rows_per_tile))
    _lib.check(_lib.load().anemoi_gt_rowchain_fwd(_lib.C.byref(a), _dt(x), _stream()), "gt_rowchain_fwd")
    return x_out, q_out


def gnn_edge_chain(e: Tensor, g1: Tensor, idx1: Tensor, g2: Tensor, idx2: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor, w2: Tensor,
                   b2: Tensor, ln_w: Tensor, ln_b: Optional[Tensor], eps: float) -> Tensor:
    """GraphConv's edge MLP (three Linears, gather-add form) + LayerNorm + residual in ONE launch (anemoi_gnn_edge_chain_fwd):
    ``LayerNorm(W2 gelu(W1 gelu(W0e e + g1[idx1] + g2[idx2] + b0) + b1) + b2) + e``.  ``w0, w1, w2``: fragment-major images
    (``pack_weight_frag``) of [512, 512] weights; ``idx1 / idx2`` int32.  Inference only."""
    _dev(e, g1, idx1, g2, idx2, w0, b0, w1, b1, w2, b2, ln_w, ln_b)
    M, D = e.shape
    dt = e.dtype
    if D != CHAIN_CHANNELS or dt not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"gnn_edge_chain: {D} channels / {dt} (built for {CHAIN_CHANNELS} channels, 16-bit dtypes)")
    for name, w in (("w0", w0), ("w1", w1), ("w2", w2)):
        if w.dim() != 1 or w.numel() != D * D or w.dtype != dt or not w.is_contiguous():
            raise ValueError(f"gnn_edge_chain: {name} must be the fragment-major image of a [{D}, {D}] weight (pack_weight_frag)")
    for name, ix in (("idx1", idx1), ("idx2", idx2)):
        if ix.dtype != torch.int32 or ix.dim() != 1 or ix.shape[0] != M or not ix.is_contiguous():
            raise ValueError(f"gnn_edge_chain: {name} must be contiguous int32 [{M}]")
    if g1.shape[1] != D or g2.shape[1] != D:
        raise ValueError("gnn_edge_chain: the gathered tables must have 512 columns")
    out = torch.empty((M, D), dtype=dt, device=e.device)
    (ep, lde), (p1, ld1), (p2, ld2) = _rows(e, "e", dt), _rows(g1, "g1", dt), _rows(g2, "g2", dt)
    _lib.check(_lib.load().anemoi_gnn_edge_chain_fwd(ep, lde, p1, ld1, idx1.data_ptr(), p2, ld2, idx2.data_ptr(), w0.data_ptr(), _vec(b0, "b0", D, dt),
                                                     w1.data_ptr(), _vec(b1, "b1", D, dt), w2.data_ptr(), _vec(b2, "b2", D, dt), _vec(ln_w, "ln_w", D, dt),
                                                     _vec(ln_b, "ln_b", D, dt), float(eps), out.data_ptr(), D, M, D, _dt(e), _stream()), "gnn_edge_chain_fwd")
    return out


def gnn_mlp_chain(x: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, ln_w: Tensor, ln_b: Optional[Tensor],
                  eps: float, residual: Optional[Tensor] = None) -> Tensor:
    """An embedding MLP in ONE launch (anemoi_gnn_mlp_chain_fwd): ``LayerNorm(W2 gelu(W1 gelu(W0 x + b0) + b1) + b2) [+ residual]``.
    x [N, K] with K in {128, 256, 384, 512} (zero-padded by the caller), w0 = pack_weight_frag of the [512, K] weight.  Inference only."""
    _dev(x, w0, b0, w1, b1, w2, b2, ln_w, ln_b, residual)
    N, K = x.shape
    D, dt = CHAIN_CHANNELS, x.dtype
    if dt not in (torch.bfloat16, torch.float16) or K % 128 or not 128 <= K <= D:
        raise NotImplementedError(f"gnn_mlp_chain: width {K} / {dt} (built for 16-bit rows of 128, 256, 384 or 512 columns)")
    for name, w, numel in (("w0", w0, D * K), ("w1", w1, D * D), ("w2", w2, D * D)):
        if w.dim() != 1 or w.numel() != numel or w.dtype != dt or not w.is_contiguous():
            raise ValueError(f"gnn_mlp_chain: {name} must be a contiguous fragment-major image of {numel} elements (pack_weight_frag)")
    if residual is not None and (tuple(residual.shape) != (N, D) or residual.dtype != dt):
        raise ValueError("gnn_mlp_chain: residual must be [N, 512] of x's dtype")
    out = torch.empty((N, D), dtype=dt, device=x.device)
    (xp, ldx), (rp, ldr) = _rows(x, "x", dt), _rows(residual, "residual", dt)
    _lib.check(_lib.load().anemoi_gnn_mlp_chain_fwd(xp, ldx, K, w0.data_ptr(), _vec(b0, "b0", D, dt), w1.data_ptr(), _vec(b1, "b1", D, dt), w2.data_ptr(),
                                                    _vec(b2, "b2", D, dt), _vec(ln_w, "ln_w", D, dt), _vec(ln_b, "ln_b", D, dt), float(eps), rp, ldr,
                                                    out.data_ptr(), D, N, D, _dt(x), _stream()), "gnn_mlp_chain_fwd")
    return out


def gnn_node_chain(x: Tensor, agg: Tensor, wa: Tensor, ba: Tensor, wb: Tensor, bb: Tensor, wc: Tensor, bc: Tensor, ln_w: Tensor,
                   ln_b: Optional[Tensor], eps: float, wt: Optional[Tensor] = None, bt: Optional[Tensor] = None, t_out_features: int = 0,
                   seg_ptr: Optional[Tensor] = None):
    """A GraphConv block's node MLP + LayerNorm + skip in ONE launch (anemoi_gnn_node_chain_fwd):
    ``x_out = LayerNorm(Wc gelu(Wb gelu(Wa [x | agg] + ba) + bb) + bc) + x`` and optionally ``t_out = x_out Wt^T [+ bt]``.
    Weights are fragment-major images (``pack_weight_frag``).  Returns ``x_out`` or ``(x_out, t_out)``.  Inference only."""
    _dev(x, agg, wa, ba, wb, bb, wc, bc, ln_w, ln_b, wt, bt)
    N, D = x.shape
    dt = x.dtype
    if D != CHAIN_CHANNELS or dt not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"gnn_node_chain: {D} channels / {dt} (built for {CHAIN_CHANNELS} channels, 16-bit dtypes)")
    if seg_ptr is not None:  # ``agg`` = the dst-sorted EDGE rows [M, 512]; the kernel forms their segmented sums (segment_sum_rows' arithmetic)
        _dev(seg_ptr)
        if agg.dim() != 2 or agg.shape[1] != D or seg_ptr.dtype != torch.int32 or tuple(seg_ptr.shape) != (N + 1,) or not seg_ptr.is_contiguous():
            raise ValueError("gnn_node_chain: with seg_ptr, agg must be the [M, 512] edge rows and seg_ptr contiguous int32 [N + 1]")
    elif tuple(agg.shape) != (N, D):
        raise ValueError("gnn_node_chain: x and agg must have the same [N, 512] shape")
    for name, w, numel in (("wa", wa, 2 * D * D), ("wb", wb, D * D), ("wc", wc, D * D)):
        if w.dim() != 1 or w.numel() != numel or w.dtype != dt or not w.is_contiguous():
            raise ValueError(f"gnn_node_chain: {name} must be a contiguous fragment-major image of {numel} elements (pack_weight_frag)")
    if wt is not None and (t_out_features % D or wt.dim() != 1 or wt.numel() != t_out_features * D or wt.dtype != dt or not wt.is_contiguous()):
        raise ValueError("gnn_node_chain: wt must be the fragment-major image of a [t_out_features, 512] weight, t_out_features % 512 == 0")
    x_out = torch.empty((N, D), dtype=dt, device=x.device)
    tf = t_out_features if wt is not None else 0
    t_out = torch.empty((N, tf), dtype=dt, device=x.device) if tf else None
    (xp, ldx), (ap, lda) = _rows(x, "x", dt), _rows(agg, "agg", dt)
    tail = (wa.data_ptr(), _vec(ba, "ba", D, dt), wb.data_ptr(), _vec(bb, "bb", D, dt), wc.data_ptr(),
            _vec(bc, "bc", D, dt), _vec(ln_w, "ln_w", D, dt), _vec(ln_b, "ln_b", D, dt), float(eps), x_out.data_ptr(), D,
            0 if wt is None else wt.data_ptr(), _vec(bt, "bt", tf, dt) if (bt is not None and tf) else 0, tf,
            0 if t_out is None else t_out.data_ptr(), tf, N, D, _dt(x), _stream())
    if seg_ptr is not None:
        _lib.check(_lib.load().anemoi_gnn_node_chain_segsum_fwd(xp, ldx, ap, lda, seg_ptr.data_ptr(), *tail), "gnn_node_chain_segsum_fwd")
    else:
        _lib.check(_lib.load().anemoi_gnn_node_chain_fwd(xp, ldx, ap, lda, *tail), "gnn_node_chain_fwd")
    return x_out if t_out is None else (x_out, t_out)


GLU_KINDS = {"glu": 0, "swiglu": 1, "geglu": 2, "reglu": 3}


def glu(gate_value: Tensor, kind: str) -> Tensor:
    """out = act(gate) * value for gate_value = [gate | value] [N, 2D]; kind in GLU_KINDS.  Differentiable."""
    if _needs_grad(gate_value):
        from .autograd import GluFunction

        return GluFunction.apply(gate_value, kind)
    return _glu_fwd(gate_value, kind)


def _glu_fwd(gate_value: Tensor, kind: str) -> Tensor:
    _dev(gate_value)
    N, D2 = gate_value.shape
    out = torch.empty((N, D2 // 2), dtype=gate_value.dtype, device=gate_value.device)
    p, ld = _rows(gate_value, "gate_value")
    _lib.check(_lib.load().anemoi_glu_fwd(p, ld, out.data_ptr(), D2 // 2, N, D2 // 2, GLU_KINDS[kind], _dt(gate_value), _stream()), "glu_fwd")
    return out


def glu_backward(gate_value: Tensor, d_out: Tensor, kind: str) -> Tensor:
    _dev(gate_value, d_out)
    N, D2 = gate_value.shape
    out = torch.empty((N, D2), dtype=gate_value.dtype, device=gate_value.device)
    (p, ld), (gp, ldg) = _rows(gate_value, "gate_value"), _rows(d_out, "d_out", gate_value.dtype)
    _lib.check(_lib.load().anemoi_glu_bwd(p, ld, gp, ldg, out.data_ptr(), D2, N, D2 // 2, GLU_KINDS[kind], _dt(gate_value), _stream()), "glu_bwd")
    return out


def assemble_input(x: Tensor, attrs: Optional[Tensor], width: int, col_mul: Optional[Tensor] = None, col_add: Optional[Tensor] = None,
                   out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """x [T, N, V] (time slices of one batch / ensemble member), attrs [N, A] -> [N, width] = [x[0] | ... | x[T-1] | attrs | 0...].
    With ``col_mul`` / ``col_add`` (fp32 [V]) the time / variable columns become x * mul[v] + add[v] (the input normaliser as a
    column program); ``out_dtype``: the model dtype when x is fp32 data fed to a 16-bit model (default: x's dtype)."""
    _dev(x, attrs, col_mul, col_add)
    T, N, V = x.shape
    A = 0 if attrs is None else attrs.shape[1]
    odt = x.dtype if out_dtype is None else out_dtype
    if x.stride(2) != 1 or width < T * V + A or (attrs is not None and (attrs.shape[0] != N or attrs.dtype != odt)):
        raise ValueError("assemble_input: x must be [T, N, V] with contiguous variables, attrs [N, A] of the output dtype")
    out = torch.empty((N, width), dtype=odt, device=x.device)
    ap, lda = _rows(attrs, "attrs", odt)
    plain_pad = T == 1 and A == 0 and odt != torch.float32 and width % 8 == 0  # cast + zero-pad of [N, V] rows: the division-free 16-byte path
    if col_mul is None and col_add is None and odt == x.dtype and not plain_pad:
        _lib.check(_lib.load().anemoi_assemble_input(x.data_ptr(), x.stride(0), x.stride(1), T, V, ap, lda, A, out.data_ptr(), width, width, N,
                                                     _dt(x), _stream()), "assemble_input")
        return out
    if (col_mul is None) != (col_add is None):
        raise ValueError("assemble_input: col_mul and col_add go together")
    if x.dtype != odt and x.dtype != torch.float32:
        raise ValueError("assemble_input: x must be in the output dtype or fp32")
    _lib.check(_lib.load().anemoi_assemble_input_norm(x.data_ptr(), _dt(x), x.stride(0), x.stride(1), T, V, _vec(col_mul, "col_mul", V, torch.float32),
                                                      _vec(col_add, "col_add", V, torch.float32), ap, lda, A, out.data_ptr(), width, width, N,
                                                      _DT[odt], _stream()), "assemble_input_norm")
    return out


def assemble_output(x_out: Tensor, x_skip: Tensor, col_map: Tensor, col_mul: Optional[Tensor] = None, col_add: Optional[Tensor] = None) -> Tensor:
    """out[n, v] = x_out[n, v] + x_skip[n, col_map[v]] (where col_map[v] >= 0); x_out [N, V_out], x_skip [N, V_in] (row
    stride allowed), col_map int32 [V_out].  With ``col_mul`` / ``col_add`` (fp32 [V_in]) the skip is the RAW input, normalised
    on the fly (x_skip * mul[m] + add[m]); the result has x_skip's dtype (the model dtype or fp32)."""
    _dev(x_out, x_skip, col_map, col_mul, col_add)
    N, V = x_out.shape
    out = torch.empty((N, V), dtype=x_skip.dtype, device=x_out.device)
    (xp, ldx), (sp, lds) = _rows(x_out, "x_out"), _rows(x_skip, "x_skip")
    if col_mul is None and col_add is None and x_skip.dtype == x_out.dtype:
        _lib.check(_lib.load().anemoi_assemble_output(xp, ldx, sp, lds, col_map.data_ptr(), out.data_ptr(), V, N, V, _dt(x_out), _stream()), "assemble_output")
        return out
    Vin = x_skip.shape[1]
    _lib.check(_lib.load().anemoi_assemble_output_norm(xp, ldx, _dt(x_out), sp, lds, col_map.data_ptr(), _vec(col_mul, "col_mul", Vin, torch.float32),
                                                       _vec(col_add, "col_add", Vin, torch.float32), out.data_ptr(), V, N, V, _dt(x_skip), _stream()),
               "assemble_output_norm")
    return out


def affine_columns(x: Tensor, col_mul: Tensor, col_add: Tensor, inverse: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """Per-variable affine map over the last dimension: x * mul + add, or (x - add) / mul with ``inverse`` (InputNormalizer
    transform / inverse_transform).  ``out`` may be x itself (in place)."""
    _dev(x, col_mul, col_add, out)
    V = x.shape[-1]
    if not x.is_contiguous():
        raise ValueError("affine_columns: x must be contiguous")
    y = torch.empty_like(x) if out is None else out
    if y.shape != x.shape or y.dtype != x.dtype or not y.is_contiguous():
        raise ValueError("affine_columns: out must match x")
    n = x.numel() // V
    _lib.check(_lib.load().anemoi_affine_columns(x.data_ptr(), V, y.data_ptr(), V, _vec(col_mul, "col_mul", V, torch.float32),
                                                 _vec(col_add, "col_add", V, torch.float32), 1 if inverse else 0, n, V, _dt(x), _stream()),
               "affine_columns")
    return y


def bound_columns_(x: Tensor, op_table: Tensor, param_table: Tensor) -> Tensor:
    """In place: apply the column program (int32 [n_ops, 4], fp32 [n_ops, 2]; see anemoi_bound_columns) to x [..., V]."""
    _dev(x, op_table, param_table)
    if not x.is_contiguous():
        raise ValueError("bound_columns_: x must be contiguous")
    V = x.shape[-1]
    _lib.check(_lib.load().anemoi_bound_columns(x.data_ptr(), V, x.numel() // V, V, op_table.data_ptr(), param_table.data_ptr(),
                                                op_table.shape[0], _dt(x), _stream()), "bound_columns")
    return x


def transpose_pad(x: Tensor, mult: int = 64) -> Tensor:
    """[N, C] -> contiguous [C, N_pad] with N_pad = N rounded up to ``mult`` and zeros in the padding."""
    _dev(x)
    n, c = x.shape
    n_pad = (n + mult - 1) // mult * mult
    out = torch.empty((c, n_pad), dtype=x.dtype, device=x.device)
    p, ld = _rows(x, "x")
    _lib.check(_lib.load().anemoi_transpose_pad(p, ld, out.data_ptr(), n_pad, n, c, n_pad, _dt(x), _stream()), "transpose_pad")
    return out


def gather_rows(x: Tensor, idx: Tensor) -> Tensor:
    """out[i] = x[idx[i]] (idx int32).  Differentiable (adjoint: rows summed back by index)."""
    if _needs_grad(x):
        from .autograd import GatherRowsFunction

        return GatherRowsFunction.apply(x, idx)
    return _gather_rows_fwd(x, idx)


def _gather_rows_fwd(x: Tensor, idx: Tensor) -> Tensor:
    _dev(x, idx)
    if idx.dtype != torch.int32 or idx.dim() != 1 or not idx.is_contiguous():
        raise ValueError("idx must be contiguous int32 [n]")
    D = x.shape[1]
    out = torch.empty((idx.shape[0], D), dtype=x.dtype, device=x.device)
    p, ld = _rows(x, "x")
    _lib.check(_lib.load().anemoi_gather_rows(p, ld, idx.data_ptr(), out.data_ptr(), D, idx.shape[0], D, _dt(x), _stream()), "gather_rows")
    return out


# ------------------------------------------------------------------------------------------ reference op mirror
@torch.library.custom_op("anemoi_amd::graph_transformer_attention", mutates_args=(), device_types="cuda")
def graph_transformer_attention(q: Tensor, k: Tensor, v: Tensor, e: Tensor, row: Tensor, colptr: Tensor, rowptr: Tensor,
                                edge_ids: Tensor, edge_dst: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """Same signature and outputs as the reference's ``anemoi::graph_transformer_attention``
    (triton/gt.py:390-428): q [N_dst,H,C], k,v [N_src,H,C], e [M,H,C] in CSC order, int64 row/colptr;
    returns (out in q.dtype, out_saved fp32, m = logsumexp fp32).  rowptr/edge_ids/edge_dst (reverse CSR)
    are only needed by the backward pass and ignored here."""
    N_dst, H, Cc = q.shape
    q2, k2, v2, e2 = (t.contiguous().view(t.shape[0], H * Cc) for t in (q, k, v, e))
    csc = CSC(row=row.to(torch.int32).contiguous(), dst=edge_dst.to(torch.int32).contiguous(),
              colptr=colptr.to(torch.int32).contiguous(), n_src=k.shape[0], n_dst=N_dst)
    if q.dtype != torch.float32:
        # the reference keeps the UNROUNDED fp32 accumulator as out_saved and rounds a copy for the caller (triton/gt.py:416-426): the
        # kernel's fp32 instantiation on the exactly upcast 16-bit operands produces that accumulator (same products, fp32 sums); the
        # module path (layers/block.py) does not come through here - it runs the fused-edge kernel in the model dtype
        saved, m = gt_attention(q2.float(), k2.float(), v2.float(), e2.float(), csc, H, return_lse=True)
        saved = saved.view(N_dst, H, Cc)
        return saved.to(q.dtype), saved, m
    out, m = gt_attention(q2, k2, v2, e2, csc, H, return_lse=True)
    out = out.view(N_dst, H, Cc)
    return out, out.clone(), m


@graph_transformer_attention.register_fake
def _graph_transformer_attention_fake(q, k, v, e, row, colptr, rowptr, edge_ids, edge_dst):
    N_dst, H, Cc = q.shape
    return (torch.empty((N_dst, H, Cc), device=q.device, dtype=q.dtype),
            torch.empty((N_dst, H, Cc), device=q.device, dtype=torch.float32),
            torch.empty((N_dst, H), device=q.device, dtype=torch.float32))


@torch.library.custom_op("anemoi_amd::graph_transformer_attention_backward", mutates_args=(), device_types="cuda")
def graph_transformer_attention_backward(d_out: Tensor, q: Tensor, k: Tensor, v: Tensor, e: Tensor, out_saved: Tensor, m: Tensor,
                                         row: Tensor, colptr: Tensor, rowptr: Tensor, edge_ids: Tensor,
                                         edge_dst: Tensor) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """Same signature as the reference's ``anemoi::graph_transformer_attention_backward`` (triton/gt.py:447-492):
    returns (dQ, dK, dV, dE) shaped like q, k, v, e.  For 16-bit operands the kernels take ``out_saved`` ROUNDED to the model dtype (the
    reference reads the fp32 accumulator in D = <dO, O>): one rounding of O, inside the stated 16-bit tolerance of the gradient tests
    (tests/test_attention_backward_gpu.py); fp32 operands use it as it is."""
    N_dst, H, Cc = q.shape
    flat = lambda t: t.contiguous().view(t.shape[0], H * Cc)  # noqa: E731
    csc = CSC(row=row.to(torch.int32).contiguous(), dst=edge_dst.to(torch.int32).contiguous(),
              colptr=colptr.to(torch.int32).contiguous(), n_src=k.shape[0], n_dst=N_dst)
    dq, dk, dv, de = gt_attention_backward(flat(d_out).to(q.dtype), flat(q), flat(k), flat(v), flat(e), flat(out_saved).to(q.dtype), m, csc,
                                           (rowptr, edge_ids, edge_dst), H)
    return dq.view_as(q), dk.view_as(k), dv.view_as(v), de.view_as(e)


@graph_transformer_attention_backward.register_fake
def _graph_transformer_attention_backward_fake(d_out, q, k, v, e, out_saved, m, row, colptr, rowptr, edge_ids, edge_dst):
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(e)


def _gta_setup_context(ctx, inputs, output):
    q, k, v, e, row, colptr, rowptr, edge_ids, edge_dst = inputs
    _out, out_saved, m = output
    ctx.save_for_backward(q, k, v, e, out_saved, m, row, colptr, rowptr, edge_ids, edge_dst)


def _gta_backward(ctx, d_out, _d_out_saved, _d_m):
    # only the gradient of the user-facing ``out`` is used (triton/gt.py:526-538)
    q, k, v, e, out_saved, m, row, colptr, rowptr, edge_ids, edge_dst = ctx.saved_tensors
    dq, dk, dv, de = graph_transformer_attention_backward(d_out, q, k, v, e, out_saved, m, row, colptr, rowptr, edge_ids, edge_dst)
    return dq, dk, dv, de, None, None, None, None, None


graph_transformer_attention.register_autograd(_gta_backward, setup_context=_gta_setup_context)


def graph_transformer_attention_conv(query: Tensor, key: Tensor, value: Tensor, edges: Tensor, csc: tuple[Tensor, Tensor],
                                     reverse: tuple[Tensor, Tensor, Tensor]) -> Tensor:
    """Drop-in for the reference's ``graph_transformer_attention_conv`` (triton/gt.py:564-576)."""
    row, colptr = csc
    rowptr, edge_ids, edge_dst = reverse
    out, _saved, _m = graph_transformer_attention(query, key, value, edges, row, colptr, rowptr, edge_ids, edge_dst)
    return out
