"""Model-parallel data movement on ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  Forward-only restatement of the reference's ``distributed/primitives.py`` / ``graph.py``
entry points that sit on the hot path: shard (local slice), gather (all-gather of unequal shards) and the halo
exchange (variable-count all-to-all: primitives.py:422-460).

MI355X notes: xGMI is point-to-point, so the halo exchange is ONE ``all_to_all_single`` on a packed send buffer
(each peer link carries only its own rows, all 7 links in parallel) rather than a ring collective, and the
mapper needs only the rows its edges touch rather than the reference's all-gather of every source row
(khop_edges.py:386-392) — see ``exchange_rows``.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor

from ..utils import segments
from ..utils.segments import collective
from .shapes import comm_rank, comm_size


# The two communication calls of the hot path.  Each is issued through ``collective`` so that a SegmentedGraph
# (utils/segments.py) can re-issue it between hipGraph segments; tests swap these two for a host-staged transport to
# run two ranks on ONE GPU (RCCL refuses two ranks per device).
def _all_to_all_single(recv: Tensor, send: Tensor, recv_counts, send_counts, group) -> None:
    collective(lambda: dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group))


def _all_gather_into_tensor(out: Tensor, inp: Tensor, group) -> None:
    collective(lambda: dist.all_gather_into_tensor(out, inp, group=group))


def _all_reduce_sum(x: Tensor, group) -> None:
    collective(lambda: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group))


def _push_rows_default(recv: Tensor, src: Tensor, send_index: Optional[Tensor], recv_counts, send_counts, group, gather_fn) -> None:
    packed = src if send_index is None else (gather_fn(src, send_index) if gather_fn is not None else src.index_select(0, send_index.long()))
    _all_to_all_single(recv, packed.contiguous(), list(recv_counts), list(send_counts), group)


# Indexed variant of the all-to-all (inference): recv <- the peers' rows, this rank's packed send order being src[send_index].
# Default: one pack kernel + the collective; the device-initiated wire (distributed/peer.py) does both in ONE kernel.
_push_rows = _push_rows_default


def recv_buffer(head_rows: int, send_counts, recv_counts, width: int, dtype, device, group) -> Tensor:
    """[head_rows + sum(recv_counts), width]: the caller fills the head (its local rows), the exchange that follows receives
    into the tail.  The device-initiated wire hands out memory its peers can store into; by default plain device memory."""
    return torch.empty((head_rows + sum(recv_counts), width), dtype=dtype, device=device)


def forward_scope(group):
    """Marks one forward of a model-parallel module (outermost scope wins).  A no-op for host-issued collectives; the
    device-initiated wire restarts its exchange sequence and lets all ranks meet on the device."""
    import contextlib

    return contextlib.nullcontext()


def scoped_forward(fn):
    """Decorator for the ``forward`` of a model-parallel module: runs it inside ``forward_scope(model_comm_group)``."""
    import functools
    import inspect

    pos = list(inspect.signature(fn).parameters).index("model_comm_group")

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        group = kwargs.get("model_comm_group") if "model_comm_group" in kwargs else (args[pos] if len(args) > pos else None)
        if group is None:
            return fn(*args, **kwargs)
        with forward_scope(group):
            return fn(*args, **kwargs)

    return wrapper


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ---------------------------------------------------------------------------------------------- forward primitives
def _shard(x: Tensor, dim: int, shard_sizes: Sequence[int], group) -> Tensor:
    start = sum(shard_sizes[: comm_rank(group)])
    return x.narrow(dim, start, shard_sizes[comm_rank(group)]).contiguous()


def _gather(x: Tensor, dim: int, shard_sizes: Sequence[int], group) -> Tensor:
    world = comm_size(group)
    x = x.contiguous()
    if dim != 0:
        x = x.transpose(0, dim).contiguous()
    if len(set(shard_sizes)) == 1:
        out = x.new_empty((sum(shard_sizes),) + tuple(x.shape[1:]))
        _all_gather_into_tensor(out, x, group)
    else:
        pad = max(shard_sizes)
        buf = x.new_zeros((pad,) + tuple(x.shape[1:]))
        buf[: x.shape[0]] = x
        outs = x.new_empty((world * pad,) + tuple(x.shape[1:]))
        _all_gather_into_tensor(outs, buf, group)
        out = torch.cat([outs[r * pad: r * pad + n] for r, n in enumerate(shard_sizes)], dim=0)
    if dim != 0:
        out = out.transpose(0, dim).contiguous()
    return out


def _a2a(send: Tensor, send_counts: Sequence[int], recv_counts: Sequence[int], group) -> Tensor:
    recv = send.new_empty((sum(recv_counts),) + tuple(send.shape[1:]))
    _all_to_all_single(recv, send.contiguous(), list(recv_counts), list(send_counts), group)
    return recv


# ---------------------------------------------------------------------------------------------- autograd (scope row f1)
# Convention: every rank back-propagates the loss terms of the rows IT OWNS; parameter gradients are therefore partial sums
# per rank and are completed by ``reduce_parameter_gradients`` (one all-reduce) after backward.  The adjoints below keep
# that invariant (reference graph.py:227-500, primitives.py:463-521):
#   shard (slice of a replicated tensor)      <- zero-expand        (the other ranks add their slices in the all-reduce)
#   gather, downstream replicated             <- slice              (_GatherParallelSection)
#   gather, downstream rank-specific ("sync") <- all-reduce + slice (_SyncParallelSection)
#   all-to-all of rows                        <- the reverse all-to-all (_HaloExchangeParallelSection / _halo_exchange_bwd)
class _ShardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dim, sizes, group):
        ctx.dim, ctx.sizes, ctx.group, ctx.full = dim, sizes, group, x.shape[dim]
        return _shard(x, dim, sizes, group)

    @staticmethod
    def backward(ctx, g):
        start = sum(ctx.sizes[: comm_rank(ctx.group)])
        shape = list(g.shape)
        shape[ctx.dim] = ctx.full
        out = g.new_zeros(shape)
        out.narrow(ctx.dim, start, g.shape[ctx.dim]).copy_(g)
        return out, None, None, None


class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dim, sizes, group, reduce_in_backward):
        ctx.dim, ctx.sizes, ctx.group, ctx.reduce = dim, sizes, group, reduce_in_backward
        return _gather(x, dim, sizes, group)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if ctx.reduce:
            g = g.clone()
            _all_reduce_sum(g, ctx.group)
        return _shard(g, ctx.dim, ctx.sizes, ctx.group), None, None, None, None


class _AllToAllRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, send, send_counts, recv_counts, group):
        ctx.send_counts, ctx.recv_counts, ctx.group = list(send_counts), list(recv_counts), group
        return _a2a(send, send_counts, recv_counts, group)

    @staticmethod
    def backward(ctx, g):
        return _a2a(g.contiguous(), ctx.recv_counts, ctx.send_counts, ctx.group), None, None, None


def reduce_parameter_gradients(module: torch.nn.Module, group) -> None:
    """Complete the per-rank partial parameter gradients of a model-parallel backward (sum over the model group)."""
    if comm_size(group) == 1:
        return
    for p in module.parameters():
        if p.grad is not None:
            _all_reduce_sum(p.grad, group)


def shard_tensor(x: Tensor, dim: int, shard_sizes: Sequence[int], group) -> Tensor:
    """Local slice of a replicated tensor (no communication): graph.py:66-103 / primitives.py:24-57."""
    if comm_size(group) == 1:
        return x
    if _needs_grad(x):
        return _ShardFn.apply(x, dim, list(shard_sizes), group)
    return _shard(x, dim, shard_sizes, group)


def gather_tensor(x: Tensor, dim: int, shard_sizes: Sequence[int], group, reduce_in_backward: bool = False) -> Tensor:
    """All-gather shards of possibly unequal size along ``dim``: primitives.py:60-183.  ``reduce_in_backward``: what
    follows is rank-specific work on the full tensor (the reference's ``sync_tensor``), so the gradients of all ranks
    are summed before the local slice is taken."""
    if comm_size(group) == 1:
        return x
    if _needs_grad(x):
        return _GatherFn.apply(x, dim, list(shard_sizes), group, reduce_in_backward)
    return _gather(x, dim, shard_sizes, group)


def all_to_all_rows(send: Tensor, send_counts: Sequence[int], recv_counts: Sequence[int], group) -> Tensor:
    """Variable-count all-to-all over dim 0 of a packed [sum(send_counts), ...] buffer (differentiable)."""
    if _needs_grad(send):
        return _AllToAllRowsFn.apply(send, list(send_counts), list(recv_counts), group)
    return _a2a(send, send_counts, recv_counts, group)


def halo_exchange(x: Tensor, send_index: Tensor, send_counts: Sequence[int], recv_counts: Sequence[int], group,
                  gather_fn=None) -> Tensor:
    """[n_local, D] -> [n_local + n_halo, D]: pack x[send_index] (one gather kernel), one all_to_all_single, append.

    ``send_index``: concatenation over peers of the local row ids to send (int32 on device for the HIP gather).
    ``gather_fn(x, idx)`` packs the send buffer (ops.gather_rows on the GPU)."""
    if comm_size(group) == 1:
        return x
    packed = gather_fn(x, send_index) if gather_fn is not None else x.index_select(0, send_index.long())
    recv = all_to_all_rows(packed, send_counts, recv_counts, group)
    return torch.cat([x, recv], dim=0)


def halo_exchange_into(buf: Tensor, n_local: int, send_index: Tensor, send_counts: Sequence[int], recv_counts: Sequence[int], group,
                       gather_fn) -> Tensor:
    """Inference variant of ``halo_exchange`` without the concatenation: ``buf`` [n_local + n_halo, D] already holds the local
    rows in its head; the halo rows are received straight into its tail."""
    _push_rows(buf[n_local:], buf[:n_local], send_index, list(recv_counts), list(send_counts), group, gather_fn)
    return buf


def exchange_rows(x_local: Tensor, want_global_ids: Tensor, shard_sizes: Sequence[int], group, gather_fn=None,
                  plan: Optional[dict] = None):
    """Needed-rows exchange: every rank holds a contiguous shard of a [N, D] table and obtains the rows
    ``want_global_ids`` (ascending global ids, any owner).  Two all-to-alls at plan time (ids), one per call (rows).
    Returns (rows [len(want), D], plan) — pass ``plan`` back in to skip the id exchange (static graph)."""
    world, rank = comm_size(group), comm_rank(group)
    if world == 1:
        rows = gather_fn(x_local, want_global_ids.to(torch.int32)) if gather_fn is not None else x_local.index_select(0, want_global_ids.long())
        return rows, plan
    if plan is None:
        if segments._ACTIVE is not None:
            raise RuntimeError("exchange_rows: the needed-rows plan must exist before a SegmentedGraph capture (run a warm-up forward first)")
        bounds = torch.cumsum(torch.tensor(shard_sizes, dtype=torch.long, device=want_global_ids.device), 0)
        owner = torch.searchsorted(bounds, want_global_ids.long(), right=True)
        recv_counts = torch.bincount(owner, minlength=world)
        send_counts = torch.empty_like(recv_counts)
        dist.all_to_all_single(send_counts, recv_counts, group=group)
        recv_counts_l, send_counts_l = recv_counts.tolist(), send_counts.tolist()
        # tell every owner which of its rows we want (ids are ascending, hence grouped by owner already)
        starts = torch.cat([bounds.new_zeros(1), bounds[:-1]])
        req_local = want_global_ids.long() - starts[owner]
        asked = req_local.new_empty(sum(send_counts_l))
        dist.all_to_all_single(asked, req_local.contiguous(), output_split_sizes=send_counts_l, input_split_sizes=recv_counts_l, group=group)
        plan = dict(send_index=asked.to(torch.int32).contiguous(), send_counts=send_counts_l, recv_counts=recv_counts_l)
    if not _needs_grad(x_local) and x_local.dim() == 2:
        # inference: pack + exchange as one indexed push (one kernel on the device-initiated wire), received in place
        rows = recv_buffer(0, plan["send_counts"], plan["recv_counts"], x_local.shape[1], x_local.dtype, x_local.device, group)
        _push_rows(rows, x_local, plan["send_index"], plan["recv_counts"], plan["send_counts"], group, gather_fn)
        return rows, plan
    packed = gather_fn(x_local, plan["send_index"]) if gather_fn is not None else x_local.index_select(0, plan["send_index"].long())
    rows = all_to_all_rows(packed, plan["send_counts"], plan["recv_counts"], group)
    return rows, plan
