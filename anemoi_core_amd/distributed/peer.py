"""Device-initiated row exchange over hipIpc-mapped peer memory — the product wire of the sharded INFERENCE forward.

What the reference does with host-issued NCCL collectives (halo all-to-all per processor layer, distributed/primitives.py:
422-460 / layers/block.py:1159-1172; all-gather of shards, primitives.py:60-183; source-row sync of the mappers,
distributed/khop_edges.py:386-392) is here ONE kernel per exchange on the rank's own stream (csrc/peer.hip): rows are stored
straight into the peers' receive buffers, published with an epoch flag, and the peers' flags awaited.  Nothing of it involves
the host, so a rank's whole sharded forward is captured as ONE hipGraph (RCCL collectives cannot be captured on this stack and
force a chain of 19 graphs with 18 host-issued collectives, utils/segments.py).

``install(group)`` swaps the wire under ``distributed/primitives.py`` (the three communication calls of the hot path and the
two hooks ``recv_buffer`` / ``forward_scope``); everything above — partitions, halo plans, needed-rows plans, packing order —
is unchanged.  Training keeps the differentiable RCCL path (``all_to_all_rows`` and friends route here only without grad).

Channels.  Every exchange of a forward is a *channel*: a receive region in this rank's payload arena, P flag words + (seq,
ticket) in its flag block, and a device table with where this rank's rows go in each peer's region.  Channels are created
collectively (one ``all_gather_object`` of region offsets) the first time a forward runs and are looked up by their position in
the forward afterwards (SPMD: every rank issues the same sequence).  ``forward_scope`` marks the start of a forward: position 0
and a barrier (an exchange without rows), which is what makes re-using a channel's receive region once per forward safe.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor

from .. import _lib
from . import primitives as P
from .shapes import comm_rank, comm_size

_ALIGN = 256
_HDR_WORDS = 16  # word 0 = status
_TICKS_PER_S = 100_000_000  # wall_clock64 on gfx950


class _Raw:
    """Device memory that torch did not allocate, as a __cuda_array_interface__ object (zero-copy ``torch.as_tensor``)."""

    def __init__(self, ptr: int, nbytes: int, owner=None) -> None:
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


def _view(ptr: int, nbytes: int, device, owner=None) -> Tensor:
    return torch.as_tensor(_Raw(ptr, nbytes, owner), device=device)


class PeerWireError(RuntimeError):
    pass


class _Channel:
    __slots__ = ("sig", "table", "flags_ptr", "region", "region_off", "row_bytes", "recv_rows", "head_rows", "send_rows", "buffer", "index")


class PeerWire:
    def __init__(self, group, arena_mb: Optional[int] = None, max_channels: int = 512, timeout_s: Optional[float] = None) -> None:
        self.group, self.world, self.rank = group, comm_size(group), comm_rank(group)
        if self.world < 2:
            raise PeerWireError("PeerWire needs a group of at least 2 ranks")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.load()
        self.arena_bytes = int(arena_mb if arena_mb is not None else int(os.environ.get("ANEMOI_PEER_ARENA_MB", "512"))) << 20
        self.max_channels = max_channels
        self.stride = self.world + 4  # flags[P], seq, ticket, 2 spare words
        self.flag_bytes = 4 * (_HDR_WORDS + max_channels * self.stride)
        self.timeout_ticks = int(float(timeout_s if timeout_s is not None else os.environ.get("ANEMOI_PEER_TIMEOUT_S", "20")) * _TICKS_PER_S)
        kind = {"default": 0, "fine": 1, "uncached": 2}[os.environ.get("ANEMOI_PEER_PAYLOAD", "default")]
        self.payload_ptr = self.flag_ptr = 0
        self.peer_payload, self.peer_flags, self._opened = [0] * self.world, [0] * self.world, []
        # set-up is collective: a rank that fails must not leave the others waiting in the next collective, so every phase ends
        # with an exchange of (result | error) and ALL ranks raise together
        try:
            self.payload_ptr = self._alloc(self.arena_bytes, kind)
            self.flag_ptr = self._alloc(self.flag_bytes, 2)
            mine = (self._export(self.payload_ptr), self._export(self.flag_ptr), os.getpid(), None)
        except Exception as e:  # noqa: BLE001
            mine = (None, None, os.getpid(), f"{type(e).__name__}: {e}")
        everyone = self._agree(mine, "allocating / exporting the arenas")
        err = None
        try:
            for p, (hp, hf, pid, _) in enumerate(everyone):
                if p == self.rank:
                    self.peer_payload[p], self.peer_flags[p] = self.payload_ptr, self.flag_ptr
                    continue
                if pid == os.getpid():
                    raise PeerWireError("two ranks of the group live in one process: hipIpc handles cannot be opened by their exporter")
                self.peer_payload[p], self.peer_flags[p] = self._open(hp), self._open(hf)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        self._agree((None, None, os.getpid(), err), "opening the peers' arenas")
        self.payload = _view(self.payload_ptr, self.arena_bytes, self.device, self)
        self.status = _view(self.flag_ptr, 4, self.device, self).view(torch.int32)
        self._diag = _view(self.flag_ptr, 4 * _HDR_WORDS, self.device, self).view(torch.int32)  # words 2-4: exchange diagnostics (csrc/peer.hip)
        self._bump = 0
        self._channels: list = []
        self._seq = 0
        self._depth = 0
        self._pending: Optional[_Channel] = None  # set by recv_buffer(), consumed by the exchange into that buffer
        self._barrier_ch = self._new_channel(("barrier",), [0] * self.world, [0] * self.world, 16, 0, barrier=True)
        dist.barrier(group=group)  # every rank has opened every arena before anyone stores into one

    def _agree(self, mine: tuple, what: str) -> list:
        everyone: list = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        bad = [(p, e[3]) for p, e in enumerate(everyone) if e[3] is not None]
        if bad:
            self.close()
            raise PeerWireError(f"peer wire set-up failed while {what}: " + "; ".join(f"rank {p}: {m}" for p, m in bad))
        return everyone

    # ------------------------------------------------------------------------------------------------ memory
    def _alloc(self, nbytes: int, kind: int) -> int:
        out = C.c_void_p()
        _lib.check(self.lib.anemoi_peer_alloc(C.byref(out), nbytes, kind), "peer_alloc")
        return int(out.value)

    def _export(self, ptr: int) -> bytes:
        buf = C.create_string_buffer(64)
        _lib.check(self.lib.anemoi_peer_export(ptr, buf), "peer_export")
        return bytes(buf.raw)

    def _open(self, handle: bytes) -> int:
        out = C.c_void_p()
        _lib.check(self.lib.anemoi_peer_open(C.create_string_buffer(handle, 64), C.byref(out)), "peer_open")
        self._opened.append(int(out.value))
        return int(out.value)

    def close(self) -> None:
        """Unmap the peers' arenas and free this rank's (call on every rank, after the last exchange of ALL ranks)."""
        torch.cuda.synchronize()
        for ptr in self._opened:
            self.lib.anemoi_peer_close(ptr)
        self._opened = []
        for name in ("payload_ptr", "flag_ptr"):
            if getattr(self, name, 0):
                self.lib.anemoi_peer_free(getattr(self, name))
                setattr(self, name, 0)

    def owns(self, t: Tensor) -> bool:
        return self.payload_ptr <= t.data_ptr() < self.payload_ptr + self.arena_bytes

    # ------------------------------------------------------------------------------------------------ channels
    def _new_channel(self, sig, send_counts: Sequence[int], recv_counts: Sequence[int], row_bytes: int, head_rows: int,
                     barrier: bool = False) -> _Channel:
        from ..utils import segments

        if segments._ACTIVE is not None or torch.cuda.is_current_stream_capturing():
            raise PeerWireError("a peer channel must exist before graph capture (run a warm-up forward first)")
        cid = len(self._channels)
        if cid >= self.max_channels:
            raise PeerWireError(f"more than {self.max_channels} exchanges in one forward scope (is distributed.primitives.forward_scope missing around the caller?)")
        W, me = self.world, self.rank
        recv_rows = int(sum(recv_counts))
        nbytes = (head_rows + recv_rows) * row_bytes
        off = (self._bump + _ALIGN - 1) // _ALIGN * _ALIGN
        if off + nbytes > self.arena_bytes:
            raise PeerWireError(f"peer arena exhausted ({self.arena_bytes >> 20} MiB; ANEMOI_PEER_ARENA_MB)")
        self._bump = off + nbytes
        # where the rows of source q start in MY region; every rank learns where ITS rows go at every peer
        starts, run = [], off + head_rows * row_bytes
        for q in range(W):
            starts.append(run)
            run += int(recv_counts[q]) * row_bytes
        everyone: list = [None] * W
        dist.all_gather_object(everyone, (sig[0], row_bytes, starts, [int(c) for c in recv_counts]), group=self.group)
        for p, (kind, rb, _, rc) in enumerate(everyone):
            if kind != sig[0] or rb != row_bytes or rc[me] != int(send_counts[p]):
                raise PeerWireError(f"exchange {cid}: rank {p} expects {rc[me]} rows of {rb} bytes ({kind}) from rank {me}, which sends "
                                    f"{int(send_counts[p])} rows of {row_bytes} bytes ({sig[0]})")
        flag_word = _HDR_WORDS + cid * self.stride
        tab = torch.zeros((6, W), dtype=torch.int64)
        begin = 0
        for p in range(W):
            tab[0, p] = self.peer_payload[p] + everyone[p][2][me]
            tab[1, p] = self.peer_flags[p] + 4 * (flag_word + me)
            tab[2, p], tab[3, p] = begin, int(send_counts[p])
            begin += int(send_counts[p])
            tab[4, p] = int(p != me and (barrier or send_counts[p] > 0))
            tab[5, p] = int(p != me and (barrier or recv_counts[p] > 0))
        ch = _Channel()
        ch.sig, ch.table, ch.flags_ptr = sig, tab.to(self.device), self.flag_ptr + 4 * flag_word
        ch.region_off, ch.row_bytes, ch.recv_rows, ch.head_rows, ch.send_rows = off, row_bytes, recv_rows, head_rows, begin
        ch.region = self.payload[off: off + nbytes]
        ch.buffer = ch.index = None
        self._channels.append(ch)
        return ch

    def _next_channel(self, sig, send_counts, recv_counts, row_bytes: int, head_rows: int = 0) -> _Channel:
        """The channel at the current position of the forward: created on first use, checked against ``sig`` afterwards."""
        self._seq += 1
        if self._seq < len(self._channels):
            ch = self._channels[self._seq]
            if ch.sig != sig:
                raise PeerWireError(f"exchange {self._seq} of this forward was {ch.sig} when the channels were created and is {sig} now; "
                                    "the sequence of exchanges of a forward must not change (call PeerWire.reset() after re-partitioning)")
            return ch
        if self._seq != len(self._channels):
            raise PeerWireError("internal: channel sequence out of order")
        return self._new_channel(sig, send_counts, recv_counts, row_bytes, head_rows)

    def reset(self) -> None:
        """Forget all channels (collective; after the partition or the model changed)."""
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        del self._channels[1:]
        self._bump = 0
        self._seq = 0

    def _launch(self, ch: _Channel, src: Optional[Tensor], ld_bytes: int, send_index: Optional[Tensor]) -> None:
        rc = self.lib.anemoi_peer_exchange_rows(
            0 if src is None else src.data_ptr(), ld_bytes, 0 if send_index is None else send_index.data_ptr(), ch.table.data_ptr(), self.world,
            ch.row_bytes, ch.send_rows, ch.flags_ptr, self.flag_ptr, self.timeout_ticks, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "peer_exchange_rows")

    # ------------------------------------------------------------------------------------------------ the wire
    @contextlib.contextmanager
    def forward_scope(self):
        """Outermost scope = one forward: exchange positions restart and all ranks meet (stream-ordered, on the device)."""
        self._depth += 1
        try:
            if self._depth == 1:
                self._seq = 0
                if len(self._channels) == 1:
                    # the first forward after install / reset: the ranks arrive from host-side set-up (model build, plans) that
                    # may differ by many seconds, and the device barrier below has a time-out - meet on the host first.  Later
                    # forwards are kept in step by their own exchanges.
                    dist.barrier(group=self.group)
                self._launch(self._barrier_ch, None, 0, None)
            yield
        finally:
            self._depth -= 1

    def recv_buffer(self, head_rows: int, send_counts, recv_counts, width: int, dtype) -> Tensor:
        """[head_rows + sum(recv_counts), width] living in the arena: the caller fills the head, the exchange that follows
        (``push_rows`` with this tensor's tail as ``recv``) lets the peers fill the tail in place."""
        es = torch.empty((), dtype=dtype).element_size()
        sig = ("rows", tuple(int(c) for c in send_counts), tuple(int(c) for c in recv_counts), width * es, int(head_rows))
        ch = self._next_channel(sig, send_counts, recv_counts, width * es, head_rows)
        if ch.buffer is None or ch.buffer.dtype != dtype:
            ch.buffer = ch.region.view(dtype).view(head_rows + ch.recv_rows, width)
        self._pending = ch
        return ch.buffer

    def push_rows(self, recv: Tensor, src: Tensor, send_index: Optional[Tensor], recv_counts, send_counts) -> None:
        """recv[sum(recv_counts), ...] <- the rows the peers hold for this rank; packed row i of this rank's send order is
        src[send_index[i]] (src[i] without an index).  One kernel; ``recv`` in the arena is filled in place, any other tensor
        through the channel's region and one copy."""
        if src.dim() != 2 or src.stride(1) != 1:
            src = src.reshape(src.shape[0], -1).contiguous()
        es = src.element_size()
        row_bytes = src.shape[1] * es
        if row_bytes % 16 or (src.stride(0) * es) % 16 or src.data_ptr() % 16:
            raise PeerWireError(f"peer exchange: rows of {row_bytes} bytes (stride {src.stride(0) * es}) are not 16-byte multiples")
        ch, self._pending = self._pending, None
        direct = ch is not None and (recv.numel() == 0 or (self.owns(recv) and recv.data_ptr() == ch.region.data_ptr() + ch.head_rows * ch.row_bytes))
        if not direct:
            if ch is not None:
                raise PeerWireError("recv_buffer() must be followed by the exchange into that buffer")
            sig = ("rows", tuple(int(c) for c in send_counts), tuple(int(c) for c in recv_counts), row_bytes, 0)
            ch = self._next_channel(sig, send_counts, recv_counts, row_bytes, 0)
        if send_index is None and src.shape[0] != ch.send_rows:
            raise PeerWireError(f"peer exchange: {src.shape[0]} packed rows, the plan sends {ch.send_rows}")
        self._launch(ch, src, src.stride(0) * es, send_index)
        if not direct and ch.recv_rows:
            if not recv.is_contiguous():  # reshape() of a strided tensor is a temporary: the received rows would be dropped silently
                raise PeerWireError("peer exchange: a staged receive buffer must be contiguous")
            recv.view(ch.recv_rows, -1).copy_(ch.region.view(recv.dtype).view(ch.recv_rows, -1))

    def all_gather(self, out: Tensor, inp: Tensor) -> None:
        """out[W * n, ...] <- every rank's inp[n, ...] (equal shards).  Shards whose rows are not 16-byte multiples (84 bf16
        output variables = 168 bytes) travel as flat runs of 16-byte pieces: the concatenation of shards is the same bytes."""
        n = inp.shape[0]
        src = inp.reshape(n, -1)
        if not src.is_contiguous():
            src = src.contiguous()
        row_bytes = src.shape[1] * src.element_size()
        if row_bytes % 16:
            if (n * row_bytes) % 16 or src.data_ptr() % 16:
                raise PeerWireError(f"peer all_gather: a shard of {n} x {row_bytes} bytes is not a 16-byte multiple")
            src = src.view(torch.uint8).reshape(-1, 16)
            n, row_bytes = src.shape[0], 16
        counts = [n] * self.world
        ch = self._next_channel(("gather", n, row_bytes), counts, counts, row_bytes, 0)
        if ch.index is None:  # every peer gets the SAME n rows: the packed send order repeats them W times
            ch.index = torch.arange(n, dtype=torch.int32, device=self.device).repeat(self.world)
        self._launch(ch, src, src.stride(0) * src.element_size(), ch.index)
        if not out.is_contiguous():  # reshape() of a strided tensor is a temporary: the rows would be dropped silently
            raise PeerWireError("peer all_gather: the output tensor must be contiguous")
        out.view(-1).view(torch.uint8).copy_(ch.region)

    def selftest(self) -> None:
        """Collective: a small all-gather and a variable-count exchange over the wire, every word checked, verdict agreed by all
        ranks - a wire that loses or delays rows is found HERE (PeerWireError on every rank, the caller keeps RCCL) and not in the
        middle of a model's forward.  Leaves no channels behind."""
        W, me, dev = self.world, self.rank, self.device
        rows = 48
        pat = lambda r, n: (torch.arange(n * 64, device=dev, dtype=torch.int32).view(n, 64) * (r + 3) + 17 * r)  # noqa: E731
        err = None
        try:
            with torch.no_grad():
                for _ in range(2):  # twice: the second pass reuses the channels (epochs advance, regions are overwritten)
                    with self.forward_scope():
                        gathered = torch.empty(W * rows, 64, dtype=torch.int32, device=dev)
                        self.all_gather(gathered, pat(me, rows))
                        counts = [0 if p == me else 1 + (me + p) % 5 for p in range(W)]  # symmetric in (me, p)
                        recv = torch.empty(sum(counts), 64, dtype=torch.int32, device=dev)
                        self.push_rows(recv, pat(me, 8), torch.cat([torch.arange(c, dtype=torch.int32, device=dev) for c in counts]), counts, counts)
                    self.check()
                    want = torch.cat([pat(p, 8)[: counts[p]] for p in range(W)])
                    if not (torch.equal(gathered, torch.cat([pat(p, rows) for p in range(W)])) and torch.equal(recv, want)):
                        raise PeerWireError(f"rank {me}: rows received over the hipIpc wire differ from what the peers sent")
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        everyone: list = [None] * W
        dist.all_gather_object(everyone, err, group=self.group)
        self.reset()
        bad = [(p, e) for p, e in enumerate(everyone) if e is not None]
        if bad:
            raise PeerWireError("hipIpc wire self-test failed: " + "; ".join(f"rank {p}: {m}" for p, m in bad))

    def stats(self, reset: bool = True) -> dict:
        """Exchange diagnostics since the last reset (synchronises): exchanges run, the time between "my rows released" and "every
        expected peer's flag seen" summed over them and the longest one [us], and the peer of a time-out (None: none)."""
        torch.cuda.synchronize()
        w = [int(v) & 0xFFFFFFFF for v in self._diag[:5].tolist()]
        if reset:
            self._diag[2:5].zero_()
            torch.cuda.synchronize()
        us = 1e6 / _TICKS_PER_S
        return {"exchanges": w[2], "wait_us_total": round(w[3] * us, 1), "wait_us_max": round(w[4] * us, 1),
                "timeout_peer": (w[0] & 0xFFFF) if w[0] else None}

    def check(self) -> None:
        """Raise if a wait of this rank timed out (synchronises)."""
        torch.cuda.synchronize()
        st = int(self.status.item()) & 0xFFFFFFFF
        if st:
            raise PeerWireError(f"rank {self.rank}: no flag from peer {st & 0xFFFF} within {self.timeout_ticks / _TICKS_PER_S:.0f} s")


_WIRE: Optional[PeerWire] = None


def current() -> Optional[PeerWire]:
    return _WIRE


def install(group, **kwargs) -> PeerWire:
    """Make the device-initiated exchange the wire of ``group``'s inference forward (collective: every rank calls it)."""
    global _WIRE
    wire = PeerWire(group, **kwargs)
    try:
        wire.selftest()
    except PeerWireError:
        wire.close()
        raise
    _WIRE = wire
    fallback = (P._all_to_all_single, P._all_gather_into_tensor)

    def mine(g, *tensors) -> bool:
        # ONE predicate for all five hooks: the wire carries the inference forward only (grad mode off).  With grad mode on every
        # exchange - also one whose rows happen not to require grad, e.g. the encoder's raw input rows in a training step - stays on
        # RCCL: outside a forward scope the channel sequence is never rewound, and every step would create new channels.
        return g is wire.group and not torch.is_grad_enabled()

    def a2a(recv, send, recv_counts, send_counts, g):
        if mine(g, send) and (send.shape[1:].numel() * send.element_size()) % 16 == 0:
            return wire.push_rows(recv, send, None, recv_counts, send_counts)
        return fallback[0](recv, send, recv_counts, send_counts, g)

    def push(recv, src, send_index, recv_counts, send_counts, g, gather_fn):
        if mine(g, src) and (src.shape[1:].numel() * src.element_size()) % 16 == 0:
            return wire.push_rows(recv, src, send_index, recv_counts, send_counts)
        return P._push_rows_default(recv, src, send_index, recv_counts, send_counts, g, gather_fn)

    def allgather(out, inp, g):
        if mine(g, inp) and inp.shape[0] > 0 and (inp.numel() * inp.element_size()) % 16 == 0 and inp.data_ptr() % 16 == 0:
            return wire.all_gather(out, inp)
        return fallback[1](out, inp, g)

    def recv_buffer(head_rows, send_counts, recv_counts, width, dtype, device, g):
        if mine(g) and (width * torch.empty((), dtype=dtype).element_size()) % 16 == 0:
            return wire.recv_buffer(head_rows, send_counts, recv_counts, width, dtype)
        return torch.empty((head_rows + sum(recv_counts), width), dtype=dtype, device=device)

    def scope(g):
        return wire.forward_scope() if mine(g) else contextlib.nullcontext()

    wire._saved = (P._all_to_all_single, P._all_gather_into_tensor, P._push_rows, P.recv_buffer, P.forward_scope)
    P._all_to_all_single, P._all_gather_into_tensor, P._push_rows, P.recv_buffer, P.forward_scope = a2a, allgather, push, recv_buffer, scope
    return wire


def uninstall() -> None:
    """Back to the wire that was in place before ``install`` (the RCCL collectives): collective, every rank calls it."""
    global _WIRE
    if _WIRE is None:
        return
    wire, _WIRE = _WIRE, None
    P._all_to_all_single, P._all_gather_into_tensor, P._push_rows, P.recv_buffer, P.forward_scope = wire._saved
    torch.cuda.synchronize()
    dist.barrier(group=wire.group)  # nobody unmaps while a peer may still store
    wire.close()
