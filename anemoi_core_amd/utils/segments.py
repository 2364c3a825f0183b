"""Segmented hipGraph capture of a forward that contains collectives.

RCCL collectives cannot be captured into a hipGraph on this stack (tools/nccl_capture_probe.py: the capture aborts or
hangs), yet at N > 1 GPUs the per-rank kernels shrink to a few microseconds each and an eagerly launched forward becomes
host-bound (~10 us of Python/ctypes per kernel).  ``SegmentedGraph`` therefore records the forward as a CHAIN: every
stretch of kernels between two collectives becomes one hipGraph, and the collectives themselves are kept as closures on
fixed buffers that are re-issued eagerly, in order, between the graph launches.  All segments allocate from one private
memory pool, so a tensor produced in segment k is still at the same address when segment k+1 (or the collective in
between) consumes it on replay.

Contract for code that runs under ``capture``: every collective goes through ``collective(fn)`` with ``fn`` a
zero-argument closure that issues exactly the communication call on tensors that already exist (no allocation, no
kernel launch of its own) — ``distributed/primitives.py`` does that for all of them.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

_ACTIVE: Optional["SegmentedGraph"] = None


def collective(fn: Callable[[], None]) -> None:
    """Run a communication call; under an active ``SegmentedGraph.capture`` it also becomes a segment boundary."""
    rec = _ACTIVE
    if rec is None:
        fn()
        return
    rec._end_segment()
    fn()
    rec._chain.append(fn)
    rec._begin_segment()


class SegmentedGraph:
    def __init__(self) -> None:
        self._chain: List[Callable[[], None]] = []  # graph.replay and collective closures, in issue order
        self._graphs: List[torch.cuda.CUDAGraph] = []
        self._pool = None
        self._current: Optional[torch.cuda.CUDAGraph] = None
        self.output = None

    # -- recording --------------------------------------------------------------------------------------------
    def _begin_segment(self) -> None:
        torch.cuda.synchronize()  # nothing of the collective (or of its watchdog's event polling) overlaps a capture
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._pool, capture_error_mode="thread_local")
        self._current = g

    def _end_segment(self) -> None:
        g, self._current = self._current, None
        g.capture_end()
        self._graphs.append(g)
        self._chain.append(g.replay)

    def capture(self, fn: Callable[[], object]):
        """Record ``fn()`` (already warmed up: static caches built, plans exchanged).  Returns ``fn``'s result, whose
        tensors are overwritten in place by every ``replay``."""
        global _ACTIVE
        assert _ACTIVE is None, "nested SegmentedGraph capture"
        self._pool = torch.cuda.graph_pool_handle()
        torch.cuda.synchronize()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            _ACTIVE = self
            try:
                self._begin_segment()
                try:
                    self.output = fn()
                finally:
                    if self._current is not None:
                        self._end_segment()
            finally:
                _ACTIVE = None
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        return self.output

    # -- replay -----------------------------------------------------------------------------------------------
    @property
    def num_graphs(self) -> int:
        return len(self._graphs)

    @property
    def num_collectives(self) -> int:
        return len(self._chain) - len(self._graphs)

    def replay(self):
        for item in self._chain:
            item()
        return self.output
