#!/usr/bin/env python
"""Cost of one device-initiated halo exchange (csrc/peer.hip), measured between P processes on ONE GPU.

    python tools/peer_latency.py [--world 2] [--rows 188] [--layers 16] [--replays 200]

Every rank captures ONE hipGraph = forward barrier + `layers` exchanges of `rows` rows x 1 KiB to each neighbour (ring
neighbours r-1, r+1, as the latitude-band partition gives) into in-place receive buffers, replays it `replays` times and reports
the device time per exchange (HIP events around the replays, max over ranks).  With nothing else in the graph the ranks run in
lock step, so the figure is the full round trip: stores into the peer's buffer, system-scope release, flag, the peer's flag seen.
On one device the "link" is the local fabric; across MI355X devices the stores travel over xGMI instead (same kernel, same
protocol) - this tool runs unchanged under torchrun with one rank per device (ANEMOI_PEER_LATENCY_MULTI=1)."""
import argparse
import os
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, init_file, args):
    sys.path.insert(0, REPO)
    multi = os.environ.get("ANEMOI_PEER_LATENCY_MULTI") == "1"
    torch.cuda.set_device(rank if multi else 0)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from anemoi_core_amd import ops
    from anemoi_core_amd.distributed import peer, primitives as P

    group = dist.group.WORLD
    wire = peer.install(group, arena_mb=64, timeout_s=20)
    dev = torch.device("cuda", torch.cuda.current_device())
    D, nl = 512, 1281
    nbrs = sorted({(rank - 1) % world, (rank + 1) % world} - {rank})
    counts = [args.rows if p in nbrs else 0 for p in range(world)]
    send_index = torch.randint(0, nl, (sum(counts),), device=dev, dtype=torch.int32)
    x = torch.randn(nl, D, device=dev).to(torch.bfloat16)

    def forward():
        with P.forward_scope(group):
            for _ in range(args.layers):
                buf = P.recv_buffer(nl, counts, counts, D, x.dtype, dev, group)
                P.halo_exchange_into(buf, nl, send_index, counts, counts, group, ops.gather_rows)
        return buf

    with torch.inference_mode():
        for _ in range(3):
            forward()
        wire.check()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            forward()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            forward()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        wire.check()
    us = e0.elapsed_time(e1) * 1e3 / args.replays
    t = torch.tensor([us], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        per = float(t.item()) / (args.layers + 1)
        print(f"[peer_latency] world {world}, {args.rows} rows x 1 KiB to each of {len(nbrs)} neighbour(s), {args.layers} exchanges + barrier per graph: "
              f"{float(t.item()):.1f} us per graph = {per:.2f} us per exchange (max over ranks, {args.replays} replays)")
    dist.barrier()
    peer.uninstall()
    dist.destroy_process_group()


def loopback(args):
    """ONE process: the exchange kernel storing into its own arena, flagging itself and waiting for that flag - the kernel's
    own cost (launch, row copy, system-scope release, flag round trip through uncached memory) without another process or
    device in the picture.  Between two MI355X devices the stores and the flag additionally cross one xGMI link."""
    import ctypes as C

    sys.path.insert(0, REPO)
    from anemoi_core_amd import _lib

    lib = _lib.load()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    D, nl, rows = 512, 1281, 2 * args.rows
    pay, flg = C.c_void_p(), C.c_void_p()
    _lib.check(lib.anemoi_peer_alloc(C.byref(pay), rows * D * 2, 0), "alloc")
    _lib.check(lib.anemoi_peer_alloc(C.byref(flg), 4096, 2), "alloc")
    x = torch.randn(nl, D, device=dev).to(torch.bfloat16)
    idx = torch.randint(0, nl, (rows,), device=dev, dtype=torch.int32)
    tab = torch.tensor([[pay.value], [flg.value + 64], [0], [rows], [1], [1]], dtype=torch.int64, device=dev)

    def go():
        _lib.check(lib.anemoi_peer_exchange_rows(x.data_ptr(), D * 2, idx.data_ptr(), tab.data_ptr(), 1, D * 2, rows, flg.value + 64,
                                                 flg.value, 5 * 100_000_000, torch.cuda.current_stream().cuda_stream), "exchange")

    for _ in range(3):
        go()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(args.layers):
            go()
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.replays / args.layers
    from anemoi_core_amd.distributed.peer import _view

    got = _view(pay.value, rows * D * 2, dev).view(torch.bfloat16).view(rows, D)
    assert torch.equal(got, x[idx.long()]), "loopback rows differ"
    print(f"[peer_latency] loopback, one process: {rows} rows x 1 KiB copied + released + flagged + flag awaited: {us:.2f} us per exchange "
          f"({args.layers} per graph, {args.replays} replays)")
    lib.anemoi_peer_free(pay)
    lib.anemoi_peer_free(flg)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--loopback", action="store_true")
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--rows", type=int, default=188)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--replays", type=int, default=200)
    a = ap.parse_args()
    if a.loopback:
        loopback(a)
        sys.exit(0)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(worker, args=(a.world, os.path.join(tmp, "init"), a), nprocs=a.world, join=True)
