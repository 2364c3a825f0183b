#!/usr/bin/env python
"""Developer aid: the per-rank time floor of the model-parallel forward (row e), measurable on a ONE-GPU box.

P ranks are started on the one GPU (gloo group + the debug host transport) so that every static cache and exchange plan of
the sharded forward is built exactly as on P GPUs.  Then only RANK 0 is timed, with the wire replaced by a no-op (the halo /
needed-rows receive buffers keep the rows of the last real exchange): that is one rank's share of the work with a
zero-latency, zero-bandwidth-cost exchange - an upper bound of what P GPUs can deliver:

  * "one hipGraph"       the whole per-rank forward as ONE graph (what device-initiated peer writes would allow),
  * "segmented chain"    the product's SegmentedGraph replay (one graph per stretch between two collectives) with no-op
                         collectives: the host-side price of re-issuing collectives between graphs is in, the wire is not.
  * --wire ipc           additionally: the product's device-initiated wire (distributed/peer.py) with rank 0's exchange kernels
                         doing everything but WAIT - rows are stored into the (idle) peers' buffers, released and flagged; the
                         peers' flags are not awaited.  One graph; what the exchange kernels themselves add to the floor.

    python tools/rank_floor.py --world 8 --hidden-res 5        (prints one JSON line)
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker(rank, world, init_file, args, out_file):
    import bench
    from anemoi_core_amd.distributed import host_transport
    from anemoi_core_amd.distributed import primitives as P
    from anemoi_core_amd.utils.segments import SegmentedGraph, collective

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    host_transport.install()
    group = dist.group.WORLD
    dev = torch.device("cuda", 0)
    wire = None
    if args.wire == "ipc":
        from anemoi_core_amd.distributed import peer

        wire = peer.install(group)
    ns = argparse.Namespace(data_grid=args.data_grid, hidden_res=args.hidden_res, kind="gt", channels=512, layers=args.layers, heads=16, vars=84)
    g, model, x = bench.build(ns, dev)
    model = model.to(dev).to(torch.bfloat16)
    inp = {"data": x.to(dev).to(torch.bfloat16)}
    step = lambda: model(inp, model_comm_group=group)["data"]  # noqa: E731
    with torch.inference_mode():
        for _ in range(2):
            step()  # real exchanges: plans, caches, receive buffers
        torch.cuda.synchronize()
        dist.barrier()
        res = None
        t_ipc = None
        scratch = None
        if wire is not None:
            wire.check()
            if rank == 0:
                # rank 0 goes on ALONE: its exchange kernels keep doing everything but wait - the rows are stored (into a
                # scratch area of its own, the peers are about to leave), released and flagged; nobody's flag is awaited
                scratch = torch.empty(max(ch.send_rows * ch.row_bytes for ch in wire._channels) + 256, dtype=torch.uint8, device=dev)
                for ch in wire._channels:
                    ch.table[0].fill_(scratch.data_ptr())
                    ch.table[1].fill_(wire.flag_ptr + 8)  # a spare word of this rank's own flag block
                    ch.table[5].zero_()
                torch.cuda.synchronize()
        dist.barrier()
        if rank != 0:
            # the idle peers' contexts on the shared GPU disturb rank 0's wall clock: they leave first
            os._exit(0)
        time.sleep(8.0)
        if wire is not None:
            if rank == 0:
                for _ in range(3):
                    step()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    step()
                torch.cuda.current_stream().wait_stream(s)
                gi = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gi):
                    step()
                for _ in range(3):
                    gi.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    gi.replay()
                torch.cuda.synchronize()
                t_ipc = (time.perf_counter() - t0) / args.steps * 1e3
                del gi
            P._all_to_all_single, P._all_gather_into_tensor, P._push_rows, P.recv_buffer, P.forward_scope = wire._saved
        if rank == 0:
            # the wire becomes a no-op; "collective" still marks the segment boundaries of the product's replay scheme
            noop = lambda *a, **k: collective(lambda: None)  # noqa: E731
            P._all_to_all_single, P._all_gather_into_tensor, P._all_reduce_sum = noop, noop, noop

            def timeit(run, n=args.steps):
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    run()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3

            sg = SegmentedGraph()
            sg.capture(step)
            t_seg = timeit(sg.replay)
            # one graph: collectives are plain no-ops (no segment boundary)
            nop2 = lambda *a, **k: None  # noqa: E731
            P._all_to_all_single, P._all_gather_into_tensor, P._all_reduce_sum = nop2, nop2, nop2
            step()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step()
            t_one = timeit(gr.replay)
            fam = bench.profile_forward(step, torch.bfloat16, 20.0)
            comp = bench.component_times(model, step, g.num_data, g.num_hidden, 512, args.layers, 20.0)
            plan = model.processor._halo_cache.get("plan")
            res = {"world": world, "hidden_res": args.hidden_res, "data_grid": args.data_grid, "layers": args.layers,
                   "local_rows": int(plan.info.num_local_nodes), "halo_rows": int(sum(plan.recv_counts)),
                   "ms_rank_one_graph": round(t_one, 4), "ms_rank_segmented_noop_wire": round(t_seg, 4), "graphs": sg.num_graphs,
                   "ms_rank_one_graph_ipc_push_no_wait": None if t_ipc is None else round(t_ipc, 4),
                   "collectives": sg.num_collectives, "components": {k: v for k, v in comp.items() if k.endswith("_ms")},
                   "families": {k: {"calls": v["calls"], "us": round(v["us"], 1)} for k, v in fam.items()}}
            json.dump(res, open(out_file, "w"))
    os._exit(0)  # the peers are gone: no collective shutdown


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--hidden-res", type=int, default=5)
    ap.add_argument("--data-grid", default="o96")
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--wire", default="", choices=["", "ipc"])
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "res.json")
        mp.spawn(worker, args=(a.world, os.path.join(tmp, "init"), a, out), nprocs=a.world, join=True)
        print(json.dumps(json.load(open(out))))
