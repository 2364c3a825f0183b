#!/usr/bin/env python
"""Benchmark of the hot path: AnemoiModelEncProcDec forward on the O96 GraphTransformer configuration.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full forward (encoder -> 16 processor layers -> decoder) over one synthetic ERA5-shaped input that
is already resident in HBM.  metric = forward nodes*channels / s = N_data * num_channels / t_forward (BASELINE.json,
SURVEY.md §8d).  At N > 1 the hidden mesh is sharded across the ranks (halo all-to-all per processor layer, needed-rows
exchange in the decoder): the total work is fixed -> "scaling": "strong".

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel, measured live with HIP events on the launch
stream) and, at N = 1, "cpu_baseline" (the oracle timed on the host cores on a bounded sample) and "kernels".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--data-grid", default="o96")
    ap.add_argument("--hidden-res", type=int, default=5)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--vars", type=int, default=84, help="variables per data node (ERA5-like)")
    ap.add_argument("--kind", default="gt", choices=["gt", "gnn"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="do not capture the forward in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernels-only", action="store_true", help="developer aid: only the per-kernel table of one processor layer")
    return ap.parse_args()


def build(args, device):
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec
    from anemoi_core_amd.models.configs import make_data_indices, model_config

    g = build_synthetic_graph(args.data_grid, args.hidden_res)
    torch.manual_seed(0)
    model = AnemoiModelEncProcDec(
        model_config=model_config(args.kind, args.channels, args.layers, args.heads, 8),
        data_indices=make_data_indices(args.vars, args.vars), statistics={"data": None},
        n_step_input=2, n_step_output=1, graph_data=g,
    ).eval()
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 2, 1, g.num_data, args.vars, generator=gen)
    return g, model, x


def kernel_cases(model, g, args, dtype, device):
    """The kernels of ONE processor layer at the benchmark shapes: {name: (callable, bound, algorithmic work)}."""
    from anemoi_core_amd import ops
    from anemoi_core_amd.layers.graphcache import get_csc, get_edge_features

    blk = model.processor.proc[0]
    N, D, H = g.num_hidden, args.channels, args.heads
    M = g.proc_edge_index.shape[1]
    ea, ei, _ = model.processor_graph_provider.get_edges(batch_size=1)
    csc = get_csc(ei, (N, N), True)
    feat = get_edge_features(ea, None)
    fe = ea.shape[1]
    x = torch.randn(N, D, device=device).to(dtype)
    w4, b4 = blk._fused.get("qkvs", [blk.lin_query, blk.lin_key, blk.lin_value, blk.lin_self])
    qkvs = ops.linear(x, w4, b4)
    hid = blk.node_dst_mlp.mlp[0].out_features
    h = torch.randn(N, hid, device=device).to(dtype)
    es = dtype.itemsize if hasattr(dtype, "itemsize") else torch.tensor([], dtype=dtype).element_size()
    ln = blk.layer_norm_attention
    cases = {
        "layernorm": (lambda: ops.layer_norm(x, ln.weight, ln.bias, ln.eps), "hbm", 2 * N * D * es),
        "linear_qkvs(512->2048)": (lambda: ops.linear(x, w4, b4), "mfma", 2.0 * N * D * 4 * D),
        "gt_attention_fused_edge": (lambda: ops.gt_attention_fused_edge(qkvs[:, :D], qkvs[:, D:2 * D], qkvs[:, 2 * D:3 * D], feat,
                                                                        blk._fused.packed_edge(blk.lin_edge), csc, H, addend=qkvs[:, 3 * D:]), "hbm",
                                    es * 5 * N * D + 4 * M * feat.shape[1] + 4 * (M + N + 1)),  # q,k,v,self read + out write + feat + idx
        "linear_proj(512->512)+res": (lambda: ops.linear(x, blk.projection.weight, blk.projection.bias, residual=x), "mfma", 2.0 * N * D * D),
        "linear_mlp1(512->2048)+gelu": (lambda: ops.linear(x, blk.node_dst_mlp.mlp[0].weight, blk.node_dst_mlp.mlp[0].bias, act="gelu"), "mfma", 2.0 * N * D * hid),
        "linear_mlp2(2048->512)+res": (lambda: ops.linear(h, blk.node_dst_mlp.mlp[2].weight, blk.node_dst_mlp.mlp[2].bias, residual=x), "mfma", 2.0 * N * hid * D),
    }
    return cases


def time_kernels(model, g, args, dtype, device):
    """Per-kernel device time (HIP events on the launch stream around hipGraph-captured back-to-back launches)."""
    cases = kernel_cases(model, g, args, dtype, device)
    out = {}
    for name, (fn, bound, work) in cases.items():
        for _ in range(3):
            fn()
        reps, replays = 20, 5
        # `reps` back-to-back launches captured in a hipGraph: measures device time, not the Python/ctypes launch rate
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                fn()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(replays):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * replays)
        if bound == "hbm":
            ach, peak, unit = work / us / 1e3, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = work / us / 1e6, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s"
        out[name] = {"us": round(us, 2), "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                     "work": work}
    return out


def profile_forward(step, dtype):
    """One extra EAGER forward with a HIP-event pair (on the launch stream) around every kernel entry point: per kernel
    family the call count, summed device time and summed ALGORITHMIC work (flops for the GEMMs, bytes for the rest).
    These are the figures the `roofline` object is built from; profiles/ holds the rocprofv3 --kernel-trace --stats
    summary of the same command for cross-checking."""
    from anemoi_core_amd import ops

    es = torch.tensor([], dtype=dtype).element_size()
    rec = []

    def wrap(name, fn, work):
        def inner(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            rec.append((name, work(out, *a, **kw), e0, e1))
            return out
        return inner

    def lin_work(out, x, w, bias=None, **kw):
        K = w.shape[1]
        fam = "linear_mfma_*" if (x.dtype != torch.float32 and x.shape[1] % 8 == 0 and (kw.get("x2") is None or kw["x2"].shape[1] % 8 == 0) and w.shape[0] % 4 == 0) else "linear_generic_kernel"
        return fam, 2.0 * x.shape[0] * K * w.shape[0], "flop"

    def attn_work(out, q, k, v, feat, wp, csc, H, **kw):
        D = q.shape[1]
        # compulsory traffic: q, out, self-term (3 N_dst D) + k, v (2 N_src D) + edge features + indices (lin_edge fused)
        return "gt_attn_fused_edge_fwd_kernel", es * (3 * csc.n_dst * D + 2 * csc.n_src * D) + 4 * csc.num_edges * feat.shape[1] + 4 * (csc.num_edges + csc.n_dst + 1), "byte"

    def ln_work(out, x, *a, **kw):
        return "layernorm_fwd_kernel", 2 * x.numel() * es, "byte"

    def gemm_work(out, x, w, *a, **kw):  # the LayerNorm-fold GEMMs (statistics producer / folding consumer)
        return "linear_mfma_*", 2.0 * x.shape[0] * w.shape[1] * w.shape[0], "flop"

    saved = {n: getattr(ops, n) for n in ("linear", "gt_attention_fused_edge", "layer_norm", "linear_with_row_stats", "linear_ln_folded")}
    ops.linear_with_row_stats = wrap("linear_stats", saved["linear_with_row_stats"], gemm_work)
    ops.linear_ln_folded = wrap("linear_lnfold", saved["linear_ln_folded"], gemm_work)
    ops.linear = wrap("linear", saved["linear"], lin_work)
    ops.gt_attention_fused_edge = wrap("attn", saved["gt_attention_fused_edge"], attn_work)
    ops.layer_norm = wrap("ln", saved["layer_norm"], ln_work)
    try:
        torch.cuda.synchronize()
        # Back the queue up with a SLEEP kernel (no power draw, unlike a GEMM burst, which lowers the clocks for ms
        # afterwards - tools/event_probe.py) so that the host is done enqueueing before the first kernel starts: every
        # event pair then brackets exactly one kernel (+ ~2.5 us of marker cost), never a wait for the host.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(1_000_000)
        e1.record()
        torch.cuda.synchronize()
        per_ms = 1_000_000 / max(e0.elapsed_time(e1), 1e-3)
        torch.cuda._sleep(int(per_ms * 12.0))  # ~12 ms: the eager forward with its ~330 event records enqueues in ~5 ms
        step()
        torch.cuda.synchronize()
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
    fam = {}
    for _, (name, work, unit), e0, e1 in rec:
        d = fam.setdefault(name, {"calls": 0, "us": 0.0, "work": 0.0, "unit": unit})
        d["calls"] += 1
        d["us"] += e0.elapsed_time(e1) * 1e3
        d["work"] += work
    return fam


def component_times(model, step, n_data, n_hidden, channels, layers):
    """Device time of encoder / processor / decoder in one more eager forward (events on forward hooks, queue backed up by a
    sleep kernel as in profile_forward) and the per-component N*D/t of SURVEY.md 8(d)."""
    marks, handles = {}, []

    def pre(name):
        def f(mod, args, kwargs):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks[name] = [e, None]
        return f

    def post(name):
        def f(mod, args, kwargs, out):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks[name][1] = e
        return f

    mods = {"encoder": next(iter(model.encoder.values())), "processor": model.processor, "decoder": next(iter(model.decoder.values()))}
    for n, m in mods.items():
        handles.append(m.register_forward_pre_hook(pre(n), with_kwargs=True))
        handles.append(m.register_forward_hook(post(n), with_kwargs=True))
    try:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(1_000_000)
        e1.record()
        torch.cuda.synchronize()
        torch.cuda._sleep(int(1_000_000 / max(e0.elapsed_time(e1), 1e-3) * 12.0))
        step()
        torch.cuda.synchronize()
    finally:
        for h in handles:
            h.remove()
    ms = {n: marks[n][0].elapsed_time(marks[n][1]) for n in mods}
    return {"encoder_ms": round(ms["encoder"], 3), "processor_ms": round(ms["processor"], 3), "decoder_ms": round(ms["decoder"], 3),
            "processor_nodes_channels_layers_per_s": n_hidden * channels * layers / (ms["processor"] * 1e-3),
            "encoder_nodes_channels_per_s": n_data * channels / (ms["encoder"] * 1e-3),
            "decoder_nodes_channels_per_s": n_data * channels / (ms["decoder"] * 1e-3)}


def gpu_eager_baseline(model_fp32_params, cfg, g, x, device, steps=5):
    """The same restatement (oracle = op sequence of the reference's "pyg" backend: index_select gathers, elementwise
    temporaries, segment softmax, index_add; torch.nn.functional Linear / LayerNorm / GELU through rocBLAS / MIOpen) run
    EAGERLY on the GPU in the benchmark dtype policy's closest eager equivalent (bf16 tensors, torch kernels).  This is
    the "single-GPU PyTorch-ROCm forward" denominator of the north star's >= 5x target; neither
    PyG nor the reference's Python exists on the GPU box."""
    from oracle import gt_oracle as O

    dt = torch.bfloat16
    p = {k: (v.to(device).to(dt) if v.is_floating_point() else v.to(device)) for k, v in model_fp32_params.items()}
    H, L = cfg["num_heads"], cfg["num_layers"]
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    tf = lambda a: torch.from_numpy(a).to(device).to(dt)  # noqa: E731
    xd = x.to(device).to(dt)
    enc_ei, proc_ei, dec_ei = t(g.enc_edge_index), t(g.proc_edge_index), t(g.dec_edge_index)

    def fwd():
        B, T, E, N, V = xd.shape
        x_data = torch.cat([xd[0, :, 0].permute(1, 0, 2).reshape(N, T * V), O.node_attributes(p, "data")], -1)
        x_hid = O.node_attributes(p, "hidden")
        enc_ea = O.provider_edge_attr(p, "encoder_graph_provider.data", tf(g.enc_edge_attr))
        proc_ea = O.provider_edge_attr(p, "processor_graph_provider", tf(g.proc_edge_attr))
        dec_ea = O.provider_edge_attr(p, "decoder_graph_provider.data", tf(g.dec_edge_attr))
        lat = O.gt_forward_mapper(p, "encoder.data", x_data, x_hid, enc_ea, enc_ei, H)
        h = O.gt_processor(p, "processor", lat, proc_ea, proc_ei, L, H) + lat
        return O.gt_backward_mapper(p, "decoder.data", h, x_data, dec_ea, dec_ei, H)

    with torch.inference_mode():
        for _ in range(2):
            fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fwd()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    return {"value": x.shape[3] * cfg["num_channels"] / (ms * 1e-3), "unit": "nodes*channels/s", "ms_per_step": round(ms, 3),
            "kind": "eager PyTorch-ROCm restatement of the reference's pyg op sequence (bf16, torch/rocBLAS kernels, no hipGraph)"}


def cpu_baseline(model_fp32_params, cfg, g, x, layers_sample):
    """The oracle (CPU restatement of the reference, parity-pinned) on the host cores: encoder + ``layers_sample``
    processor layers + decoder are timed once each, the full forward is t_enc + L * t_layer + t_dec."""
    from oracle import gt_oracle as O

    ncpu = os.cpu_count() or 1
    p = model_fp32_params
    H, L = cfg["num_heads"], cfg["num_layers"]
    t = torch.from_numpy
    with torch.no_grad():
        B, T, E, N, V = x.shape
        x_data = torch.cat([x[0, :, 0].permute(1, 0, 2).reshape(N, T * V), O.node_attributes(p, "data")], -1)
        x_hid = O.node_attributes(p, "hidden")
        enc_ea = O.provider_edge_attr(p, "encoder_graph_provider.data", t(g.enc_edge_attr))
        proc_ea = O.provider_edge_attr(p, "processor_graph_provider", t(g.proc_edge_attr))
        dec_ea = O.provider_edge_attr(p, "decoder_graph_provider.data", t(g.dec_edge_attr))
        # pick the thread count that serves the CPU path best (eager torch ops on [M, H, C] temporaries do not scale
        # to hundreds of threads): one processor layer is timed at each candidate, the fastest is used throughout
        best = None
        for th in sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu}):
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            O.gt_processor_block(p, "processor.proc.0", x_hid.new_zeros(x_hid.shape[0], cfg["num_channels"]).normal_(), proc_ea, t(g.proc_edge_index), H)
            dt = time.perf_counter() - t0
            if best is None or dt < best[1]:
                best = (th, dt)
            if dt > 2 * best[1]:
                break
        cores = best[0]
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        lat = O.gt_forward_mapper(p, "encoder.data", x_data, x_hid, enc_ea, t(g.enc_edge_index), H)
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        h = lat
        for i in range(layers_sample):
            h = O.gt_processor_block(p, f"processor.proc.{i}", h, proc_ea, t(g.proc_edge_index), H)
        t_layer = (time.perf_counter() - t0) / layers_sample
        t0 = time.perf_counter()
        O.gt_backward_mapper(p, "decoder.data", h, x_data, dec_ea, t(g.dec_edge_index), H)
        t_dec = time.perf_counter() - t0
    t_full = t_enc + L * t_layer + t_dec
    return {"value": N * cfg["num_channels"] / t_full, "unit": "nodes*channels/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 on {cores} of {ncpu} host threads (fastest of a thread sweep): encoder ({t_enc:.2f}s) + {layers_sample} of {L} processor layers "
                      f"({t_layer:.3f}s each) + decoder ({t_dec:.2f}s), same O96 graph/inputs; full forward = enc + {L}*layer + dec = {t_full:.2f}s",
            "seconds_forward": round(t_full, 3)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (ROCm) device; there is no CPU fallback for the product path")
    # Developer hook (never set by the driver): ANEMOI_BENCH_TRANSPORT=host runs the N > 1 code path with all ranks on ONE
    # GPU over gloo + the test-only host transport, to exercise sharding and segmented capture on a 1-GPU box.
    host_transport = os.environ.get("ANEMOI_BENCH_TRANSPORT") == "host"
    if host_transport:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist

        if host_transport:
            dist.init_process_group("gloo")
            from tests import gpu_host_transport

            gpu_host_transport.install()
        else:
            dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]

    g, model, x = build(args, device)
    params_fp32 = {k: v.detach().clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(device).to(dtype)
    x_dev = x.to(device).to(dtype)
    inp = {"data": x_dev}

    if args.kernels_only:
        with torch.inference_mode():
            for k, v in time_kernels(model, g, args, dtype, device).items():
                print("  %-32s %8.2f us  %8.1f %-8s frac %.3f" % (k, v["us"], v["achieved"], v["unit"], v["frac"]))
        return

    def step():
        return model(inp, model_comm_group=group)["data"]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    graph = None
    with torch.inference_mode():
        for _ in range(max(2, args.warmup // 2)):  # builds the static caches, sizes the allocator
            out = step()
        sync_all()
        # N = 1: the whole forward is ONE hipGraph.  N > 1: RCCL collectives cannot be captured on this stack (the capture
        # aborts through the ProcessGroupNCCL watchdog or hangs; tools/nccl_capture_probe.py), so the forward becomes a
        # chain of hipGraphs with the collectives re-issued eagerly in between (anemoi_core_amd/utils/segments.py).
        if not args.no_graph and world == 1:
            try:  # capture the whole forward (kernels are enqueued on torch's current stream through the C ABI)
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    step()
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = step()
            except Exception as e:  # noqa: BLE001
                if rank == 0:
                    print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        elif not args.no_graph:
            from anemoi_core_amd.utils.segments import SegmentedGraph

            ok, why = 1, ""
            try:
                expect = step().clone()
                graph = SegmentedGraph()
                out = graph.capture(step)
                graph.replay()
                torch.cuda.synchronize()
                if not torch.equal(out, expect):  # kernels are deterministic: a replay must reproduce the eager run bit for bit
                    ok, why = 0, "segmented replay differs from the eager forward"
            except Exception as e:  # noqa: BLE001
                import traceback

                ok, why = 0, f"{type(e).__name__}: {str(e).splitlines()[0]} @ " + " <- ".join(
                    f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(traceback.extract_tb(e.__traceback__)[-6:]))
            flag = torch.tensor([ok], device="cpu" if host_transport else device, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)  # every rank takes the same path
            if int(flag.item()) == 0:
                if why:
                    print(f"[bench] rank {rank}: segmented hipGraph capture unusable ({why}); all ranks run eagerly", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        run = graph.replay if graph is not None else step
        for _ in range(args.warmup):
            run()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        sync_all()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device="cpu" if host_transport else device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms = elapsed / args.steps * 1e3
    value = g.num_data * args.channels / (ms * 1e-3)

    if rank == 0:
        res = {
            "metric": "forward nodes*channels/sec on O96 GraphTransformer",
            "value": value, "unit": "nodes*channels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (seeded N(0,1) inputs, random-init weights, own O96/icosphere topology generator)",
            "config": {"workload": f"AnemoiModelEncProcDec forward, {args.data_grid.upper()} data grid ({g.num_data} nodes, {args.vars} vars x 2 steps) -> "
                                   f"icosphere res {args.hidden_res} hidden mesh ({g.num_hidden} nodes, {g.proc_edge_index.shape[1]} edges), "
                                   f"{'GraphTransformer' if args.kind == 'gt' else 'GNN'} processor {args.layers} layers x {args.channels} ch x {args.heads} heads, "
                                   f"enc {g.enc_edge_index.shape[1]} / dec {g.dec_edge_index.shape[1]} edges, batch 1",
                       "parallelism": f"hidden mesh sharded over {world} GPU(s), halo all-to-all per layer" if world > 1 else "single GPU",
                       "hip_graph": graph is not None,
                       "graph_segments": getattr(graph, "num_graphs", 1) if graph is not None else 0},
        }
        if world == 1 and args.kind == "gt" and not args.no_kernel_timing:
            with torch.inference_mode():
                fam = profile_forward(step, dtype)
                res["kernels"] = time_kernels(model, g, args, dtype, device)
                try:
                    res["components"] = component_times(model, step, g.num_data, g.num_hidden, args.channels, args.layers)
                except Exception as e:  # noqa: BLE001  (an informational leg must never take the benchmark line down)
                    res["components"] = {"error": f"{type(e).__name__}: {e}"}
            res["kernel_families"] = {k: {"calls": v["calls"], "total_us": round(v["us"], 1), "avg_us": round(v["us"] / v["calls"], 2),
                                          "work": v["work"], "unit": v["unit"]} for k, v in fam.items()}
            traffic = {}
            tpath = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE), see DESIGN.md
                traffic = json.load(open(tpath))
            dom = max(fam, key=lambda k: fam[k]["us"])
            d = fam[dom]
            if d["unit"] == "flop":
                ach, peak, unit, bound = d["work"] / d["us"] / 1e6, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", "mfma"
            else:
                ach, peak, unit, bound = d["work"] / d["us"] / 1e3, HBM_PEAK_GBS, "GB/s", "hbm"
            res["roofline"] = {"kernel": dom, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                               "traffic": traffic.get(dom), "calls_per_step": d["calls"], "avg_launch_us": round(d["us"] / d["calls"], 2),
                               "how": "sum of algorithmic work of all launches of the family in one forward / sum of their HIP-event durations (each bracket carries ~2.5 us of event-marker cost, so achieved is a slight under-estimate; profiles/r01_kernel_trace_summary.txt has the rocprofv3 durations)"}
            at = fam.get("gt_attn_fused_edge_fwd_kernel")
            if at:
                res["roofline"]["gather_scatter"] = {"kernel": "gt_attn_fused_edge_fwd_kernel", "bound": "hbm", "achieved": round(at["work"] / at["us"] / 1e3, 1),
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(at["work"] / at["us"] / 1e3 / HBM_PEAK_GBS, 4),
                                                     "traffic": traffic.get("gt_attn_fused_edge_fwd_kernel"), "calls_per_step": at["calls"],
                                                     "avg_launch_us": round(at["us"] / at["calls"], 2)}
        if world == 1 and args.kind == "gt" and not args.no_cpu_baseline:
            cfg = {"num_heads": args.heads, "num_layers": args.layers, "num_channels": args.channels}
            try:
                res["gpu_eager_baseline"] = gpu_eager_baseline(params_fp32, cfg, g, x, device)
                res["speedup_vs_gpu_eager"] = round(value / res["gpu_eager_baseline"]["value"], 2)
            except Exception as e:  # noqa: BLE001  (a baseline leg must never take the benchmark line down)
                res["gpu_eager_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline(params_fp32, cfg, g, x, layers_sample=2)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
