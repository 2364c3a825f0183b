#!/usr/bin/env python
"""Benchmark of the hot path: AnemoiModelEncProcDec forward (BASELINE.json configurations).

    python bench.py --gpus N --steps K --warmup W [--config o96|o96-res6|n320|gnn]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full forward (encoder -> processor layers -> decoder) over one synthetic ERA5-shaped input that is already
resident in HBM.  metric = forward nodes*channels / s = N_data * num_channels / t_forward (BASELINE.json).
``--gpus N`` with N > 1 and no torchrun environment launches the N ranks itself (one process per GPU, RCCL).  At N > 1 the
hidden mesh is sharded across the ranks (halo all-to-all per processor layer, needed-rows exchange in the decoder): the
total work is fixed -> "scaling": "strong".

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel family, measured live with HIP events on the launch
stream), "kernel_families" (algorithmic flops AND bytes per family), at N = 1 "cpu_baseline" (the oracle timed on the host
cores on a bounded sample) and "kernels"; at N > 1 "rccl" (what the exchange carries).

Configurations (BASELINE.json `configs`): o96 = config 2 (the headline; default), o96-res6 = its stated res-6 variant,
n320 = config 4 (mapper stress), gnn = config 5 (GraphConv processor on O96).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 peak

CONFIGS = {
    "o96": dict(data_grid="o96", hidden_res=5, kind="gt"),
    "o96-res6": dict(data_grid="o96", hidden_res=6, kind="gt"),
    "n320": dict(data_grid="n320", hidden_res=6, kind="gt"),
    "gnn": dict(data_grid="o96", hidden_res=5, kind="gnn"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="o96", choices=sorted(CONFIGS), help="BASELINE.json configuration (default: the headline)")
    ap.add_argument("--data-grid", default=None)
    ap.add_argument("--hidden-res", type=int, default=None)
    ap.add_argument("--kind", default=None, choices=["gt", "gnn"])
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--vars", type=int, default=84, help="variables per data node (ERA5-like)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="do not capture the forward in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernels-only", action="store_true", help="developer aid: only the per-kernel table of one processor layer")
    args = ap.parse_args()
    for k, v in CONFIGS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks (one process per GPU) under torch.distributed.run on this
    node and hand their output through (rank 0 prints the JSON line).  Mirrors what the reference's strategy does when it
    hands every rank its process group (training/src/anemoi/training/distributed/strategy.py:172-230)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (the host driver supports nothing else)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


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
    x = torch.randn(N, D, device=device).to(dtype)
    w4, b4 = blk._fused.get("qkvs", [blk.lin_query, blk.lin_key, blk.lin_value, blk.lin_self])
    qkvs = ops.linear(x, w4, b4)
    hid = blk.node_dst_mlp.mlp[0].out_features
    h = torch.randn(N, hid, device=device).to(dtype)
    es = torch.tensor([], dtype=dtype).element_size()
    ln = blk.layer_norm_attention
    cases = {
        "layernorm": (lambda: ops.layer_norm(x, ln.weight, ln.bias, ln.eps), "hbm", 2 * N * D * es),
        f"linear_qkvs({D}->{4 * D})": (lambda: ops.linear(x, w4, b4), "mfma", 2.0 * N * D * 4 * D),
        "gt_attention_fused_edge": (lambda: ops.gt_attention_fused_edge(qkvs[:, :D], qkvs[:, D:2 * D], qkvs[:, 2 * D:3 * D], feat,
                                                                        blk._fused.packed_edge(blk.lin_edge), csc, H, addend=qkvs[:, 3 * D:]), "hbm",
                                    es * 5 * N * D + 4 * M * feat.shape[1] + 4 * (M + N + 1)),  # q,k,v,self read + out write + feat + idx
        f"linear_proj({D}->{D})+res": (lambda: ops.linear(x, blk.projection.weight, blk.projection.bias, residual=x), "mfma", 2.0 * N * D * D),
        f"linear_mlp1({D}->{hid})+gelu": (lambda: ops.linear(x, blk.node_dst_mlp.mlp[0].weight, blk.node_dst_mlp.mlp[0].bias, act="gelu"), "mfma", 2.0 * N * D * hid),
        f"linear_mlp2({hid}->{D})+res": (lambda: ops.linear(h, blk.node_dst_mlp.mlp[2].weight, blk.node_dst_mlp.mlp[2].bias, residual=x), "mfma", 2.0 * N * hid * D),
    }
    if D == ops.CHAIN_CHANNELS and hid % ops.CHAIN_CHANNELS == 0 and dtype != torch.float32:
        # the row-resident chain: projection + LayerNorm + MLP + the next block's LayerNorm and q|k|v|self projection in ONE launch
        mlp, lnm = blk.node_dst_mlp, blk.layer_norm_mlp_dst
        wp, w2 = ops.pack_weight_frag(blk.projection.weight), ops.pack_weight_frag(mlp.mlp[2].weight)
        if ops.gt_layer_chain2_supported(x, hid, 4 * D):
            # role-split waves (the default for blocks of >= 4 096 rows), the LayerNorms' affine parts folded into the weights
            w1g, d1 = ops.fold_layer_norm(mlp.mlp[0].weight, mlp.mlp[0].bias, lnm.weight, lnm.bias)
            wq4, bq4 = blk._fused.get("qkvs", [blk.lin_query, blk.lin_key, blk.lin_value, blk.lin_self])
            wqg, dq = ops.fold_layer_norm(wq4, bq4, ln.weight, ln.bias)
            vec = torch.cat([blk.projection.bias.float(), d1, mlp.mlp[2].bias.float(), dq]).to(dtype).contiguous()
            w1gf, wqgf = ops.pack_weight_frag(w1g), ops.pack_weight_frag(wqg)
            cases["gt_layer_chain2(proj+mlp+next qkvs, role-split)"] = (
                lambda: ops.gt_layer_chain2(x, x, wp, w1gf, w2, vec, hid, lnm.eps, wqg=wqgf, q_out_features=4 * D, lnq_eps=ln.eps),
                "mfma", 2.0 * N * (D * D + 2 * D * hid + D * 4 * D))
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


def gnn_scatter_sum_figure(model, g, dtype, device):
    """GraphConv's scatter-sum (reference layers/conv.py:81) runs INSIDE the node chain launch (csrc/gnn_chain.hip: a wave sums the in-edge
    rows of its panel rows while the panel loads), so it has no kernel of its own to bracket.  Its cost = the launch fed with the dst-sorted
    EDGE rows + segment pointer MINUS the same launch fed with a precomputed [N, 512] table, both as hipGraph-captured back-to-back
    launches at the processor's size; its bytes = the M x 512 edge rows it reads (the table it saves is never written)."""
    from anemoi_core_amd import ops
    from anemoi_core_amd.layers.conv import node_mlp_chain, DeferredAggregate
    from anemoi_core_amd.layers.graphcache import get_csc

    blk = model.processor.proc[0]
    N, D = g.num_hidden, 512
    ea, ei, _ = model.processor_graph_provider.get_edges(batch_size=1)
    csc = get_csc(ei, (N, N), True)
    M = csc.num_edges
    x = torch.randn(N, D, device=device).to(dtype)
    e = torch.randn(M, D, device=device).to(dtype)
    table = ops.segment_sum_rows(e, csc.colptr)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                fn()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 100

    us_seg = timed(lambda: node_mlp_chain(blk.node_mlp, x, DeferredAggregate(e, csc.colptr)))
    us_tab = timed(lambda: node_mlp_chain(blk.node_mlp, x, table))
    us_own = timed(lambda: ops.segment_sum_rows(e, csc.colptr))
    es = torch.tensor([], dtype=dtype).element_size()
    byts = es * M * D + 4 * (N + 1)
    diff = max(us_seg - us_tab, 1e-3)
    return {"kernel": "segmented sum of the edge rows inside gnn_node_chain_kernel", "bound": "hbm", "us_launch_with_sum": round(us_seg, 2),
            "us_launch_table_fed": round(us_tab, 2), "us": round(diff, 2), "algorithmic_bytes": byts, "achieved": round(byts / diff / 1e3, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byts / diff / 1e3 / HBM_PEAK_GBS, 4),
            "own_launch": {"kernel": "segment_sum_rows_kernel", "us": round(us_own, 2), "bytes": byts + es * N * D,
                           "achieved": round((byts + es * N * D) / us_own / 1e3, 1), "frac": round((byts + es * N * D) / us_own / 1e3 / HBM_PEAK_GBS, 4)},
            "how": "difference of two hipGraph-captured launch series (20 launches x 5 replays each): the node chain fed with the edge rows + segment "
                   "pointer against the same launch fed with a precomputed table; `own_launch` = the stand-alone segment-sum kernel it replaced (reads "
                   "the rows, writes the table)"}


def _backed_up_queue(ms: float = 12.0):
    """Back the queue up with a SLEEP kernel (no power draw, unlike a GEMM burst, which lowers the clocks for ms afterwards -
    tools/event_probe.py) so that the host is done enqueueing before the first kernel starts: every event pair then brackets
    exactly one kernel (+ ~2.5 us of marker cost), never a wait for the host."""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(1_000_000)
    e1.record()
    torch.cuda.synchronize()
    per_ms = 1_000_000 / max(e0.elapsed_time(e1), 1e-3)
    torch.cuda._sleep(int(per_ms * ms))


def profile_forward(step, dtype, host_ms: float = 12.0):
    """One extra EAGER forward with a HIP-event pair (on the launch stream) around every kernel entry point: per kernel
    family the call count, summed device time and summed ALGORITHMIC work - flops (GEMMs) and compulsory bytes (every
    family: operands + outputs once).  These are the figures the `roofline` object is built from; profiles/ holds the
    rocprofv3 --kernel-trace --stats summary of the same command for cross-checking."""
    from anemoi_core_amd import ops

    es = torch.tensor([], dtype=dtype).element_size()
    rec = []

    def wrap(name, fn, work):
        def inner(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            rec.append((work(out, a, kw), e0, e1))
            return out
        return inner

    def lin_work(res_, a_, kw):
        x, w = a_[0], a_[1]
        N, K, O = x.shape[0], w.shape[1], w.shape[0]
        mfma = x.dtype != torch.float32 and x.shape[1] % 8 == 0 and (kw.get("x2") is None or kw["x2"].shape[1] % 8 == 0) and O % 4 == 0
        extra = sum(1 for k in ("residual", "g1", "g2") if kw.get(k) is not None)
        byts = es * (N * K + O * K + N * O * (1 + extra)) + 4 * N * sum(1 for k in ("idx1", "idx2") if kw.get(k) is not None)
        return ("linear_mfma_*" if mfma else "linear_generic_kernel"), 2.0 * N * K * O, byts

    def attn_work(res_, a_, kw):
        q, feat, csc = a_[0], a_[3], a_[5]
        D = q.shape[1]
        # compulsory traffic: q, out, self-term (3 N_dst D) + k, v (2 N_src D) + edge features + indices (lin_edge fused)
        return "gt_attn_fused_edge_fwd_kernel", 0.0, es * (3 * csc.n_dst * D + 2 * csc.n_src * D) + 4 * csc.num_edges * feat.shape[1] + 4 * (csc.num_edges + csc.n_dst + 1)

    def ln_work(res_, a_, kw):
        x = a_[0]
        return "layernorm_fwd_kernel", 0.0, 2 * x.numel() * es + (x.numel() * es if kw.get("residual") is not None else 0)

    def gemm_work(res_, a_, kw):
        x, w = a_[0], a_[1]  # the LayerNorm-fold GEMMs (statistics producer / folding consumer)
        N, K, O = x.shape[0], w.shape[1], w.shape[0]
        return "linear_mfma_*", 2.0 * N * K * O, es * (N * K + O * K + N * O)

    def chain2_work(res_, a_, kw):
        # the row-resident layer chain (csrc/gt_chain2.hip): projection + MLP-1 + MLP-2 [+ the next block's q|k|v|self projection];
        # compulsory bytes: attention rows and skip in, x2 [and the projections] out, every weight once, the per-column vectors (the hidden
        # activations and the two LayerNorms never touch memory)
        attn, vec, Hd = a_[0], a_[5], a_[6]
        N, D = attn.shape
        Oq = kw.get("q_out_features", 0)
        extra = 1 if kw.get("extra") is not None else 0
        return "gt_chain2_kernel", 2.0 * N * (D * D + 2 * D * Hd + D * Oq), es * (N * D * (3 + extra) + N * Oq + D * D + 2 * D * Hd + D * Oq + vec.numel())

    def cluster_work(res_, a_, kw):
        # the cluster chain (csrc/gt_cluster_chain.hip): the layer chain's work and compulsory bytes for block tails of few rows; what its four
        # members compute redundantly (the projection) or move between them (the partial sums) is NOT algorithmic work
        name, flops, byts = chain2_work(res_, a_, kw)
        return "gt_cluster_chain_kernel", flops, byts

    def rowchain_work(res_, a_, kw):
        # a mapper side's embedding -> LayerNorm -> projection launch (csrc/gt_rowchain.hip): rows in, the projection [and the embedded rows] out
        x, vec, Oq = a_[0], a_[3], a_[4]
        N, K, D = x.shape[0], x.shape[1], 512
        want = 1 if kw.get("want_x_out", True) else 0
        return "gt_rowchain_kernel", 2.0 * N * (K * D + D * Oq), es * (N * K + N * D * want + N * Oq + K * D + D * Oq + vec.numel())

    def edge_chain_work(res_, a_, kw):
        # GraphConv's edge MLP (three [M x 512] -> 512 GEMMs in gather-add form) + LayerNorm + residual in one launch: e in, e' out,
        # the two gathered node-level rows per edge, the indices, every weight once
        e = a_[0]
        M, D = e.shape
        return "gnn_edge_chain_kernel", 2.0 * M * 3 * D * D, es * (4 * M * D + 3 * D * D) + 8 * M

    def mlp_chain_work(res_, a_, kw):
        # an embedding MLP (csrc/gnn_chain.hip, MLP instantiation of the edge chain kernel: counted in ITS family, as the traces name it)
        x = a_[0]
        N, K = x.shape
        D = res_.shape[1]
        has_res = len(a_) > 10 and a_[10] is not None or kw.get("residual") is not None
        return "gnn_edge_chain_kernel", 2.0 * N * (K * D + 2 * D * D), es * (N * K + N * D * (2 if has_res else 1) + K * D + 2 * D * D)

    def node_chain_work(res_, a_, kw):
        x = a_[0]
        N, D = x.shape
        T = kw.get("t_out_features", 0) if kw.get("wt") is not None else 0
        rows_in = a_[1].shape[0]  # the aggregated table [N, D] or, with the scatter-sum inside the launch (seg_ptr), the edge rows [M, D]
        return "gnn_node_chain_kernel", 2.0 * N * (4 * D * D + D * T), es * (2 * N * D + rows_in * D + N * T + 4 * D * D + D * T)

    def segrows_work(res_, a_, kw):
        x, ptr = a_[0], a_[1]
        return "segment_sum_rows_kernel", 0.0, es * (x.numel() + (ptr.shape[0] - 1) * x.shape[1]) + 4 * ptr.shape[0]

    def segsum_work(res_, a_, kw):
        z, csc = a_[0], a_[5]  # read z, e_old; write e_new, agg (2(3 M D + N D))
        M, D = z.shape
        return "edge_ln_res_segsum_kernel", 0.0, es * (3 * M * D + csc.n_dst * D) + 4 * (csc.n_dst + 1)

    def rows_work(name):
        def f(res_, a_, kw):
            o = res_ if isinstance(res_, torch.Tensor) else res_[0]
            return name, 0.0, 2 * o.numel() * es
        return f

    table = {"linear": ("linear", lin_work), "gt_attention_fused_edge": ("attn", attn_work), "layer_norm": ("ln", ln_work),
             "linear_with_row_stats": ("linear_stats", gemm_work), "linear_ln_folded": ("linear_lnfold", gemm_work),
             "gt_layer_chain2": ("chain2", chain2_work), "gt_cluster_chain": ("cluster", cluster_work), "gt_row_chain": ("rowchain", rowchain_work), "gnn_edge_chain": ("edge_chain", edge_chain_work),
             "gnn_node_chain": ("node_chain", node_chain_work), "gnn_mlp_chain": ("mlp_chain", mlp_chain_work), "segment_sum_rows": ("segrows", segrows_work),
             "edge_ln_residual_segment_sum": ("segsum", segsum_work), "gather_rows": ("gather", rows_work("gather_rows_kernel")),
             "gather_add_rows": ("gather_add", rows_work("gather_add_rows_kernel"))}
    saved = {n: getattr(ops, n) for n in table}
    for n, (tag, work) in table.items():
        setattr(ops, n, wrap(tag, saved[n], work))
    empty = []
    try:
        _backed_up_queue(host_ms)
        step()
        for _ in range(32):  # calibration: brackets with NOTHING between the two markers, in the same backed-up queue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            empty.append((e0, e1))
        torch.cuda.synchronize()
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
    empty_us = sorted(a.elapsed_time(b) * 1e3 for a, b in empty)[len(empty) // 2]
    # What a bracket ADDS to the kernel inside it, measured rather than assumed: the same small kernel (a 2 MB device copy)
    # 48 times in its own bracket each, against 48 back-to-back launches inside ONE bracket (per launch), same backed-up queue.
    # An empty bracket (`empty_us`) over-states it: the two markers of a real bracket overlap the kernel's launch and retire.
    a_cal = torch.empty(1 << 20, dtype=torch.bfloat16, device="cuda")
    b_cal = torch.empty_like(a_cal)
    for _ in range(4):
        b_cal.copy_(a_cal)
    _backed_up_queue(host_ms)
    singles = []
    for _ in range(48):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b_cal.copy_(a_cal)
        e1.record()
        singles.append((e0, e1))
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(48):
        b_cal.copy_(a_cal)
    t1.record()
    torch.cuda.synchronize()
    per_launch = t0.elapsed_time(t1) * 1e3 / 48
    marker = max(0.0, sorted(a.elapsed_time(b) * 1e3 for a, b in singles)[len(singles) // 2] - per_launch)
    fam = {}
    for (name, flops, byts), e0, e1 in rec:
        d = fam.setdefault(name, {"calls": 0, "us": 0.0, "flops": 0.0, "bytes": 0.0, "marker_us": round(marker, 2), "empty_bracket_us": round(empty_us, 2)})
        d["calls"] += 1
        d["us"] += e0.elapsed_time(e1) * 1e3  # RAW bracket (kernel + the two marker gaps): what `achieved` is computed from
        d["flops"] += flops
        d["bytes"] += byts
    return fam


def component_times(model, step, n_data, n_hidden, channels, layers, host_ms: float = 12.0):
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
        _backed_up_queue(host_ms)
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


def _oracle_inputs(O, p, g, x, conv=torch.from_numpy):
    B, T, E, N, V = x.shape
    x_data = torch.cat([x[0, :, 0].permute(1, 0, 2).reshape(N, T * V), O.node_attributes(p, "data")], -1)
    x_hid = O.node_attributes(p, "hidden")
    return x_data, x_hid


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
        x_data, x_hid = _oracle_inputs(O, p, g, xd)
        enc_ea = O.provider_edge_attr(p, "encoder_graph_provider.data", tf(g.enc_edge_attr))
        proc_ea = O.provider_edge_attr(p, "processor_graph_provider", tf(g.proc_edge_attr))
        dec_ea = O.provider_edge_attr(p, "decoder_graph_provider.data", tf(g.dec_edge_attr))
        if cfg["kind"] == "gt":
            lat = O.gt_forward_mapper(p, "encoder.data", x_data, x_hid, enc_ea, enc_ei, H)
            h = O.gt_processor(p, "processor", lat, proc_ea, proc_ei, L, H) + lat
            return O.gt_backward_mapper(p, "decoder.data", h, x_data, dec_ea, dec_ei, H)
        xs, lat = O.gnn_forward_mapper(p, "encoder.data", x_data, x_hid, enc_ea, enc_ei)
        h = O.gnn_processor(p, "processor", lat, proc_ea, proc_ei, L) + lat
        return O.gnn_backward_mapper(p, "decoder.data", h, xs, dec_ea, dec_ei)

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


def physical_cores() -> int:
    """Distinct (physical id, core id) pairs of /proc/cpuinfo (SMT siblings counted once); os.cpu_count() if unreadable."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(model_fp32_params, cfg, g, x, budget_s: float = 150.0):
    """The oracle (CPU restatement of the reference, parity-pinned) on the host cores, fp32, same graph / inputs / weights:
    FULL forwards (encoder + all processor layers + decoder), 3 warm-up + 10 timed, median.  The thread
    count is the fastest of a sweep over one processor layer (eager torch ops on [M, H, C] temporaries do not scale to hundreds
    of threads).  Configurations whose forward would not fit `budget_s` get fewer repetitions (never fewer than 1 + 3); the
    sample says what was run."""
    from oracle import gt_oracle as O

    ncpu, ncore = os.cpu_count() or 1, physical_cores()
    p = model_fp32_params
    H, L, gt = cfg["num_heads"], cfg["num_layers"], cfg["kind"] == "gt"
    with torch.no_grad():
        x_data, x_hid = _oracle_inputs(O, p, g, x)
        proc_ea = O.provider_edge_attr(p, "processor_graph_provider", torch.from_numpy(g.proc_edge_attr))
        proc_ei = torch.from_numpy(g.proc_edge_index)
        h0 = x_hid.new_zeros(x_hid.shape[0], cfg["num_channels"]).normal_()

        def layer(i, h, e):
            if gt:
                return O.gt_processor_block(p, f"processor.proc.{i}", h, e, proc_ei, H), e
            return O.gconv_processor_block(p, f"processor.proc.{i}", h, e, proc_ei)

        e1 = layer(0, h0, proc_ea)[1] if not gt else proc_ea  # GNN layers >= 1 take the embedded edges
        best = None
        for th in sorted({c for c in (8, 16, 32, 64, ncore, ncpu) if c <= ncpu}):
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            layer(1 if L > 1 else 0, h0, e1 if L > 1 else proc_ea)
            dt = time.perf_counter() - t0
            if best is None or dt < best[1]:
                best = (th, dt)
            if dt > 2 * best[1]:
                break
        cores = best[0]
        torch.set_num_threads(cores)
        del h0, e1

        def forward():
            t0 = time.perf_counter()
            O.enc_proc_dec_forward(p, cfg, g, x)
            return time.perf_counter() - t0

        t_start = time.perf_counter()
        first = forward()
        warm = 3 if first * 13 <= budget_s else 1
        for _ in range(warm - 1):
            forward()
        left = budget_s - (time.perf_counter() - t_start)
        reps = max(3, min(10, int(left / max(first, 1e-3))))
        runs = sorted(forward() for _ in range(reps))
        # beside the fastest thread count: ALL physical cores (BASELINE.md section 3 words the CPU line that way) - one warm-up, two timed
        all_cores = None
        if ncore != cores and first * 3 <= 120:
            torch.set_num_threads(ncore)
            forward()
            ac = sorted(forward() for _ in range(2))
            all_cores = {"cores": ncore, "seconds_forward": round(ac[0], 3), "timed_forwards": 2}
            torch.set_num_threads(cores)
    t_full = statistics.median(runs)
    B, T, E, N, V = x.shape
    if all_cores is not None:
        all_cores["value"] = N * cfg["num_channels"] / all_cores["seconds_forward"]
    return {"value": N * cfg["num_channels"] / t_full, "unit": "nodes*channels/s", "cores": cores, "physical_cores": ncore, "logical_cpus": ncpu,
            "all_physical_cores": all_cores, "kind": "port",
            "sample": f"oracle fp32, FULL forward (encoder + {L} processor layers + decoder) on {cores} threads of {ncore} physical cores / {ncpu} "
                      f"logical CPUs (fastest thread count of a sweep over one processor layer): {warm} warm-up + {reps} timed forwards, "
                      f"median {t_full:.2f} s (min {runs[0]:.2f}, max {runs[-1]:.2f})",
            "seconds_forward": round(t_full, 3), "warmup_forwards": warm, "timed_forwards": reps}


def rccl_block(model, group, world, graph):
    """What the model-parallel exchange carries per forward (from the halo / needed-rows plans of this rank = rank 0)."""
    out = {"backend": torch.distributed.get_backend(group), "world_size": torch.distributed.get_world_size(group)}
    try:
        plan = getattr(model.processor, "_halo_cache", {}).get("plan")
        if plan is not None:
            es = next(model.parameters()).element_size()
            D = model.num_channels
            # the row that crosses the wire: the LayerNorm'd node row (D values, the reference's payload) - or, where a rank's block tails run on
            # the cluster chain (a share below the chain's row gate, inference), the owner's k|v row (2 D values) for every layer but the first
            import anemoi_core_amd.layers.block as B

            kv_rows = (B._CLUSTER_CHAIN and B._CLUSTER_HALO and B._LAYER_CHAIN and 0 < int(plan.info.num_local_nodes) < B._LAYER_CHAIN_MIN_ROWS
                       and D == 512 and es == 2)
            width = 2 * D if kv_rows else D
            out.update({"halo_rows_recv": int(sum(plan.recv_counts)), "halo_rows_send": int(sum(plan.send_counts)),
                        "halo_recv_rows_per_peer": [int(c) for c in plan.recv_counts],
                        "halo_row_payload": "k|v rows of the owners (2 x channels)" if kv_rows else "LayerNorm'd node rows (channels)",
                        "halo_bytes_recv_per_layer": int(sum(plan.recv_counts)) * width * es,
                        "halo_bytes_per_peer_per_layer_max": int(max(plan.recv_counts)) * width * es if plan.recv_counts else 0,
                        "local_rows": int(plan.info.num_local_nodes), "layers": len(model.processor.proc)})
    except Exception as e:  # noqa: BLE001
        out["plan_error"] = f"{type(e).__name__}: {e}"
    if graph is not None and hasattr(graph, "num_collectives"):
        out["collectives_per_forward"] = graph.num_collectives
    return out


def rocprof_cross_check(tag: str, family: str, d: dict, bound: str):
    """The dominant family's duration per forward in the newest COMMITTED rocprofv3 kernel-trace summary of the timed replays
    (profiles/rNN_kernel_trace_summary_<config>.txt, tools/rocprof_summary.py --timed), as achieved / frac with this run's work:
    the figure the live HIP-event measurement has to agree with."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r[0-9][0-9]_kernel_trace_summary_{tag}.txt")))
    if not files:
        return None
    prefix = family.rstrip("*")
    total_us, launches, steps = 0.0, 0, None
    for line in open(files[-1]):
        parts = line.split()
        if line.startswith("#") and "launches" in line:
            continue
        if len(parts) >= 6 and parts[0].startswith(prefix):
            try:
                launches += int(parts[-6])
                total_us += float(parts[-5])
            except ValueError:
                pass
    if not launches or launches % d["calls"]:
        return {"file": os.path.basename(files[-1]), "note": f"{launches} launches of the family in the trace, {d['calls']} per forward here: not comparable"}
    steps = launches // d["calls"]
    us = total_us / steps
    work, scale, peak = (d["flops"], 1e6, MFMA_BF16_PEAK_TFLOPS) if bound == "mfma" else (d["bytes"], 1e3, HBM_PEAK_GBS)
    return {"file": os.path.basename(files[-1]), "replays": steps, "family_us_per_forward": round(us, 1), "avg_launch_us": round(us / d["calls"], 2),
            "achieved": round(work / us / scale, 1), "frac": round(work / us / scale / peak, 4)}


def latest_traffic(tag: str):
    """HBM bytes per launch from the newest committed rocprofv3 --pmc passes for this configuration (FETCH_SIZE x2 + WRITE_SIZE,
    see DESIGN.md / profiles/README.md): profiles/rNN_pmc_traffic[_<config>].json."""
    suffix = "" if tag == "o96" else f"_{tag}"
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r[0-9][0-9]_pmc_traffic{suffix}.json")))
    if not files:
        return {}, None
    return json.load(open(files[-1])), os.path.basename(files[-1])


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; an N-GPU request must not "
                         "report a number measured on another rank count")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (ROCm) device; there is no CPU fallback for the product path")
    if world > 1:  # the ranks of a node share its cores: the set-up's CPU-side index work (partitions, halo plans) must not run world x cores threads
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    # Wire of the N > 1 run (ANEMOI_BENCH_TRANSPORT):
    #   unset  = the product: RCCL process group (backend "nccl") for set-up and as the fallback, the device-initiated hipIpc
    #            exchange (anemoi_core_amd/distributed/peer.py) as the data path - one hipGraph per rank; verified against
    #            rank 0's own unsharded forward before anything is timed, else all ranks fall back to the RCCL chain;
    #   rccl   = the RCCL chain only (host-issued all-to-alls between hipGraph segments);
    #   ipc / host = developer hooks (never set by the driver): all ranks on ONE GPU over a gloo group, with the hipIpc exchange
    #            resp. the host-staged debug transport - the N > 1 code path on a 1-GPU box, not a scaling number.
    transport = os.environ.get("ANEMOI_BENCH_TRANSPORT", "")
    if transport not in ("", "rccl", "ipc", "host"):
        raise SystemExit(f"bench.py: unknown ANEMOI_BENCH_TRANSPORT={transport!r}")
    host_transport = transport in ("host", "ipc")  # all ranks share device 0, control plane on gloo
    if host_transport:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group, wire, wire_note, wire_check = None, None, None, None
    if world > 1:
        import torch.distributed as dist

        if host_transport:
            dist.init_process_group("gloo")
            from anemoi_core_amd.distributed import host_transport as ht

            ht.install()
        else:
            dist.init_process_group("nccl", device_id=device)
        group = dist.group.WORLD
        if transport in ("", "ipc"):
            from anemoi_core_amd.distributed import peer

            try:
                wire = peer.install(group, timeout_s=float(os.environ.get("ANEMOI_PEER_TIMEOUT_S", "30")))
            except Exception as e:  # noqa: BLE001  (PeerWire set-up fails on all ranks together)
                if transport == "ipc":
                    raise
                wire_note = f"hipIpc wire unavailable ({type(e).__name__}: {str(e)[:200]}); RCCL chain"
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

    graph, graph_checked = None, None
    with torch.inference_mode():
        for _ in range(max(2, args.warmup // 2)):  # builds the static caches, sizes the allocator
            out = step()
        sync_all()
        if wire is not None:
            # the device-initiated wire has to EARN its place before anything is timed: THREE different inputs and a repeat of the
            # first go through the same receive buffers, every rank's sharded output must match rank 0's unsharded forward of that
            # input (bf16: 2e-2 of the output scale, the bound of the sharded parity tests) and no wait may have timed out - a
            # rank that reads halo rows of the PREVIOUS forward (a release / acquire hole between two devices) fails here
            ok, why = 1, ""
            checks = [x_dev, torch.roll(x_dev, 1, dims=3).contiguous(), (0.5 * x_dev.flip(3)).contiguous()]
            holder = [None]
            if rank == 0:
                try:  # on a COPY: the model's static caches (rank-local graphs, collectively built plans) stay sharded
                    import copy

                    ref_model = copy.deepcopy(model)
                    holder = [[ref_model({"data": xi})["data"].float().cpu() for xi in checks]]
                    del ref_model
                except Exception as e:  # noqa: BLE001
                    ok, why = 0, f"unsharded reference forward failed: {type(e).__name__}: {str(e)[:200]}"
            # every collective of this check is issued by EVERY rank whatever happened to it above
            torch.distributed.broadcast_object_list(holder, src=0)
            refs = holder[0]
            worst = 0.0
            for j in (0, 1, 2, 0):
                try:
                    oj = model({"data": checks[j]}, model_comm_group=group)["data"]
                    wire.check()
                    if refs is not None:
                        err = float((oj.float().cpu() - refs[j]).abs().max())
                        scale = max(1.0, float(refs[j].abs().max()))
                        worst = max(worst, err / scale)
                        if ok and not err <= (2e-5 if dtype == torch.float32 else 2e-2) * scale:
                            ok, why = 0, f"input {j}: sharded output differs from the unsharded forward by {err:.3e} (scale {scale:.2f})"
                except Exception as e:  # noqa: BLE001  (a time-out; the other ranks keep issuing their exchanges, which no longer wait)
                    if ok:
                        ok, why = 0, f"input {j}: {type(e).__name__}: {str(e)[:200]}"
            if refs is None:
                ok = 0
            holder = refs = None
            wire_check = {"inputs": 3, "forwards": 4, "max_err_over_scale": round(worst, 6)}
            flag = torch.tensor([ok], device="cpu" if host_transport else device, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if transport == "ipc":
                    raise SystemExit(f"bench.py: rank {rank}: hipIpc wire failed its check: {why or 'another rank failed'}")
                from anemoi_core_amd.distributed import peer

                whys: list = [None] * world
                torch.distributed.all_gather_object(whys, why)
                peer.uninstall()
                wire, wire_note = None, "hipIpc wire failed its check (" + "; ".join(f"rank {r}: {w}" for r, w in enumerate(whys) if w)[:600] + "); RCCL chain"
                for _ in range(2):
                    out = step()
                sync_all()
            else:
                wire.stats(reset=True)  # the timed region's exchange diagnostics start from zero
            torch.cuda.empty_cache()
        # N = 1: the whole forward is ONE hipGraph.  N > 1: RCCL collectives cannot be captured on this stack (the capture
        # aborts through the ProcessGroupNCCL watchdog or hangs; tools/nccl_capture_probe.py), so the forward becomes a
        # chain of hipGraphs with the collectives re-issued eagerly in between (anemoi_core_amd/utils/segments.py).
        if not args.no_graph and world == 1:
            expect = step().clone()
            try:  # capture the whole forward (kernels are enqueued on torch's current stream through the C ABI)
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    step()
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = step()
                graph.replay()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
            if graph is not None:
                # the kernels are deterministic: what is timed (the replay) must reproduce the eager forward bit for bit
                graph_checked = bool(torch.equal(out, expect))
                if not graph_checked:
                    raise SystemExit("bench.py: the hipGraph replay of the forward differs from the eager forward; refusing to time it")
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
                graph_checked = bool(ok)
            except Exception as e:  # noqa: BLE001
                import traceback

                ok, why = 0, f"{type(e).__name__}: {str(e).splitlines()[0]} @ " + " <- ".join(
                    f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(traceback.extract_tb(e.__traceback__)[-6:]))
            flag = torch.tensor([ok], device="cpu" if host_transport else device, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)  # every rank takes the same path
            if int(flag.item()) == 0:
                if why:
                    print(f"[bench] rank {rank}: segmented hipGraph capture unusable ({why}); all ranks run eagerly", file=sys.stderr)
                graph, graph_checked = None, None
                torch.cuda.synchronize()
        run = graph.replay if graph is not None else step
        if world > 1 and wire is not None:
            wire.stats(reset=True)  # (the capture / check forwards above ran exchanges too: count the warm-up and timed forwards only)
        for _ in range(args.warmup):
            run()
        sync_all()
        sentinel = os.environ.get("ANEMOI_BENCH_SENTINEL") == "1"  # profiling aid: marks the timed region in a kernel trace
        if sentinel:
            torch.cuda._sleep(100)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        sync_all()
        elapsed = time.perf_counter() - t0
        if sentinel:
            torch.cuda._sleep(100)
            torch.cuda.synchronize()
    rank_diag, rccl_leg = None, None
    if world > 1:
        tt = torch.tensor([elapsed], device="cpu" if host_transport else device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        # per-rank diagnostics of the timed region (every rank contributes; rank 0 prints): halo rows, what the exchange kernels
        # spent waiting for the peers' flags, a time-out's peer id
        mine = {"rank": rank}
        try:
            plan = getattr(model.processor, "_halo_cache", {}).get("plan")
            if plan is not None:
                mine.update({"local_rows": int(plan.info.num_local_nodes), "n_halo": int(sum(plan.recv_counts)), "n_send": int(sum(plan.send_counts))})
            if wire is not None:
                st = wire.stats(reset=True)
                n_fwd = args.steps + args.warmup
                mine.update({"exchanges_per_forward": round(st["exchanges"] / max(n_fwd, 1), 2), "exchange_wait_us_per_forward": round(st["wait_us_total"] / max(n_fwd, 1), 1),
                             "exchange_wait_us_max": st["wait_us_max"], "timeout_peer": st["timeout_peer"]})
        except Exception as e:  # noqa: BLE001
            mine["error"] = f"{type(e).__name__}: {str(e)[:200]}"
        rank_diag = [None] * world
        torch.distributed.all_gather_object(rank_diag, mine)
        if wire is not None:
            # ONE driver run yields the comparison: the same forward over the RCCL chain (19 hipGraph segments + host-issued
            # collectives), timed like the headline region.  A failure here is reported, never fatal: `value` is the wire's.
            ok_all = 1
            try:
                wire.check()
            except Exception:  # noqa: BLE001
                ok_all = 0
            flag = torch.tensor([ok_all], device="cpu" if host_transport else device, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)  # the verdict is agreed BEFORE the collective uninstall
            from anemoi_core_amd.distributed import peer

            ipc_segments = getattr(graph, "num_graphs", 1) if graph is not None else 0
            n_channels = len(wire._channels)
            peer.uninstall()
            wire_was, wire = "ok" if int(flag.item()) else "a wait timed out during the timed region", None
            rccl_leg = {"ms_per_step_ipc": elapsed / args.steps * 1e3, "graph_segments_ipc": ipc_segments, "ipc_status": wire_was, "peer_channels": n_channels}
            if transport != "ipc" or os.environ.get("ANEMOI_BENCH_RCCL_LEG", "1") == "1":
                try:
                    with torch.inference_mode():
                        for _ in range(2):
                            step()
                        sync_all()
                        g2 = None
                        if not args.no_graph:
                            from anemoi_core_amd.utils.segments import SegmentedGraph

                            ok2 = 1
                            try:
                                g2 = SegmentedGraph()
                                g2.capture(step)
                                g2.replay()
                                torch.cuda.synchronize()
                            except Exception:  # noqa: BLE001
                                ok2 = 0
                            f2 = torch.tensor([ok2], device="cpu" if host_transport else device, dtype=torch.int32)
                            torch.distributed.all_reduce(f2, op=torch.distributed.ReduceOp.MIN)
                            if int(f2.item()) == 0:
                                g2 = None
                        run2 = g2.replay if g2 is not None else step
                        for _ in range(args.warmup):
                            run2()
                        sync_all()
                        t1 = time.perf_counter()
                        for _ in range(args.steps):
                            run2()
                        sync_all()
                        e2 = torch.tensor([time.perf_counter() - t1], device="cpu" if host_transport else device, dtype=torch.float64)
                        torch.distributed.all_reduce(e2, op=torch.distributed.ReduceOp.MAX)
                    rccl_leg.update({"ms_per_step_rccl": float(e2.item()) / args.steps * 1e3, "graph_segments_rccl": getattr(g2, "num_graphs", 0) if g2 is not None else 0,
                                     "collectives_per_forward_rccl": getattr(g2, "num_collectives", None) if g2 is not None else None})
                except Exception as e:  # noqa: BLE001
                    rccl_leg["rccl_leg_error"] = f"{type(e).__name__}: {str(e)[:300]}"
    ms = elapsed / args.steps * 1e3
    value = g.num_data * args.channels / (ms * 1e-3)

    if rank == 0:
        kind_name = "GraphTransformer" if args.kind == "gt" else "GNN (GraphConv)"
        res = {
            "metric": f"forward nodes*channels/sec on {args.data_grid.upper()} {kind_name}",
            "value": value, "unit": "nodes*channels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (seeded N(0,1) inputs, random-init weights, own O96/N320/icosphere topology generator)",
            "config": {"name": args.config,
                       "workload": f"AnemoiModelEncProcDec forward, {args.data_grid.upper()} data grid ({g.num_data} nodes, {args.vars} vars x 2 steps) -> "
                                   f"icosphere res {args.hidden_res} hidden mesh ({g.num_hidden} nodes, {g.proc_edge_index.shape[1]} edges), "
                                   f"{kind_name} processor {args.layers} layers x {args.channels} ch" + (f" x {args.heads} heads" if args.kind == "gt" else "") +
                                   f", enc {g.enc_edge_index.shape[1]} / dec {g.dec_edge_index.shape[1]} edges, batch 1",
                       "parallelism": f"hidden mesh sharded over {world} GPU(s), halo exchange per layer" if world > 1 else "single GPU",
                       "hip_graph": graph is not None, "graph_equals_eager": graph_checked,
                       "graph_segments": getattr(graph, "num_graphs", 1) if graph is not None else 0},
        }
        if world > 1:
            res["rccl"] = rccl_block(model, group, world, graph)
            res["rccl"]["wire"] = ("hipIpc device-initiated exchange (peer stores + epoch flags inside the rank's hipGraph)" if rccl_leg is not None
                                   else "RCCL all_to_all_single between hipGraph segments")
            if rccl_leg is not None:
                res["rccl"].update(rccl_leg)
            if wire_check is not None:
                res["rccl"]["wire_check"] = wire_check
            if rank_diag is not None:
                res["rccl"]["ranks"] = rank_diag
            if wire_note:
                res["rccl"]["wire_note"] = wire_note
            if host_transport:
                res["rccl"]["note"] = f"ANEMOI_BENCH_TRANSPORT={transport}: all ranks on ONE GPU (gloo control plane) - a code-path check, not a scaling number"
        if world == 1 and not args.no_kernel_timing:
            host_ms = max(12.0, 4.0 * ms)
            with torch.inference_mode():
                fam = profile_forward(step, dtype, host_ms)
                if args.kind == "gt":
                    res["kernels"] = time_kernels(model, g, args, dtype, device)
                try:
                    res["components"] = component_times(model, step, g.num_data, g.num_hidden, args.channels, args.layers, host_ms)
                except Exception as e:  # noqa: BLE001  (an informational leg must never take the benchmark line down)
                    res["components"] = {"error": f"{type(e).__name__}: {e}"}
            # committed traces / counter passes are per CONFIGURATION: a run at another width / depth / mesh has none to be compared with
            cfg0 = CONFIGS[args.config]
            ptag = args.config if (args.channels == 512 and args.layers == 16 and args.heads == 16 and args.hidden_res == cfg0.get("hidden_res")
                                   and args.data_grid == cfg0.get("data_grid")) else f"{args.config}-c{args.channels}-l{args.layers}"
            traffic, traffic_file = latest_traffic(ptag)

            def roof(name, bound):
                d = fam[name]
                # ONE definition of the kernel's duration for `achieved` / `frac`: the HIP-event bracket on the launch stream as it
                # is (kernel + the gaps to its two markers) - a LOWER bound of the kernel's rate that needs no calibration.  Beside
                # it: the same with the calibrated bracket overhead subtracted (an UPPER bound: `bracket_overhead_us` is measured on a
                # small kernel, whose launch / retire the markers overlap less than a 30 us GEMM's) and the figure from the
                # committed rocprofv3 trace of the timed replays (`rocprof_cross_check`), which lies between the two.
                us_net = max(d["us"] - d["calls"] * d["marker_us"], 1e-3)
                work, scale, peak, unit = ((d["flops"], 1e6, MFMA_BF16_PEAK_TFLOPS, "TFLOP/s") if bound == "mfma"
                                           else (d["bytes"], 1e3, HBM_PEAK_GBS, "GB/s"))
                ach, ach_net = work / d["us"] / scale, work / us_net / scale
                tr = traffic.get(name)
                return {"kernel": name, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                        "achieved_overhead_subtracted": round(ach_net, 1), "frac_overhead_subtracted": round(ach_net / peak, 4),
                        "bracket_overhead_us": d["marker_us"], "empty_bracket_us": d["empty_bracket_us"],
                        "traffic": tr, "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"]),
                        "traffic_over_algorithmic": round(tr / (d["bytes"] / d["calls"]), 3) if tr else None,
                        "calls_per_step": d["calls"], "avg_launch_us": round(d["us"] / d["calls"], 2)}

            res["kernel_families"] = {k: {"calls": v["calls"], "total_us": round(v["us"], 1), "avg_us": round(v["us"] / v["calls"], 2),
                                          "flops": v["flops"], "bytes": v["bytes"],
                                          "tflops": round(v["flops"] / v["us"] / 1e6, 1) if v["flops"] else None,
                                          "gbs_algorithmic": round(v["bytes"] / v["us"] / 1e3, 1)} for k, v in fam.items()}
            dom = max(fam, key=lambda k: fam[k]["us"])
            res["roofline"] = roof(dom, "mfma" if fam[dom]["flops"] else "hbm")
            res["roofline"]["traffic_source"] = traffic_file
            res["roofline"]["how"] = ("sum of algorithmic work of all launches of the family in one forward / sum of their HIP-event brackets on the "
                                      "launch stream (kernel + the gaps to its two markers: a lower bound of the rate); `*_overhead_subtracted` "
                                      "= the same minus `bracket_overhead_us` per launch (48 bracketed launches of one small kernel against the "
                                      "same 48 back to back, same backed-up queue: an upper bound); `rocprof_cross_check` = the family's "
                                      "duration in the committed rocprofv3 --kernel-trace summary of the timed replays (profiles/)")
            res["roofline"]["rocprof_cross_check"] = rocprof_cross_check(ptag, dom, fam[dom], "mfma" if fam[dom]["flops"] else "hbm")
            # the gather / scatter side of the path: the attention (GraphTransformer); for GraphConv the scatter-sum, which since round 4
            # runs INSIDE the node chain launch (its bytes: the edge rows in, x in / out, the projection out, the weights once)
            gs = "gt_attn_fused_edge_fwd_kernel" if args.kind == "gt" else next(
                (k for k in ("segment_sum_rows_kernel", "edge_ln_res_segsum_kernel", "gnn_node_chain_kernel") if k in fam), None)
            if gs in fam:
                res["roofline"]["gather_scatter"] = roof(gs, "hbm")
                res["roofline"]["gather_scatter"]["rocprof_cross_check"] = rocprof_cross_check(ptag, gs, fam[gs], "hbm")
                if gs == "gnn_node_chain_kernel":
                    try:  # the scatter-sum's own figure: differential timing, the whole launch's line beside it
                        whole = res["roofline"]["gather_scatter"]
                        with torch.inference_mode():
                            res["roofline"]["gather_scatter"] = gnn_scatter_sum_figure(model, g, dtype, device)
                        res["roofline"]["gather_scatter"]["whole_launch"] = whole
                    except Exception as e:  # noqa: BLE001
                        res["roofline"]["gather_scatter"]["scatter_sum_error"] = f"{type(e).__name__}: {e}"
                if gs == "gnn_node_chain_kernel" and "scatter_sum_error" in res["roofline"]["gather_scatter"]:
                    res["roofline"]["gather_scatter"]["note"] = ("GraphConv's scatter-sum inside the node MLP launch: a GEMM chain whose panel load is "
                                                                  "the segmented sum; the HBM figure prices the whole launch against its bytes")
        if world == 1 and not args.no_cpu_baseline:
            cfg = {"num_heads": args.heads, "num_layers": args.layers, "num_channels": args.channels, "kind": args.kind}
            try:
                res["gpu_eager_baseline"] = gpu_eager_baseline(params_fp32, cfg, g, x, device)
                res["speedup_vs_gpu_eager"] = round(value / res["gpu_eager_baseline"]["value"], 2)
            except Exception as e:  # noqa: BLE001  (a baseline leg must never take the benchmark line down)
                res["gpu_eager_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline(params_fp32, cfg, g, x, budget_s=float(os.environ.get("ANEMOI_BENCH_CPU_BUDGET_S", "150")))
        print(json.dumps(res), flush=True)
    if world > 1:
        if wire is not None:  # (only when the RCCL leg did not run: it uninstalls the wire itself, verdict agreed first)
            from anemoi_core_amd.distributed import peer

            peer.uninstall()
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
