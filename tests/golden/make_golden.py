#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING the Python reference.

Runs only in the build container (needs /root/reference).  The reference has no golden
numeric vectors of its own for this path, so the fixtures are produced
here: seeded inputs + the reference's state_dict + the reference's outputs, fp32, CPU,
``graph_attention_backend="pyg"`` (the only backend that runs without a GPU).  The
un-vendored dependencies are provided by ``ref_standins.py``.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.pt)

The fixtures are DATA (inputs, parameters, expected outputs); no reference source is stored.
"""
from __future__ import annotations

import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_standins as rs  # noqa: E402

rs.install()

from anemoi.models.distributed.shapes import BipartiteGraphShardInfo, GraphShardInfo  # noqa: E402
from anemoi.models.layers.block import (  # noqa: E402
    GraphConvMapperBlock,
    GraphConvProcessorBlock,
    GraphTransformerMapperBlock,
    GraphTransformerProcessorBlock,
)
from anemoi.models.layers.conv import GraphTransformerConv  # noqa: E402
from anemoi.models.layers.mapper import (  # noqa: E402
    GNNBackwardMapper,
    GNNForwardMapper,
    GraphTransformerBackwardMapper,
    GraphTransformerForwardMapper,
)
from anemoi.models.layers.processor import GNNProcessor, GraphTransformerProcessor  # noqa: E402
from anemoi.models.layers.utils import load_layer_kernels  # noqa: E402

from anemoi_core_amd.graphs.synthetic import build_synthetic_graph  # noqa: E402


def _sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def _randomise(module, gen, scale=0.5):
    """Default init leaves LN at (1,0) and trainable tensors at 0: perturb so they matter."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "norm" in name or name.endswith("trainable"):
                p.add_(scale * torch.randn(p.shape, generator=gen))


def _rand_graph(gen, n_src, n_dst, m, empty_dst=()):
    src = torch.randint(0, n_src, (m,), generator=gen)
    dst = torch.randint(0, n_dst, (m,), generator=gen)
    for d in empty_dst:  # force zero-in-degree destination nodes
        dst = torch.where(dst == d, (dst + 1) % n_dst, dst)
        if (d + 1) % n_dst in empty_dst:
            raise ValueError
    ei = torch.stack([src, dst])
    perm = torch.sort(ei[1], stable=True)[1]
    return ei[:, perm].contiguous()


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB")


# ----------------------------------------------------------------------------------- conv level
def gen_conv():
    gen = torch.Generator().manual_seed(1234)
    cases = []
    # shapes from the reference's own kernel test (models/tests/integration/triton/test_triton_gt.py:49-57)
    shapes = [(4, 10, 2, 4, 30), (4, 10, 6, 4, 30), (4, 10, 2, 6, 30), (4, 10, 6, 6, 30), (64, 50, 16, 32, 300), (33, 17, 4, 16, 0)]
    for n_src, n_dst, H, C, m in shapes:
        empty = (3, 7) if n_dst >= 10 else ()
        ei = _rand_graph(gen, n_src, n_dst, m, empty) if m > 0 else torch.zeros(2, 0, dtype=torch.long)
        q = torch.randn(n_dst, H, C, generator=gen)
        k = torch.randn(n_src, H, C, generator=gen)
        v = torch.randn(n_src, H, C, generator=gen)
        e = torch.randn(m, H, C, generator=gen)
        conv = GraphTransformerConv(out_channels=C)
        out = conv(q, k, v, e, ei, (n_src, n_dst))
        cases.append(dict(q=q, k=k, v=v, e=e, edge_index=ei, size=(n_src, n_dst), out=out.detach()))
    # an unsorted-edge case: the conv itself is order independent
    ei = _rand_graph(gen, 20, 12, 90)
    perm = torch.randperm(90, generator=gen)
    q, k, v, e = torch.randn(12, 4, 8, generator=gen), torch.randn(20, 4, 8, generator=gen), torch.randn(20, 4, 8, generator=gen), torch.randn(90, 4, 8, generator=gen)
    out = GraphTransformerConv(out_channels=8)(q, k, v, e[perm], ei[:, perm], (20, 12))
    cases.append(dict(q=q, k=k, v=v, e=e[perm], edge_index=ei[:, perm], size=(20, 12), out=out.detach()))
    # a large-score case exercising the running-max rescale of an online softmax
    ei = _rand_graph(gen, 16, 8, 64)
    q, k, v, e = 6 * torch.randn(8, 2, 16, generator=gen), 6 * torch.randn(16, 2, 16, generator=gen), torch.randn(16, 2, 16, generator=gen), torch.randn(64, 2, 16, generator=gen)
    out = GraphTransformerConv(out_channels=16)(q, k, v, e, ei, (16, 8))
    cases.append(dict(q=q, k=k, v=v, e=e, edge_index=ei, size=(16, 8), out=out.detach()))
    save("conv.pt", cases)


# ----------------------------------------------------------------------------------- block level
def gen_blocks():
    gen = torch.Generator().manual_seed(4321)
    lk = load_layer_kernels()
    out = {}
    torch.manual_seed(11)
    for tag, qk_norm, C, hid, H in [("proc_qknorm", True, 128, 64, 8), ("proc", False, 64, 256, 4)]:
        blk = GraphTransformerProcessorBlock(
            in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=11, qk_norm=qk_norm,
            layer_kernels=lk, graph_attention_backend="pyg",
        ).eval()
        _randomise(blk, gen)
        N, M = 60, 400
        ei = _rand_graph(gen, N, N, M, (5,))
        x = torch.randn(N, C, generator=gen)
        ea = torch.randn(M, 11, generator=gen)
        y, ea_out = blk(x, ea, ei, GraphShardInfo(nodes=[N], edges=[M]), 1, N)
        assert ea_out is ea
        out[tag] = dict(cfg=dict(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=11, qk_norm=qk_norm),
                        params=_sd(blk), x=x, edge_attr=ea, edge_index=ei, out=y.detach())
    for tag, qk_norm, upd in [("map", False, False), ("map_qknorm_updsrc", True, True)]:
        C, hid, H = 64, 128, 4
        blk = GraphTransformerMapperBlock(
            in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=7, qk_norm=qk_norm,
            update_src_nodes=upd, layer_kernels=lk, graph_attention_backend="pyg",
        ).eval()
        _randomise(blk, gen)
        Ns, Nd, M = 70, 40, 300
        ei = _rand_graph(gen, Ns, Nd, M, (0,))
        xs, xd = torch.randn(Ns, C, generator=gen), torch.randn(Nd, C, generator=gen)
        ea = torch.randn(M, 7, generator=gen)
        (ys, yd), _ = blk((xs, xd), ea, ei, BipartiteGraphShardInfo(src_nodes=[Ns], dst_nodes=[Nd], edges=[M]), 1, (Ns, Nd))
        out[tag] = dict(cfg=dict(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=7, qk_norm=qk_norm, update_src_nodes=upd),
                        params=_sd(blk), x_src=xs, x_dst=xd, edge_attr=ea, edge_index=ei, out_src=ys.detach(), out_dst=yd.detach())
    # GraphConv blocks
    C = 32
    blk = GraphConvProcessorBlock(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=0, layer_kernels=lk, edge_dim=None).eval()
    _randomise(blk, gen)
    N, M = 50, 260
    ei = _rand_graph(gen, N, N, M, (9,))
    x, ea = torch.randn(N, C, generator=gen), torch.randn(M, C, generator=gen)
    y, e2 = blk(x, ea, ei, GraphShardInfo(nodes=[N], edges=[M]), None, size=(N, N))
    out["gconv_proc"] = dict(cfg=dict(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=0, edge_dim=None),
                             params=_sd(blk), x=x, edge_attr=ea, edge_index=ei, out=y.detach(), edges_out=e2.detach())
    blk = GraphConvProcessorBlock(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=1, layer_kernels=lk, edge_dim=5).eval()
    _randomise(blk, gen)
    ea5 = torch.randn(M, 5, generator=gen)
    y, e2 = blk(x, ea5, ei, GraphShardInfo(nodes=[N], edges=[M]), None, size=(N, N))
    out["gconv_proc_emb"] = dict(cfg=dict(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=1, edge_dim=5),
                                 params=_sd(blk), x=x, edge_attr=ea5, edge_index=ei, out=y.detach(), edges_out=e2.detach())
    for tag, upd in [("gconv_map", False), ("gconv_map_updsrc", True)]:
        blk = GraphConvMapperBlock(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=0, update_src_nodes=upd, layer_kernels=lk, edge_dim=None).eval()
        _randomise(blk, gen)
        Ns, Nd, M = 45, 30, 200
        ei = _rand_graph(gen, Ns, Nd, M)
        xs, xd, ea = torch.randn(Ns, C, generator=gen), torch.randn(Nd, C, generator=gen), torch.randn(M, C, generator=gen)
        (ys, yd), e2 = blk((xs, xd), ea, ei, BipartiteGraphShardInfo(src_nodes=[Ns], dst_nodes=[Nd], edges=[M]), None, size=(Ns, Nd))
        out[tag] = dict(cfg=dict(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=0, update_src_nodes=upd, edge_dim=None),
                        params=_sd(blk), x_src=xs, x_dst=xd, edge_attr=ea, edge_index=ei, out_src=ys.detach(), out_dst=yd.detach(), edges_out=e2.detach())
    save("blocks.pt", out)


def gen_blocks_train():
    """Blocks with the options whose TRAINING path VERDICT r2 item 9 asks for - edge_pre_mlp and qk_norm (block.py:585-586,
    637-687) - with the reference's forward output AND the reference's own autograd gradients of loss = sum(out * w) w.r.t. the
    input, the edge attributes and every parameter."""
    gen = torch.Generator().manual_seed(977)
    lk = load_layer_kernels()
    out = {}
    torch.manual_seed(12)
    for tag, qk_norm, pre in [("proc_edgepre_qknorm", True, True), ("proc_edgepre", False, True)]:
        C, hid, H = 64, 128, 4
        cfg = dict(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=11, qk_norm=qk_norm, edge_pre_mlp=pre)
        blk = GraphTransformerProcessorBlock(layer_kernels=lk, graph_attention_backend="pyg", **cfg).train()
        _randomise(blk, gen)
        N, M = 60, 400
        ei = _rand_graph(gen, N, N, M, (5,))
        x = torch.randn(N, C, generator=gen).requires_grad_(True)
        ea = torch.randn(M, 11, generator=gen).requires_grad_(True)
        w = torch.randn(N, C, generator=gen)
        y, _ = blk(x, ea, ei, GraphShardInfo(nodes=[N], edges=[M]), 1, N)
        (y * w).sum().backward()
        out[tag] = dict(cfg=cfg, params=_sd(blk), x=x.detach(), edge_attr=ea.detach(), edge_index=ei, w=w, out=y.detach(),
                        dx=x.grad.clone(), d_edge_attr=ea.grad.clone(),
                        grads={k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None})
        print(tag, "out", float(y.abs().mean()), "grads", len(out[tag]["grads"]))
    C, hid, H = 64, 128, 4
    cfg = dict(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=7, qk_norm=True, edge_pre_mlp=True, update_src_nodes=False)
    blk = GraphTransformerMapperBlock(layer_kernels=lk, graph_attention_backend="pyg", **cfg).train()
    _randomise(blk, gen)
    Ns, Nd, M = 70, 40, 300
    ei = _rand_graph(gen, Ns, Nd, M, (0,))
    xs = torch.randn(Ns, C, generator=gen).requires_grad_(True)
    xd = torch.randn(Nd, C, generator=gen).requires_grad_(True)
    ea = torch.randn(M, 7, generator=gen).requires_grad_(True)
    w = torch.randn(Nd, C, generator=gen)
    (ys, yd), _ = blk((xs, xd), ea, ei, BipartiteGraphShardInfo(src_nodes=[Ns], dst_nodes=[Nd], edges=[M]), 1, (Ns, Nd))
    (yd * w).sum().backward()
    out["map_edgepre_qknorm"] = dict(cfg=cfg, params=_sd(blk), x_src=xs.detach(), x_dst=xd.detach(), edge_attr=ea.detach(), edge_index=ei, w=w,
                                     out_dst=yd.detach(), dx_src=xs.grad.clone(), dx_dst=xd.grad.clone(), d_edge_attr=ea.grad.clone(),
                                     grads={k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None})
    save("blocks_train.pt", out)


# ----------------------------------------------------------------------------------- processor / mapper level
def gen_proc_mappers():
    gen = torch.Generator().manual_seed(777)
    torch.manual_seed(22)
    out = {}
    # configs follow models/tests/layers/processor/test_graphtransformer_processor.py:28-45
    cfg = dict(num_layers=2, num_channels=128, num_chunks=2, num_heads=16, mlp_hidden_ratio=4, edge_dim=13, qk_norm=True,
               cpu_offload=False, layer_kernels=None, graph_attention_backend="pyg", gradient_checkpointing=False)
    proc = GraphTransformerProcessor(**cfg).eval()
    _randomise(proc, gen)
    N, M = 100, 200
    ei = _rand_graph(gen, N, N, M)
    x, ea = torch.randn(N, 128, generator=gen), torch.randn(M, 13, generator=gen)
    y = proc(x, 1, GraphShardInfo(nodes=[N], edges=None), ea, ei)
    out["gt_processor"] = dict(cfg=cfg, params=_sd(proc), x=x, edge_attr=ea, edge_index=ei, out=y.detach())

    # unsorted edges: the processor must sort internally
    perm = torch.randperm(M, generator=gen)
    y2 = proc(x, 1, GraphShardInfo(nodes=[N], edges=None), ea[perm], ei[:, perm], edges_are_dst_sorted=False)
    out["gt_processor_unsorted"] = dict(perm=perm, out=y2.detach())

    # mapper configs follow models/tests/layers/mapper/test_graphtransformer_mapper.py:39-58 (hidden reduced to 64)
    mcfg = dict(in_channels_src=5, in_channels_dst=3, hidden_dim=64, num_chunks=2, num_heads=4, mlp_hidden_ratio=4, edge_dim=6,
                qk_norm=False, cpu_offload=False, layer_kernels=None, shard_strategy="edges", graph_attention_backend="pyg",
                gradient_checkpointing=False)
    Ns, Nd, M = 100, 200, 450
    ei = _rand_graph(gen, Ns, Nd, M, (17,))
    xs, xd, ea = torch.randn(Ns, 5, generator=gen), torch.randn(Nd, 3, generator=gen), torch.randn(M, 6, generator=gen)
    si = BipartiteGraphShardInfo(src_nodes=None, dst_nodes=[Nd], edges=None)
    fwd = GraphTransformerForwardMapper(**mcfg).eval()
    _randomise(fwd, gen)
    xs_out, xd_out = fwd((xs, xd), 1, si, ea, ei)
    assert xs_out is xs
    out["gt_forward_mapper"] = dict(cfg=mcfg, params=_sd(fwd), x_src=xs, x_dst=xd, edge_attr=ea, edge_index=ei, out_dst=xd_out.detach())

    bcfg = dict(mcfg, in_channels_src=64, in_channels_dst=3, out_channels_dst=7, qk_norm=True)
    bwd = GraphTransformerBackwardMapper(**bcfg).eval()
    _randomise(bwd, gen)
    xs64 = torch.randn(Ns, 64, generator=gen)
    yd = bwd((xs64, xd), 1, si, ea, ei)
    out["gt_backward_mapper"] = dict(cfg=bcfg, params=_sd(bwd), x_src=xs64, x_dst=xd, edge_attr=ea, edge_index=ei, out_dst=yd.detach())

    # GNN processor + mappers (models/tests/layers/processor/test_graphconv_processor.py)
    gcfg = dict(num_layers=3, num_channels=32, num_chunks=1, mlp_extra_layers=0, edge_dim=9, cpu_offload=False, layer_kernels=None,
                gradient_checkpointing=False)
    gproc = GNNProcessor(**gcfg).eval()
    _randomise(gproc, gen)
    N, M = 80, 300
    ei = _rand_graph(gen, N, N, M)
    x, ea = torch.randn(N, 32, generator=gen), torch.randn(M, 9, generator=gen)
    y = gproc(x, 1, GraphShardInfo(nodes=[N], edges=None), ea, ei)
    out["gnn_processor"] = dict(cfg=gcfg, params=_sd(gproc), x=x, edge_attr=ea, edge_index=ei, out=y.detach())

    gm = dict(in_channels_src=5, in_channels_dst=3, hidden_dim=32, mlp_extra_layers=0, edge_dim=6, num_chunks=1, cpu_offload=False,
              layer_kernels=None, gradient_checkpointing=False)
    Ns, Nd, M = 60, 90, 250
    ei = _rand_graph(gen, Ns, Nd, M)
    xs, xd, ea = torch.randn(Ns, 5, generator=gen), torch.randn(Nd, 3, generator=gen), torch.randn(M, 6, generator=gen)
    si = BipartiteGraphShardInfo(src_nodes=None, dst_nodes=[Nd], edges=None)
    gf = GNNForwardMapper(**gm).eval()
    _randomise(gf, gen)
    ys, yd = gf((xs, xd), 1, si, ea, ei)
    out["gnn_forward_mapper"] = dict(cfg=gm, params=_sd(gf), x_src=xs, x_dst=xd, edge_attr=ea, edge_index=ei, out_src=ys.detach(), out_dst=yd.detach())
    gb_cfg = dict(gm, in_channels_src=32, in_channels_dst=32, out_channels_dst=4)
    gb = GNNBackwardMapper(**gb_cfg).eval()
    _randomise(gb, gen)
    xs32, xd32 = torch.randn(Ns, 32, generator=gen), torch.randn(Nd, 32, generator=gen)
    yd = gb((xs32, xd32), 1, si, ea, ei)  # GNNBackwardMapper does not embed dst: it must already be hidden_dim wide
    out["gnn_backward_mapper"] = dict(cfg=gb_cfg, params=_sd(gb), x_src=xs32, x_dst=xd32, edge_attr=ea, edge_index=ei, out_dst=yd.detach())
    save("proc_mappers.pt", out)


# ----------------------------------------------------------------------------------- full model
class _IndexGroup(SimpleNamespace):
    def __len__(self):
        return len(self.full)


def make_data_indices(n_vars_in, n_prog):
    names = {f"v{i}": i for i in range(n_vars_in)}
    prog = list(range(n_prog))
    ns = SimpleNamespace(
        model=SimpleNamespace(
            input=_IndexGroup(prognostic=prog, full=list(range(n_vars_in)), name_to_index=names),
            output=_IndexGroup(prognostic=prog, full=prog, diagnostic=[], name_to_index={f"v{i}": i for i in prog}),
            _forcing=[],
        ),
        data=SimpleNamespace(input=SimpleNamespace(name_to_index=names)),
        name_to_index=names,
    )
    return {"data": ns}


def make_hetero(g):
    hd = rs.HeteroData()
    hd["data"].x = torch.from_numpy(g.data_latlon)
    hd["data"].num_nodes = g.num_data
    hd["hidden"].x = torch.from_numpy(g.hidden_latlon)
    hd["hidden"].num_nodes = g.num_hidden
    for key, ei, ea in (
        (("data", "to", "hidden"), g.enc_edge_index, g.enc_edge_attr),
        (("hidden", "to", "hidden"), g.proc_edge_index, g.proc_edge_attr),
        (("hidden", "to", "data"), g.dec_edge_index, g.dec_edge_attr),
    ):
        hd[key].edge_index = torch.from_numpy(ei).to(torch.int32)
        hd[key].edge_length = torch.from_numpy(ea[:, :1].copy())
        hd[key].edge_dirs = torch.from_numpy(ea[:, 1:].copy())
    return hd


def model_config(kind, num_channels, num_layers, num_heads, trainable):
    gt = kind == "gt"
    P = "anemoi.models.layers"
    common = dict(cpu_offload=False, gradient_checkpointing=False, layer_kernels=None, trainable_size=trainable,
                  sub_graph_edge_attributes=["edge_length", "edge_dirs"])
    if gt:
        common.update(num_heads=num_heads, mlp_hidden_ratio=4, qk_norm=False, shard_strategy="edges",
                      graph_attention_backend="pyg", edge_pre_mlp=False)
        enc = dict(common, _target_=f"{P}.mapper.GraphTransformerForwardMapper", num_chunks=2)
        proc = dict(common, _target_=f"{P}.processor.GraphTransformerProcessor", num_chunks=1, num_layers=num_layers)
        dec = dict(common, _target_=f"{P}.mapper.GraphTransformerBackwardMapper", num_chunks=2, initialise_data_extractor_zero=False)
    else:
        common.update(mlp_extra_layers=0)
        enc = dict(common, _target_=f"{P}.mapper.GNNForwardMapper", num_chunks=1)
        proc = dict(common, _target_=f"{P}.processor.GNNProcessor", num_chunks=1, num_layers=num_layers)
        dec = dict(common, _target_=f"{P}.mapper.GNNBackwardMapper", num_chunks=1)
    return rs.DotDict({
        "model": {
            "num_channels": num_channels,
            "trainable_parameters": {"data": trainable, "hidden": trainable, "data2hidden": trainable, "hidden2data": trainable, "hidden2hidden": trainable},
            "model": {"hidden_nodes_name": "hidden", "latent_skip": True},
            "encoder": enc, "processor": proc, "decoder": dec,
            "residual": {"_target_": "anemoi.models.layers.residual.SkipConnection", "step": -1},
            "bounding": [],
        }
    })


def gen_model():
    from anemoi.models.models import AnemoiModelEncProcDec

    g = build_synthetic_graph("o8", 3)  # tiny config: 642 hidden nodes (icosphere res 3), small data grid
    out = {}
    for kind in ("gt", "gnn"):
        torch.manual_seed(33)
        gen = torch.Generator().manual_seed(99)
        n_vars, n_prog, n_step = 4, 4, 2
        cfg = dict(kind=kind, num_channels=64, num_layers=2, num_heads=4, trainable=8, n_vars=n_vars, n_step_input=n_step,
                   data_grid="o8", hidden_resolution=3)
        model = AnemoiModelEncProcDec(
            model_config=model_config(kind, 64, 2, 4, 8),
            data_indices=make_data_indices(n_vars, n_prog),
            statistics={"data": None},
            n_step_input=n_step,
            n_step_output=1,
            graph_data=make_hetero(g),
        ).eval()
        _randomise(model, gen, scale=0.3)
        x = torch.randn(1, n_step, 1, g.num_data, n_vars, generator=gen)
        with torch.no_grad():
            y = model({"data": x})["data"]
        out[kind] = dict(cfg=cfg, params=_sd(model), x=x, out=y)
        print(kind, "model out", tuple(y.shape), float(y.abs().mean()))
    save("model_tiny.pt", out)


def gen_model_batch():
    """Batch > 1 and n_step_output > 1 through the REFERENCE's AnemoiModelEncProcDec:
    graph_provider.py:210-231 (edge_inc batch expansion), layers/residual.py:53-57 (skip repeated over the output steps),
    encoder_processor_decoder.py:131-163 ((batch ensemble grid) (time vars) rearrangements).  Small models (32 channels, one
    processor layer) so that the fixture stays small; one set of parameters per n_step_output (the decoder's width changes)."""
    from anemoi.models.models import AnemoiModelEncProcDec

    g = build_synthetic_graph("o8", 3)
    out = {}
    for kind in ("gt", "gnn"):
        for t_out in (1, 2):
            torch.manual_seed(41 + t_out)
            gen = torch.Generator().manual_seed(77 + t_out)
            n_vars, n_prog, n_step = 4, 3, 2  # one diagnostic-free forcing variable: input width 4, output width 3
            cfg = dict(kind=kind, num_channels=32, num_layers=1, num_heads=4, trainable=4, n_vars=n_vars, n_prog=n_prog, n_step_input=n_step,
                       n_step_output=t_out, data_grid="o8", hidden_resolution=3)
            model = AnemoiModelEncProcDec(
                model_config=model_config(kind, 32, 1, 4, 4),
                data_indices=make_data_indices(n_vars, n_prog),
                statistics={"data": None},
                n_step_input=n_step,
                n_step_output=t_out,
                graph_data=make_hetero(g),
            ).eval()
            _randomise(model, gen, scale=0.3)
            cases = []
            for B, E in ((2, 1), (3, 1)) if t_out == 1 else ((1, 1), (2, 1)):
                x = torch.randn(B, n_step, E, g.num_data, n_vars, generator=gen)
                with torch.no_grad():
                    y = model({"data": x})["data"]
                assert y.shape == (B, t_out, E, g.num_data, n_prog), y.shape
                cases.append(dict(x=x, out=y))
                print(kind, "t_out", t_out, "B", B, "E", E, "out", tuple(y.shape), float(y.abs().mean()))
            try:
                model({"data": torch.randn(1, n_step, 2, g.num_data, n_vars, generator=gen)})
                ens_err = None
            except Exception as e:  # noqa: BLE001
                ens_err = type(e).__name__
            out[f"{kind}_t{t_out}"] = dict(cfg=cfg, params=_sd(model), cases=cases, ensemble_error=ens_err)
    save("model_batch.pt", out)


def gen_model_grads():
    """Gradients of the REFERENCE's tiny models (same parameters / input as model_tiny.pt): loss = sum(out * w) with the seeded w the
    gradient tests use; pins the backward (scope row f1) to the reference's own autograd instead of the oracle's."""
    from anemoi.models.models import AnemoiModelEncProcDec

    base = torch.load(os.path.join(HERE, "model_tiny.pt"), weights_only=False)
    g = build_synthetic_graph("o8", 3)
    out = {}
    for kind in ("gt", "gnn"):
        c = base[kind]
        cfg = c["cfg"]
        torch.manual_seed(33)
        model = AnemoiModelEncProcDec(
            model_config=model_config(kind, cfg["num_channels"], cfg["num_layers"], cfg["num_heads"], cfg["trainable"]),
            data_indices=make_data_indices(cfg["n_vars"], cfg["n_vars"]),
            statistics={"data": None},
            n_step_input=cfg["n_step_input"],
            n_step_output=1,
            graph_data=make_hetero(g),
        ).train()
        model.load_state_dict(c["params"], strict=True)
        x = c["x"].clone().requires_grad_(True)
        w = torch.randn(c["out"].shape, generator=torch.Generator().manual_seed(2))
        y = model({"data": x})["data"]
        assert float((y.detach() - c["out"]).abs().max()) < 1e-5
        (y * w).sum().backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        out[kind] = dict(grads=grads, dx=x.grad.clone(), loss_weight_seed=2)
        print(kind, "model grads", len(grads), "parameters, |dx| mean", float(x.grad.abs().mean()))
    save("model_tiny_grads.pt", out)


# ----------------------------------------------------------------------------------- sharding (gloo, multi-process)
def _shard_worker(rank, world, init_file, g_proc, params, cfg, x, ea, result_dir):
    import torch.distributed as dist

    rs.install()
    from anemoi.models.distributed.graph import shard_tensor
    from anemoi.models.distributed.shapes import GraphShardInfo as GSI, get_shard_sizes
    from anemoi.models.layers.processor import GraphTransformerProcessor as GTP

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    group = dist.new_group(list(range(world)))
    proc = GTP(**cfg).eval()
    proc.load_state_dict(params)
    ei = torch.from_numpy(g_proc)
    sizes = get_shard_sizes(x, 0, group)
    x_loc = shard_tensor(x, 0, sizes, group)
    with torch.no_grad():
        y_loc = proc(x_loc, 1, GSI(nodes=sizes, edges=None), ea, ei, model_comm_group=group)
    halo = proc.proc[0]._cached_halo_info
    part = proc.proc[0]._cached_partition
    torch.save(
        dict(node_sizes=sizes, edge_splits=list(part.edge_splits), dst_splits=list(part.dst_splits),
             num_local_nodes=halo.num_local_nodes, num_halo_nodes=halo.num_halo_nodes,
             send_indices=[t.clone() for t in halo.send_indices], recv_counts=list(halo.recv_counts),
             edge_index_local=halo.edge_index_local.clone(), out_local=y_loc.clone()),
        os.path.join(result_dir, f"w{world}_r{rank}.pt"),
    )
    dist.barrier()
    dist.destroy_process_group()


def gen_sharding():
    import torch.multiprocessing as mp

    from anemoi_core_amd.graphs.synthetic import edge_attributes, icosphere_latlon, multiscale_edges

    gen = torch.Generator().manual_seed(555)
    torch.manual_seed(44)
    ei = multiscale_edges(2)  # 162 nodes, 1200 symmetric edges
    ll, _ = icosphere_latlon(2)
    ea3 = torch.from_numpy(edge_attributes(ll, ll, ei))
    N, M = ll.shape[0], ei.shape[1]
    ea = torch.cat([ea3, torch.randn(M, 4, generator=gen)], dim=1)
    cfg = dict(num_layers=3, num_channels=32, num_chunks=1, num_heads=4, mlp_hidden_ratio=2, edge_dim=7, qk_norm=False,
               cpu_offload=False, layer_kernels=None, shard_strategy="edges", graph_attention_backend="pyg", gradient_checkpointing=False)
    proc = GraphTransformerProcessor(**cfg).eval()
    _randomise(proc, gen)
    params = _sd(proc)
    x = torch.randn(N, 32, generator=gen)
    with torch.no_grad():
        y = proc(x, 1, GraphShardInfo(nodes=[N], edges=None), ea, torch.from_numpy(ei))
    out = dict(cfg=cfg, params=params, x=x, edge_attr=ea, edge_index=torch.from_numpy(ei), out=y, ranks={})
    with tempfile.TemporaryDirectory() as tmp:
        for world in (2, 3):
            init_file = os.path.join(tmp, f"init{world}")
            mp.spawn(_shard_worker, args=(world, init_file, ei, params, cfg, x, ea, tmp), nprocs=world, join=True)
            ranks = [torch.load(os.path.join(tmp, f"w{world}_r{r}.pt")) for r in range(world)]
            y_cat = torch.cat([r["out_local"] for r in ranks])
            print(f"world {world}: sharded vs unsharded max abs diff {float((y_cat - y).abs().max()):.3e}")
            out["ranks"][world] = ranks
    save("sharding.pt", out)


def gen_gnn_gated():
    """GraphConv blocks with a gated edge / node MLP (`mlp_implementation` = swiglu, geglu: layers/conv.py:29-81 builds its edge MLP with
    it, layers/block.py:286-345 the node MLP and the edge embedding), forward and the reference's own autograd."""
    gen = torch.Generator().manual_seed(2468)
    lk = load_layer_kernels()
    out = {}
    torch.manual_seed(13)
    C = 32
    N, M = 50, 260
    ei = _rand_graph(gen, N, N, M, (9,))
    for kind, edge_dim, extra in (("swiglu", None, 0), ("geglu", 5, 1)):
        blk = GraphConvProcessorBlock(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=extra, layer_kernels=lk, edge_dim=edge_dim,
                                      mlp_implementation=kind).eval()
        _randomise(blk, gen)
        x = torch.randn(N, C, generator=gen).requires_grad_(True)
        ea = torch.randn(M, edge_dim or C, generator=gen).requires_grad_(True)
        y, e2 = blk(x, ea, ei, GraphShardInfo(nodes=[N], edges=[M]), None, size=(N, N))
        wy, we = torch.randn(y.shape, generator=gen), torch.randn(e2.shape, generator=gen)
        ((y * wy).sum() + (e2 * we).sum()).backward()
        out[f"proc_{kind}"] = dict(cfg=dict(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=extra, edge_dim=edge_dim, mlp_implementation=kind),
                                   params=_sd(blk), x=x.detach(), edge_attr=ea.detach(), edge_index=ei, out=y.detach(), edges_out=e2.detach(),
                                   w_out=wy, w_edges=we, grad_x=x.grad.clone(), grad_edge_attr=ea.grad.clone(),
                                   grads={k: v.grad.clone() for k, v in blk.named_parameters()})
    Ns, Nd, Mm = 45, 30, 200
    eim = _rand_graph(gen, Ns, Nd, Mm)
    blk = GraphConvMapperBlock(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=0, update_src_nodes=True, layer_kernels=lk, edge_dim=None,
                               mlp_implementation="swiglu").eval()
    _randomise(blk, gen)
    xs, xd, ea = torch.randn(Ns, C, generator=gen), torch.randn(Nd, C, generator=gen), torch.randn(Mm, C, generator=gen)
    with torch.no_grad():
        (ys, yd), e2 = blk((xs, xd), ea, eim, BipartiteGraphShardInfo(src_nodes=[Ns], dst_nodes=[Nd], edges=[Mm]), None, size=(Ns, Nd))
    out["map_swiglu"] = dict(cfg=dict(in_channels=C, out_channels=C, num_chunks=1, mlp_extra_layers=0, update_src_nodes=True, edge_dim=None,
                                      mlp_implementation="swiglu"),
                             params=_sd(blk), x_src=xs, x_dst=xd, edge_attr=ea, edge_index=eim, out_src=ys, out_dst=yd, edges_out=e2)
    save("gnn_gated.pt", out)


# ----------------------------------------------------------------------------------- scope row f3 variants
def gen_variants():
    """Gated MLP variants (layers/mlp.py:25-59) and ConditionalLayerNorm (layers/normalization.py:34-94)."""
    from anemoi.models.layers.mlp import MLP
    from anemoi.models.layers.normalization import ConditionalLayerNorm

    gen = torch.Generator().manual_seed(777)
    lk = load_layer_kernels()
    out = {"mlp": {}, "block": {}, "cond": {}}
    torch.manual_seed(5)
    for kind in ("glu", "swiglu", "geglu", "reglu"):
        for extra, ln in ((0, True), (1, False)):
            m = MLP(64, 96, 48, layer_kernels=lk, n_extra_layers=extra, layer_norm=ln, mlp_implementation=kind).eval()
            _randomise(m, gen)
            x = torch.randn(70, 64, generator=gen)
            out["mlp"][f"{kind}_{extra}_{int(ln)}"] = dict(cfg=dict(in_features=64, hidden_dim=96, out_features=48, n_extra_layers=extra, layer_norm=ln,
                                                                   mlp_implementation=kind), params=_sd(m), x=x, out=m(x).detach())
    for kind in ("swiglu", "geglu"):
        C, hid, H = 64, 128, 4
        blk = GraphTransformerProcessorBlock(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=11, layer_kernels=lk,
                                             graph_attention_backend="pyg", mlp_implementation=kind).eval()
        _randomise(blk, gen)
        N, M = 60, 400
        ei = _rand_graph(gen, N, N, M, (5,))
        x, ea = torch.randn(N, C, generator=gen), torch.randn(M, 11, generator=gen)
        y, _ = blk(x, ea, ei, GraphShardInfo(nodes=[N], edges=[M]), 1, N)
        out["block"][kind] = dict(cfg=dict(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=11, mlp_implementation=kind),
                                  params=_sd(blk), x=x, edge_attr=ea, edge_index=ei, out=y.detach())
    # ConditionalLayerNorm alone and inside a processor block (layer_kernels.LayerNorm swapped, as a config would)
    cln = ConditionalLayerNorm(64, condition_shape=16, zero_init=False).eval()
    _randomise(cln, gen)
    x, cond = torch.randn(50, 64, generator=gen), torch.randn(50, 16, generator=gen)
    out["cond"]["layer"] = dict(cfg=dict(normalized_shape=64, condition_shape=16, zero_init=False), params=_sd(cln), x=x, cond=cond,
                                out=cln(x, cond).detach())
    lk_c = load_layer_kernels({"LayerNorm": {"_target_": "anemoi.models.layers.normalization.ConditionalLayerNorm", "condition_shape": 16,
                                             "zero_init": False}})
    C, hid, H = 64, 128, 4
    blk = GraphTransformerProcessorBlock(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=11, layer_kernels=lk_c,
                                         graph_attention_backend="pyg").eval()
    _randomise(blk, gen)
    N, M = 60, 400
    ei = _rand_graph(gen, N, N, M, (5,))
    x, ea, cond = torch.randn(N, C, generator=gen), torch.randn(M, 11, generator=gen), torch.randn(N, 16, generator=gen)
    y, _ = blk(x, ea, ei, GraphShardInfo(nodes=[N], edges=[M]), 1, N, cond=cond)
    out["cond"]["block"] = dict(cfg=dict(in_channels=C, hidden_dim=hid, out_channels=C, num_heads=H, edge_dim=11), params=_sd(blk), x=x, edge_attr=ea,
                                edge_index=ei, cond=cond, out=y.detach())
    # output boundings (layers/bounding.py:81-307), applied in config order on one tensor
    from anemoi.models.layers import bounding as B

    names = [f"v{i}" for i in range(10)]
    n2i = {n: i for i, n in enumerate(names)}
    n2i_stats = {n: i + 1 for i, n in enumerate(names)}  # statistics are indexed by the DATA input index: different on purpose
    stats = {k: torch.rand(12, generator=gen).double().numpy() + (1.0 if k in ("stdev", "max") else 0.0) for k in ("mean", "stdev", "min", "max")}
    stats["max"] = stats["min"] + stats["max"]
    specs = [
        ("ReluBounding", dict(variables=["v0", "v3"])),
        ("LeakyReluBounding", dict(variables=["v1"])),
        ("NormalizedReluBounding", dict(variables=["v2", "v4"], min_val=[0.1, -0.2], normalizer=["mean-std", "min-max"])),
        ("NormalizedLeakyReluBounding", dict(variables=["v5", "v9"], min_val=[0.3, 0.0], normalizer=["max", "std"])),
        ("HardtanhBounding", dict(variables=["v6"], min_val=-0.5, max_val=0.7)),
        ("LeakyHardtanhBounding", dict(variables=["v7"], min_val=0.0, max_val=1.0)),
        ("FractionBounding", dict(variables=["v8"], min_val=0.0, max_val=1.0, total_var="v0")),
        ("LeakyFractionBounding", dict(variables=["v1"], min_val=0.0, max_val=1.0, total_var="v3")),
    ]
    x = 1.5 * torch.randn(200, 10, generator=gen)
    y = x.clone()
    for cls, kw in specs:
        y = getattr(B, cls)(name_to_index=n2i, statistics=stats, name_to_index_stats=n2i_stats, **kw)(y)
    out["bounding"] = dict(specs=specs, name_to_index=n2i, name_to_index_stats=n2i_stats,
                           statistics={k: torch.as_tensor(v) for k, v in stats.items()}, x=x, out=y)
    save("variants.pt", out)


def gen_edges():
    """Model-edge fixtures (scope row f4): the reference's InputNormalizer (preprocessing/normalizer.py), the reference's
    predict_step around its tiny model (models/base.py:303-391: normalise -> forward -> de-normalise, forcing and diagnostic
    variables so that input / output / model-output index sets differ) and a graph FILE in PyG's pickle layout."""
    from omegaconf import DictConfig

    from anemoi.models.data_indices.collection import IndexCollection
    from anemoi.models.models import AnemoiModelEncProcDec
    from anemoi.models.preprocessing import Processors
    from anemoi.models.preprocessing.normalizer import InputNormalizer

    gen = torch.Generator().manual_seed(4242)
    out = {}
    # ---- normaliser alone: every method, a remap, forcing + diagnostic variables
    names = ["a", "b", "c", "d", "e", "f", "g", "h"]
    n2i = {n: i for i, n in enumerate(names)}
    data_cfg = {"normalizer": {"default": "mean-std", "remap": {"h": "a"}, "min-max": ["b"], "max": ["c"], "none": ["d"], "std": ["e"]},
                "forcing": ["d", "e"], "diagnostic": ["g"]}
    stats = {"mean": torch.randn(8, generator=gen).double().numpy() * 3.0, "stdev": torch.rand(8, generator=gen).double().numpy() * 2.0 + 0.5,
             "minimum": -torch.rand(8, generator=gen).double().numpy() * 5.0 - 1.0, "maximum": torch.rand(8, generator=gen).double().numpy() * 5.0 + 1.0}
    cfg = DictConfig({"data": data_cfg})
    di = IndexCollection(data_config=cfg.data, name_to_index=n2i)
    nm = InputNormalizer(config=cfg.data.normalizer, data_indices=di, statistics={k: v.copy() for k, v in stats.items()})
    idx = dict(data_input_full=di.data.input.full.clone(), data_output_full=di.data.output.full.clone(),
               data_input_name_to_index=dict(di.data.input.name_to_index), data_output_name_to_index=dict(di.data.output.name_to_index),
               model_input_name_to_index=dict(di.model.input.name_to_index), model_output_name_to_index=dict(di.model.output.name_to_index),
               model_input_prognostic=di.model.input.prognostic.clone(), model_output_prognostic=di.model.output.prognostic.clone(),
               model_input_full=di.model.input.full.clone(), model_output_full=di.model.output.full.clone())
    x_all = 4.0 * torch.randn(3, 5, 8, generator=gen)
    x_in = 4.0 * torch.randn(3, 5, len(di.data.input.full), generator=gen)
    x_out = 4.0 * torch.randn(3, 5, len(di.data.output.full), generator=gen)
    out["normalizer"] = dict(
        data_config=data_cfg, name_to_index=n2i, statistics={k: torch.as_tensor(v) for k, v in stats.items()}, indices=idx, buffers=_sd(nm),
        x_all=x_all, t_all=nm.transform(x_all, in_place=False), i_all=nm.inverse_transform(x_all, in_place=False),
        x_in=x_in, t_in=nm.transform(x_in, in_place=False), x_out=x_out, i_out=nm.inverse_transform(x_out, in_place=False),
        data_index=[0, 2, 7], t_idx=nm.transform(x_all[..., [0, 2, 7]].contiguous(), in_place=False, data_index=[0, 2, 7]),
        i_idx=nm.inverse_transform(x_all[..., [0, 2, 7]].contiguous(), in_place=False, data_index=[0, 2, 7]))

    # ---- predict_step of the reference's tiny GraphTransformer model with the normaliser as pre / post processor
    g = build_synthetic_graph("o8", 3)
    names = [f"v{i}" for i in range(6)]
    n2i = {n: i for i, n in enumerate(names)}
    data_cfg = {"normalizer": {"default": "mean-std", "min-max": ["v1"], "max": ["v2"], "std": ["v4"]}, "forcing": ["v4"], "diagnostic": ["v5"]}
    stats = {"mean": torch.randn(6, generator=gen).double().numpy(), "stdev": torch.rand(6, generator=gen).double().numpy() + 0.5,
             "minimum": -torch.rand(6, generator=gen).double().numpy() * 3.0 - 1.0, "maximum": torch.rand(6, generator=gen).double().numpy() * 3.0 + 1.0}
    cfg = DictConfig({"data": data_cfg})
    di = IndexCollection(data_config=cfg.data, name_to_index=n2i)
    torch.manual_seed(77)
    model = AnemoiModelEncProcDec(model_config=model_config("gt", 64, 2, 4, 8), data_indices={"data": di}, statistics={"data": stats},
                                  n_step_input=2, n_step_output=1, graph_data=make_hetero(g)).eval()
    _randomise(model, gen, scale=0.3)
    nm = InputNormalizer(config=cfg.data.normalizer, data_indices=di, statistics={k: v.copy() for k, v in stats.items()})
    pre, post = Processors([["normalizer", nm]]), Processors([["normalizer", nm]], inverse=True)
    batch = 2.0 * torch.randn(1, 3, g.num_data, len(di.data.input.full), generator=gen) + 0.5  # one spare time step: [:, 0:2] is used
    with torch.no_grad():
        y = model.predict_step({"data": batch}, {"data": pre}, {"data": post}, 2)["data"]
        y_norm = model({"data": pre(batch[:, 0:2, None, ...], in_place=False)})["data"]  # the normalised-space output, for diagnosis
    idx = dict(data_input_full=di.data.input.full.clone(), data_output_full=di.data.output.full.clone(),
               data_input_name_to_index=dict(di.data.input.name_to_index), data_output_name_to_index=dict(di.data.output.name_to_index),
               model_input_name_to_index=dict(di.model.input.name_to_index), model_output_name_to_index=dict(di.model.output.name_to_index),
               model_input_prognostic=di.model.input.prognostic.clone(), model_output_prognostic=di.model.output.prognostic.clone(),
               model_input_full=di.model.input.full.clone(), model_output_full=di.model.output.full.clone())
    out["predict_step"] = dict(cfg=dict(kind="gt", num_channels=64, num_layers=2, num_heads=4, trainable=8, n_step_input=2, data_grid="o8", hidden_resolution=3),
                               data_config=data_cfg, name_to_index=n2i, statistics={k: torch.as_tensor(v) for k, v in stats.items()}, indices=idx,
                               params=_sd(model), batch=batch, out=y, out_normalised=y_norm)
    print("predict_step out", tuple(y.shape), float(y.abs().mean()))
    save("edges.pt", out)

    # ---- a graph FILE as anemoi-graphs writes it: torch.save(HeteroData) (graphs/src/anemoi/graphs/create.py), PyG pickle layout
    g2 = build_synthetic_graph("o8", 2)
    hd = rs.PygHeteroData()
    hd["data"].x = torch.from_numpy(g2.data_latlon)
    hd["data"].node_type = "ReducedGaussianGridNodes"
    hd["data"].area_weight = torch.rand(g2.num_data, 1, generator=gen)
    hd["hidden"].x = torch.from_numpy(g2.hidden_latlon)
    hd["hidden"].node_type = "TriNodes"
    for key, ei, ea in ((("data", "to", "hidden"), g2.enc_edge_index, g2.enc_edge_attr), (("hidden", "to", "hidden"), g2.proc_edge_index, g2.proc_edge_attr),
                        (("hidden", "to", "data"), g2.dec_edge_index, g2.dec_edge_attr)):
        hd[key].edge_index = torch.from_numpy(ei)
        hd[key].edge_type = "CutOffEdges"
        hd[key].edge_length = torch.from_numpy(ea[:, :1].copy())
        hd[key].edge_dirs = torch.from_numpy(ea[:, 1:].copy())
    path = os.path.join(HERE, "graph_file.pt")
    torch.save(hd, path)
    print(f"graph_file.pt: {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["conv", "blocks", "blocks_train", "proc", "model", "batch", "grads", "sharding", "variants", "gnn_gated", "edges"]
    if "conv" in which:
        gen_conv()
    if "blocks" in which:
        gen_blocks()
    if "blocks_train" in which:
        gen_blocks_train()
    if "proc" in which:
        gen_proc_mappers()
    if "model" in which:
        gen_model()
    if "batch" in which:
        gen_model_batch()
    if "grads" in which:
        gen_model_grads()
    if "sharding" in which:
        gen_sharding()
    if "variants" in which:
        gen_variants()
    if "gnn_gated" in which:
        gen_gnn_gated()
    if "edges" in which:
        gen_edges()
