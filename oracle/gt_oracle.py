"""ORACLE — CPU restatement of the reference's encoder-processor-decoder hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product package ``anemoi_core_amd``
never does and fails loudly when its HIP library is missing.

What it restates (plain torch, fp32, CPU; every function cites the reference lines it follows,
paths relative to /root/reference/models/src/anemoi/models):

* GraphTransformerConv + PyG segment softmax / add-aggregate   layers/conv.py:84-147
* GraphConv                                                     layers/conv.py:29-81
* MLP / LayerNorm / GELU stacks                                 layers/mlp.py:97-179
* GraphTransformer{Processor,Mapper}Block                       layers/block.py:482-1273
* GraphConv{Processor,Mapper}Block                              layers/block.py:293-479
* GraphTransformerProcessor / GNNProcessor                      layers/processor.py:319-626
* GraphTransformer{Forward,Backward}Mapper, GNN mappers         layers/mapper.py:142-1087
* AnemoiModelEncProcDec.forward glue                            models/encoder_processor_decoder.py:98-330
* node attributes / static graph provider                       layers/graph.py:20-118, layers/graph_provider.py:145-291

The arithmetic at the PyG boundary lives in an un-vendored third-party dependency
(torch-geometric >= 2.3, unpinned: models/pyproject.toml:44).  Its published semantics are
restated here: ``softmax`` = exp(x - segmax) / (segsum + 1e-16), ``scatter(sum)`` = index_add.

PARITY PINNING: the reference holds no golden numeric vectors for this path.
The oracle is pinned against fixtures generated in the build container by importing the
reference itself (tests/golden/make_golden.py, fixtures tests/golden/*.pt) — see
tests/test_oracle_golden.py.  Absolute parity with a real torch-geometric install is therefore
"pinned to the imported reference + restated PyG semantics", not to upstream golden data.

All functions are pure: parameters come in as a flat ``dict[str, Tensor]`` using the
reference's state_dict key names, selected with a ``prefix``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

Params = dict


# ------------------------------------------------------------------------------------------ primitives
def linear(p: Params, prefix: str, x: Tensor) -> Tensor:
    return F.linear(x, p[prefix + ".weight"], p.get(prefix + ".bias"))


def layer_norm(p: Params, prefix: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    """torch.nn.LayerNorm over the last dim (eps 1e-5, affine; bias optional for Query/KeyNorm,
    layers/utils.py:107-121)."""
    w = p[prefix + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, p.get(prefix + ".bias"), eps)


def gelu(x: Tensor) -> Tensor:
    return F.gelu(x)  # exact erf form = torch.nn.GELU default (layers/utils.py:111)


_GATING = {"glu": torch.sigmoid, "swiglu": F.silu, "geglu": F.gelu, "reglu": F.relu}


def mlp(p: Params, prefix: str, x: Tensor, cond: Optional[Tensor] = None) -> Tensor:
    """layers/mlp.py:97-179.  mlp_implementation="mlp": Linear, GELU, (Linear, GELU)*, Linear [, LayerNorm]; the Sequential
    indices of the Linear layers are 0, 2, 4, ... .  Gated variants (mlp.py:25-59, keys ``mlp.<i>.gate_proj/value_proj``):
    gating(gate_proj(x)) * value_proj(x) per hidden layer, then a plain Linear; the variant name is not recoverable from the
    parameters, so gated fixtures carry it as p["__mlp_implementation__"]."""
    pre = prefix + ".mlp."
    gated = sorted({int(k[len(pre):].split(".")[0]) for k in p if k.startswith(pre) and ".gate_proj.weight" in k})
    if gated:
        act = _GATING[p["__mlp_implementation__"]]
        for i in gated:
            x = act(linear(p, f"{prefix}.mlp.{i}.gate_proj", x)) * linear(p, f"{prefix}.mlp.{i}.value_proj", x)
        x = linear(p, f"{prefix}.mlp.{gated[-1] + 1}", x)
    else:
        idx = sorted(int(k[len(pre):].split(".")[0]) for k in p if k.startswith(pre) and k.endswith(".weight"))
        for n, i in enumerate(idx):
            x = linear(p, f"{prefix}.mlp.{i}", x)
            if n < len(idx) - 1:
                x = gelu(x)
    if prefix + ".layer_norm.weight" in p or prefix + ".layer_norm.scale.weight" in p:
        x = any_layer_norm(p, prefix + ".layer_norm", x, cond)
    return x


def any_layer_norm(p: Params, prefix: str, x: Tensor, cond: Optional[Tensor] = None, eps: float = 1e-5) -> Tensor:
    """LayerNorm, or ConditionalLayerNorm when the parameters are ``scale.*`` / ``bias.*`` Linears of the conditioning
    (layers/normalization.py:34-94): LN(x) without affine, times (scale(cond) + 1), plus bias(cond)."""
    if prefix + ".scale.weight" in p:
        out = F.layer_norm(x, (x.shape[-1],), None, None, eps)
        return out * (linear(p, prefix + ".scale", cond) + 1.0) + linear(p, prefix + ".bias", cond)
    return layer_norm(p, prefix, x, eps)


def segment_softmax(alpha: Tensor, index: Tensor, num_segments: int) -> Tensor:
    """torch_geometric.utils.softmax as called at layers/conv.py:144 (dim 0)."""
    shape = (num_segments,) + tuple(alpha.shape[1:])
    idx = index.view(-1, *([1] * (alpha.dim() - 1))).expand_as(alpha)
    seg_max = alpha.new_zeros(shape).scatter_reduce_(0, idx, alpha, reduce="amax", include_self=False)
    ex = (alpha - seg_max.index_select(0, index)).exp()
    seg_sum = alpha.new_zeros(shape).index_add_(0, index, ex) + 1e-16
    return ex / seg_sum.index_select(0, index)


def gt_conv(query: Tensor, key: Tensor, value: Tensor, edges: Tensor, edge_index: Tensor, size: tuple,
            alpha_scale: Optional[Tensor] = None) -> Tensor:
    """GraphTransformerConv.forward/message + add-aggregate (layers/conv.py:103-147).

    query [N_dst,H,C]; key,value [N_src,H,C]; edges [M,H,C]; edge_index [2,M] (src,dst), any order.
    Zero-in-degree destinations come out as 0 (index_add into zeros).
    ``alpha_scale`` [M,H]: the training-mode dropout of conv.py:145 with an EXPLICIT mask - F.dropout multiplies the softmax weights
    by 0 (dropped) or 1 / (1 - p) (kept); the values of that factor per (edge, head), in the order of ``edge_index``."""
    n_dst = size[1]
    C = query.shape[-1]
    src, dst = edge_index[0].long(), edge_index[1].long()
    q_i = query.index_select(0, dst)
    k_j = key.index_select(0, src) + edges
    v_j = value.index_select(0, src) + edges
    alpha = (q_i * k_j).sum(dim=-1) / C**0.5  # [M,H]
    alpha = segment_softmax(alpha, dst, n_dst)
    if alpha_scale is not None:
        alpha = alpha * alpha_scale
    msg = v_j * alpha.unsqueeze(-1)
    out = query.new_zeros((n_dst,) + tuple(query.shape[1:]))
    return out.index_add_(0, dst, msg)


def gt_conv_lse(query: Tensor, key: Tensor, edges: Tensor, edge_index: Tensor, size: tuple) -> Tensor:
    """The ``m = max + log(sum)`` side output of the fused op (triton/gt.py:171-179); 0 for empty dst
    (triton/gt.py:112-119)."""
    n_dst, C = size[1], query.shape[-1]
    src, dst = edge_index[0].long(), edge_index[1].long()
    s = (query.index_select(0, dst) * (key.index_select(0, src) + edges)).sum(-1) / C**0.5
    idx = dst.view(-1, 1).expand_as(s)
    mx = torch.full((n_dst, s.shape[1]), -float("inf")).scatter_reduce_(0, idx, s, reduce="amax", include_self=True)
    sm = torch.zeros(n_dst, s.shape[1]).index_add_(0, dst, (s - mx.index_select(0, dst)).exp())
    lse = mx + sm.log()
    deg = torch.bincount(dst, minlength=n_dst)
    return torch.where(deg.view(-1, 1) > 0, lse, torch.zeros_like(lse))


def graph_conv(p: Params, prefix: str, x_src: Tensor, x_dst: Tensor, edge_attr: Tensor, edge_index: Tensor):
    """GraphConv (layers/conv.py:66-81): e' = edge_mlp(cat[x_i, x_j, e]) + e; out[d] = sum e'."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    x_i, x_j = x_dst.index_select(0, dst), x_src.index_select(0, src)
    edges_new = mlp(p, prefix + ".edge_mlp", torch.cat([x_i, x_j, edge_attr], dim=1)) + edge_attr
    out = edges_new.new_zeros((x_dst.shape[0], edges_new.shape[1])).index_add_(0, dst, edges_new)
    return out, edges_new


# ------------------------------------------------------------------------------------------ GT blocks
def _heads(t: Tensor, H: int) -> Tensor:
    return t.view(t.shape[0], H, t.shape[1] // H)  # "nodes (heads vars) -> nodes heads vars" (block.py:644-653)


def gt_attention_part(p: Params, prefix: str, x_src_n: Tensor, x_dst_n: Tensor, edge_attr: Tensor, edge_index: Tensor,
                      num_heads: int) -> Tensor:
    """get_qkve + _forward_edges_sharded_attention (layers/block.py:623-687, 761-836); inputs are
    already layer-normed.  edge_pre_mlp is Identity by default (block.py:585-586)."""
    q = linear(p, prefix + ".lin_query", x_dst_n)
    k = linear(p, prefix + ".lin_key", x_src_n)
    v = linear(p, prefix + ".lin_value", x_src_n)
    if prefix + ".edge_pre_mlp.0.weight" in p:  # build_feedforward_layer -> Sequential(Linear, GELU) (mlp.py:55-95)
        edge_attr = gelu(linear(p, prefix + ".edge_pre_mlp.0", edge_attr))
    e = linear(p, prefix + ".lin_edge", edge_attr)
    q, k, v, e = (_heads(t, num_heads) for t in (q, k, v, e))
    if prefix + ".q_norm.weight" in p:  # qk_norm (block.py:655-660): LayerNorm over C, no bias
        q = layer_norm(p, prefix + ".q_norm", q)
        k = layer_norm(p, prefix + ".k_norm", k)
    out = gt_conv(q, k, v, e, edge_index, (x_src_n.shape[0], x_dst_n.shape[0]))
    return out.reshape(out.shape[0], -1)


def gt_processor_block(p: Params, prefix: str, x: Tensor, edge_attr: Tensor, edge_index: Tensor, num_heads: int,
                       cond: Optional[Tensor] = None) -> Tensor:
    """GraphTransformerProcessorBlock.forward (layers/block.py:1219-1273), single rank; ``cond`` feeds the layer norms when
    they are ConditionalLayerNorms."""
    xn = any_layer_norm(p, prefix + ".layer_norm_attention", x, cond)
    x_r = linear(p, prefix + ".lin_self", xn)
    out = gt_attention_part(p, prefix, xn, xn, edge_attr, edge_index, num_heads)
    out = linear(p, prefix + ".projection", out + x_r) + x
    return mlp(p, prefix + ".node_dst_mlp", any_layer_norm(p, prefix + ".layer_norm_mlp_dst", out, cond)) + out


def gt_mapper_block(p: Params, prefix: str, x_src: Tensor, x_dst: Tensor, edge_attr: Tensor, edge_index: Tensor, num_heads: int):
    """GraphTransformerMapperBlock.forward (layers/block.py:963-1029)."""
    xs_n = layer_norm(p, prefix + ".layer_norm_attention_src", x_src)
    xd_n = layer_norm(p, prefix + ".layer_norm_attention_dest", x_dst)
    x_r = linear(p, prefix + ".lin_self", xd_n)
    out = gt_attention_part(p, prefix, xs_n, xd_n, edge_attr, edge_index, num_heads)
    out = linear(p, prefix + ".projection", out + x_r) + x_dst
    dst_new = mlp(p, prefix + ".node_dst_mlp", layer_norm(p, prefix + ".layer_norm_mlp_dst", out)) + out
    if prefix + ".layer_norm_mlp_src.weight" in p:  # update_src_nodes (block.py:1022-1025)
        src_new = mlp(p, prefix + ".node_src_mlp", layer_norm(p, prefix + ".layer_norm_mlp_src", x_src)) + x_src
    else:
        src_new = x_src
    return src_new, dst_new


def sort_edges_by_dst(edge_attr: Tensor, edge_index: Tensor):
    """ensure_edges_are_dst_sorted (distributed/khop_edges.py:236-262): stable sort by dst."""
    perm = torch.sort(edge_index[1], stable=True)[1]
    return edge_attr[perm], edge_index[:, perm]


def gt_processor(p: Params, prefix: str, x: Tensor, edge_attr: Tensor, edge_index: Tensor, num_layers: int, num_heads: int) -> Tensor:
    """GraphTransformerProcessor.forward (layers/processor.py:552-626): loop of blocks, edge_attr unchanged."""
    for i in range(num_layers):
        x = gt_processor_block(p, f"{prefix}.proc.{i}" if prefix else f"proc.{i}", x, edge_attr, edge_index, num_heads)
    return x


def gt_forward_mapper(p: Params, prefix: str, x_src: Tensor, x_dst: Tensor, edge_attr: Tensor, edge_index: Tensor, num_heads: int) -> Tensor:
    """GraphTransformerForwardMapper (layers/mapper.py:480-597): embed src & dst, block, return dst.
    The reference's dst-range chunk loop (mapper.py:365-381) does not change the result."""
    pre = prefix + "." if prefix else ""
    xs = linear(p, pre + "emb_nodes_src", x_src)
    xd = linear(p, pre + "emb_nodes_dst", x_dst)
    _, dst_new = gt_mapper_block(p, pre + "proc", xs, xd, edge_attr, edge_index, num_heads)
    return dst_new


def gt_backward_mapper(p: Params, prefix: str, x_src: Tensor, x_dst: Tensor, edge_attr: Tensor, edge_index: Tensor, num_heads: int) -> Tensor:
    """GraphTransformerBackwardMapper (layers/mapper.py:600-704): embed dst only, block,
    node_data_extractor = LayerNorm + Linear."""
    pre = prefix + "." if prefix else ""
    xd = linear(p, pre + "emb_nodes_dst", x_dst)
    _, dst_new = gt_mapper_block(p, pre + "proc", x_src, xd, edge_attr, edge_index, num_heads)
    return linear(p, pre + "node_data_extractor.1", layer_norm(p, pre + "node_data_extractor.0", dst_new))


# ------------------------------------------------------------------------------------------ GraphConv (GNN) path
def gconv_processor_block(p: Params, prefix: str, x: Tensor, edge_attr: Tensor, edge_index: Tensor):
    """GraphConvProcessorBlock.forward (layers/block.py:361-395)."""
    if prefix + ".emb_edges.mlp.0.weight" in p or prefix + ".emb_edges.mlp.0.gate_proj.weight" in p:
        edge_attr = mlp(p, prefix + ".emb_edges", edge_attr)
    out, edges_new = graph_conv(p, prefix + ".conv", x, x, edge_attr, edge_index)
    nodes_new = mlp(p, prefix + ".node_mlp", torch.cat([x, out], dim=1)) + x
    return nodes_new, edges_new


def gconv_mapper_block(p: Params, prefix: str, x_src: Tensor, x_dst: Tensor, edge_attr: Tensor, edge_index: Tensor, update_src_nodes: bool):
    """GraphConvMapperBlock.forward (layers/block.py:441-479)."""
    out, edges_new = graph_conv(p, prefix + ".conv", x_src, x_dst, edge_attr, edge_index)
    dst_new = mlp(p, prefix + ".node_mlp", torch.cat([x_dst, out], dim=1)) + x_dst
    src_new = mlp(p, prefix + ".node_mlp", torch.cat([x_src, x_src], dim=1)) + x_src if update_src_nodes else x_src
    return (src_new, dst_new), edges_new


def gnn_processor(p: Params, prefix: str, x: Tensor, edge_attr: Tensor, edge_index: Tensor, num_layers: int) -> Tensor:
    """GNNProcessor.forward (layers/processor.py:397-455): edges are carried layer to layer."""
    for i in range(num_layers):
        x, edge_attr = gconv_processor_block(p, f"{prefix}.proc.{i}" if prefix else f"proc.{i}", x, edge_attr, edge_index)
    return x


def gnn_forward_mapper(p: Params, prefix: str, x_src, x_dst, edge_attr, edge_index):
    """GNNForwardMapper (layers/mapper.py:707-965): emb_edges, MLP embeddings, block with update_src_nodes=True."""
    pre = prefix + "." if prefix else ""
    e = mlp(p, pre + "emb_edges", edge_attr)
    xs, xd = mlp(p, pre + "emb_nodes_src", x_src), mlp(p, pre + "emb_nodes_dst", x_dst)
    (src_new, dst_new), _ = gconv_mapper_block(p, pre + "proc", xs, xd, e, edge_index, True)
    return src_new, dst_new


def gnn_backward_mapper(p: Params, prefix: str, x_src, x_dst, edge_attr, edge_index):
    """GNNBackwardMapper (layers/mapper.py:968-1087): no node embedding, update_src_nodes=False, MLP extractor."""
    pre = prefix + "." if prefix else ""
    e = mlp(p, pre + "emb_edges", edge_attr)
    (_, dst_new), _ = gconv_mapper_block(p, pre + "proc", x_src, x_dst, e, edge_index, False)
    return mlp(p, pre + "node_data_extractor", dst_new)


# ------------------------------------------------------------------------------------------ full model
def node_attributes(p: Params, name: str) -> Tensor:
    """NamedNodesAttributes.forward (layers/graph.py:112-118), batch 1: cat[sincos(latlon), trainable]."""
    parts = [p[f"node_attributes.latlons_{name}"]]
    t = p.get(f"node_attributes.trainable_tensors.{name}.trainable")
    if t is not None:
        parts.append(t)
    return torch.cat(parts, dim=-1)


def provider_edge_attr(p: Params, prefix: str, edge_attr_sorted: Tensor) -> Tensor:
    """StaticGraphProvider._get_edges_impl (layers/graph_provider.py:233-254), batch 1."""
    t = p.get(prefix + ".trainable.trainable")
    return edge_attr_sorted if t is None else torch.cat([edge_attr_sorted, t], dim=-1)


def enc_proc_dec_forward(p: Params, cfg: dict, graph, x: Tensor) -> Tensor:
    """AnemoiModelEncProcDec.forward for one dataset "data", ensemble 1, any batch size and number of output steps
    (models/encoder_processor_decoder.py:185-330); SkipConnection(step=-1) residual repeated over the output steps
    (layers/residual.py:53-81), no boundings.  ``graph``: object with enc/proc/dec edge_index (dst-sorted
    numpy/tensor [2,M]) and edge_attr [M,3].  x: [B, T, 1, N_data, V] -> [B, T_out, 1, N_data, V_out].

    Batch > 1 as in the reference: ONE graph of B disjoint copies - node attributes and edge attributes repeated B times
    (layers/graph.py:95-118, TrainableTensor: "e f -> (repeat e) f"), edge_index copy i shifted by i * (N_src, N_dst)
    (layers/graph_provider.py:210-231)."""
    t = lambda a: a if isinstance(a, Tensor) else torch.from_numpy(a)  # noqa: E731
    B, T, E, N, V = x.shape
    assert E == 1, "ensemble members go through AnemoiEnsModelEncProcDec in the reference (out of scope)"
    kind, H, L = cfg["kind"], cfg["num_heads"], cfg["num_layers"]
    T_out = int(cfg.get("n_step_output", 1))
    x_skip = x[:, -1, ...]  # SkipConnection step=-1 -> [B,E,N,V]
    x_data_latent = torch.cat([x.permute(0, 2, 3, 1, 4).reshape(B * E * N, T * V), node_attributes(p, "data").repeat(B, 1)], dim=-1)
    x_hidden_latent = node_attributes(p, "hidden").repeat(B, 1)
    n_hidden = x_hidden_latent.shape[0] // B

    def batched(edge_index, n_src, n_dst):
        ei = t(edge_index).long()
        inc = torch.tensor([[n_src], [n_dst]], dtype=torch.long)
        return torch.cat([ei + b * inc for b in range(B)], dim=1)

    enc_ea = provider_edge_attr(p, "encoder_graph_provider.data", t(graph.enc_edge_attr)).repeat(B, 1)
    proc_ea = provider_edge_attr(p, "processor_graph_provider", t(graph.proc_edge_attr)).repeat(B, 1)
    dec_ea = provider_edge_attr(p, "decoder_graph_provider.data", t(graph.dec_edge_attr)).repeat(B, 1)
    enc_ei, proc_ei, dec_ei = batched(graph.enc_edge_index, N, n_hidden), batched(graph.proc_edge_index, n_hidden, n_hidden), batched(graph.dec_edge_index, n_hidden, N)
    if kind == "gt":
        x_latent = gt_forward_mapper(p, "encoder.data", x_data_latent, x_hidden_latent, enc_ea, enc_ei, H)
        x_data_for_dec = x_data_latent  # forward mapper returns x[0] untouched (mapper.py:597)
        x_proc = gt_processor(p, "processor", x_latent, proc_ea, proc_ei, L, H)
    else:
        x_data_for_dec, x_latent = gnn_forward_mapper(p, "encoder.data", x_data_latent, x_hidden_latent, enc_ea, enc_ei)
        x_proc = gnn_processor(p, "processor", x_latent, proc_ea, proc_ei, L)
    x_proc = x_proc + x_latent  # latent skip (:295-296)
    if kind == "gt":
        x_out = gt_backward_mapper(p, "decoder.data", x_proc, x_data_for_dec, dec_ea, dec_ei, H)
    else:
        x_out = gnn_backward_mapper(p, "decoder.data", x_proc, x_data_for_dec, dec_ea, dec_ei)
    x_out = x_out.view(B, E, N, T_out, -1).permute(0, 3, 1, 2, 4).clone()  # "(b e g) (t v) -> b t e g v"
    n_prog = x_out.shape[-1]
    x_out[..., :n_prog] += x_skip.unsqueeze(1)[..., :n_prog]  # prognostic idx = first n_prog inputs in the fixtures; broadcast over T_out
    return x_out


# ------------------------------------------------------------------------------------------ sharding math (integer)
def apply_boundings(x: Tensor, specs: list, name_to_index: dict, statistics: Optional[dict] = None,
                    name_to_index_stats: Optional[dict] = None) -> Tensor:
    """Output boundings in configuration order (layers/bounding.py:81-307; applied at
    models/encoder_processor_decoder.py:160-162).  specs: [(class name, kwargs)]."""
    x = x.clone()

    def idx(variables):  # BaseBounding._create_index: order of name_to_index
        return [i for n, i in name_to_index.items() if n in variables]

    def lht(v, lo, hi):  # layers/activations.py:16-42
        y = torch.clamp(v, lo, hi)
        y = torch.where(v < lo, lo + 0.01 * (v - lo), y)
        return torch.where(v > hi, hi + 0.01 * (v - hi), y)

    for cls, kw in specs:
        if cls in ("NormalizedReluBounding", "NormalizedLeakyReluBounding"):
            kept = [(i, v) for i, v in enumerate(kw["variables"]) if v in name_to_index]
            cols = [name_to_index[v] for _, v in kept]  # configuration order (bounding.py:151)
            mins = []
            for i, v in kept:
                si, how, m = name_to_index_stats[v], kw["normalizer"][i], kw["min_val"][i]
                st = {k: float(t[si]) for k, t in statistics.items()}
                mins.append({"mean-std": (m - st["mean"]) / st["stdev"], "min-max": (m - st["min"]) / (st["max"] - st["min"]),
                             "max": m / st["max"], "std": m / st["stdev"]}[how])
            nm = torch.tensor(mins, dtype=torch.float32)
            f = F.relu if cls == "NormalizedReluBounding" else F.leaky_relu
            x[..., cols] = f(x[..., cols] - nm) + nm
            continue
        cols = idx(kw["variables"])
        if cls == "ReluBounding":
            x[..., cols] = F.relu(x[..., cols])
        elif cls == "LeakyReluBounding":
            x[..., cols] = F.leaky_relu(x[..., cols])
        elif cls in ("HardtanhBounding", "FractionBounding"):
            x[..., cols] = F.hardtanh(x[..., cols], kw["min_val"], kw["max_val"])
        elif cls in ("LeakyHardtanhBounding", "LeakyFractionBounding"):
            x[..., cols] = lht(x[..., cols], kw["min_val"], kw["max_val"])
        else:
            raise ValueError(cls)
        if cls.endswith("FractionBounding"):
            x[..., cols] = x[..., cols] * x[..., idx([kw["total_var"]])]
    return x


def balanced_partition_sizes(total: int, parts: int) -> list:
    """distributed/balanced_partition.py:16-41."""
    base, rem = divmod(total, parts)
    return [base + 1] * rem + [base] * (parts - rem)


def edge_splits_from_dst_sorted(edge_index: Tensor, n_dst: int, dst_splits: list) -> list:
    """build_graph_partition (distributed/khop_edges.py:154-189): per dst-range degree sums."""
    deg = torch.bincount(edge_index[1].long(), minlength=n_dst)
    return [int(c.sum()) for c in torch.split(deg, dst_splits)]


def halo_info(edge_index: Tensor, dst_splits: list, edge_splits: list, rank: int) -> dict:
    """build_halo_info (distributed/halo.py:106-222) for one rank, from the GLOBAL dst-sorted edge list."""
    e0 = sum(edge_splits[:rank])
    loc = edge_index[:, e0:e0 + edge_splits[rank]].long()
    d0 = sum(dst_splits[:rank])
    d1 = d0 + dst_splits[rank]
    src, dst = loc[0], loc[1]
    is_halo = (src < d0) | (src >= d1)
    cum = torch.cumsum(torch.tensor(dst_splits), 0)
    owner = torch.searchsorted(cum, src[is_halo], right=True)
    send, recv = [], []
    for r in range(len(dst_splits)):
        m = owner == r
        recv.append(src[is_halo][m].unique(sorted=True))
        send.append(dst[is_halo][m].unique(sorted=True) - d0)
    halo_nodes = torch.cat(recv)
    n_local = d1 - d0
    relabel = {int(g): n_local + i for i, g in enumerate(halo_nodes.tolist())}
    new_src = torch.tensor([relabel[int(s)] if h else int(s) - d0 for s, h in zip(src.tolist(), is_halo.tolist())], dtype=torch.long)
    return dict(num_local_nodes=n_local, num_halo_nodes=int(halo_nodes.numel()), send_indices=send,
                recv_counts=[int(r.numel()) for r in recv], recv_global_ids=recv,
                edge_index_local=torch.stack([new_src, dst - d0]))
