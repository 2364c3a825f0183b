"""AnemoiModelEncProcDec — mirror of the reference glue (models/src/anemoi/models/models/encoder_processor_decoder.py
:33-340 and models/base.py:38-395) for the graph (GraphTransformer / GNN) model family: same constructor arguments,
``forward(x: {name: [B,T,E,N,V]}, model_comm_group=None, grid_shard_sizes=None)`` and state_dict keys
(``encoder.<ds>.*``, ``processor.*``, ``decoder.<ds>.*``, ``node_attributes.*``, ``*_graph_provider.*``).

``model_config`` is the reference's nested config (attribute dict).  ``_target_`` strings may name either this
package's classes or the reference's ``anemoi.models.layers.*`` classes (rewritten to the MI355X implementations), so an
existing config selects the HIP path without edits.  Residual = SkipConnection (step); boundings: all eight classes of
layers/bounding.py, run as one in-place column program.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor, nn

from .. import ops
from ..distributed.primitives import scoped_forward, shard_tensor
from ..distributed.shapes import BipartiteGraphShardInfo, GraphShardInfo, comm_size, get_shard_sizes
from ..layers.graph import NamedNodesAttributes
from ..layers.graph_provider import create_graph_provider
from ..layers.mapper import GraphTransformerBaseMapper
from ..utils.tensors import version

_FUSED_INPUT = os.environ.get("ANEMOI_FUSED_INPUT", "1") == "1"  # developer switch: 0 = permute-copy + cat in torch
_PAD64 = os.environ.get("ANEMOI_PAD64", "1") == "1"  # developer switch: 0 = pad the input width to a multiple of 8 only
# the decoder block's k|v projection of the hidden rows at the end of the last processor block's chain launch (ANEMOI_TAIL_KV=0: LayerNorm launch + GEMM)
_TAIL_KV = os.environ.get("ANEMOI_TAIL_KV", "1") != "0"
from ..utils.config import DotDict, instantiate

_REF_PREFIX = "anemoi.models.layers."
_OUR_PREFIX = "anemoi_core_amd.layers."


def _retarget(cfg) -> dict:
    cfg = dict(cfg)
    t = cfg.get("_target_", "")
    if isinstance(t, str) and t.startswith(_REF_PREFIX):
        cfg["_target_"] = _OUR_PREFIX + t[len(_REF_PREFIX):]
    return cfg


def _graph_views(graph_data):
    """Accept the synthetic generator's ``SyntheticGraph`` or a HeteroData-like object (``g[name].x``,
    ``g[(src,'to',dst)].edge_index / edge_length / edge_dirs``)."""
    if hasattr(graph_data, "enc_edge_index"):  # SyntheticGraph
        g = graph_data
        nodes = {"data": torch.as_tensor(g.data_latlon), "hidden": torch.as_tensor(g.hidden_latlon)}
        mk = lambda ei, ea: {"edge_index": torch.as_tensor(ei), "edge_length": torch.as_tensor(ea[:, :1]), "edge_dirs": torch.as_tensor(ea[:, 1:])}  # noqa: E731
        edges = {("data", "to", "hidden"): mk(g.enc_edge_index, g.enc_edge_attr),
                 ("hidden", "to", "hidden"): mk(g.proc_edge_index, g.proc_edge_attr),
                 ("hidden", "to", "data"): mk(g.dec_edge_index, g.dec_edge_attr)}
        return nodes, edges

    class _Edges(dict):
        def __missing__(self, key):
            return graph_data[key]

    nodes = {name: torch.as_tensor(graph_data[name].x) for name in graph_data.node_types}
    return nodes, _Edges()


class AnemoiModelEncProcDec(nn.Module):
    def __init__(self, *, model_config, data_indices: dict, statistics: Optional[dict] = None, n_step_input: int,
                 n_step_output: int, graph_data) -> None:
        super().__init__()
        model_config = model_config if isinstance(model_config, DotDict) else DotDict(model_config)
        mc = model_config.model
        self.data_indices = data_indices
        self.statistics = statistics
        self.n_step_input = n_step_input
        self.n_step_output = n_step_output
        self.dataset_names = list(data_indices.keys())
        self._graph_name_hidden = mc.model.hidden_nodes_name
        self.num_channels = mc.num_channels
        self.latent_skip = mc.model.latent_skip
        if not isinstance(self._graph_name_hidden, str):
            raise NotImplementedError("hierarchical hidden node lists belong to another model family")
        nodes, edges = _graph_views(graph_data)
        tp = dict(mc.trainable_parameters)
        self.node_attributes = NamedNodesAttributes(
            {**{ds: tp.get("data", 0) for ds in self.dataset_names}, self._graph_name_hidden: tp.get("hidden", 0)},
            {**{ds: nodes[ds] for ds in self.dataset_names}, self._graph_name_hidden: nodes[self._graph_name_hidden]},
        )
        self._calculate_shapes_and_indices(data_indices)
        self._build_networks(mc, edges)
        res = mc.get("residual", {}) or {}
        self._skip_step = int(res.get("step", -1))
        self.boundings = self._build_boundings(mc.get("bounding", []) or [])
        self._pad_zeros = self._hidden_padded = None
        self._bound_tables: dict = {}

    # -- shapes (models/base.py:96-150) -------------------------------------------------------------------
    def _calculate_shapes_and_indices(self, data_indices: dict) -> None:
        self.num_input_channels, self.num_output_channels = {}, {}
        self._internal_input_idx, self._internal_output_idx = {}, {}
        self.input_dim, self.target_dim, self.output_dim = {}, {}, {}
        self.input_dim_latent = self.node_attributes.attr_ndims[self._graph_name_hidden]
        for ds, idx in data_indices.items():
            self._internal_input_idx[ds] = list(idx.model.input.prognostic)
            self._internal_output_idx[ds] = list(idx.model.output.prognostic)
            self.num_input_channels[ds] = len(idx.model.input)
            self.num_output_channels[ds] = len(idx.model.output)
            self.input_dim[ds] = self.n_step_input * self.num_input_channels[ds] + self.node_attributes.attr_ndims[ds]
            self.target_dim[ds] = self.input_dim[ds]
            self.output_dim[ds] = self.n_step_output * self.num_output_channels[ds]
            assert len(self._internal_input_idx[ds]) == len(self._internal_output_idx[ds])
            # device-resident index vectors (non-persistent: the state_dict stays identical to the reference's) so that
            # the residual scatter involves no host->device copy and the forward is hipGraph-capturable
            self.register_buffer(f"_in_idx_{ds}", torch.tensor(self._internal_input_idx[ds], dtype=torch.long), persistent=False)
            self.register_buffer(f"_out_idx_{ds}", torch.tensor(self._internal_output_idx[ds], dtype=torch.long), persistent=False)
            cmap = None
            if len(set(self._internal_output_idx[ds])) == len(self._internal_output_idx[ds]):  # one residual per output column
                cmap = torch.full((self.num_output_channels[ds],), -1, dtype=torch.int32)
                cmap[torch.tensor(self._internal_output_idx[ds], dtype=torch.long)] = torch.tensor(self._internal_input_idx[ds], dtype=torch.int32)
            self.register_buffer(f"_col_map_{ds}", cmap, persistent=False)

    def _build_networks(self, mc, edges) -> None:
        hid = self._graph_name_hidden
        n = self.node_attributes.num_nodes
        self.encoder_graph_provider, self.encoder = nn.ModuleDict(), nn.ModuleDict()
        for ds in self.dataset_names:
            self.encoder_graph_provider[ds] = create_graph_provider(
                graph=edges[(ds, "to", hid)], edge_attributes=mc.encoder.get("sub_graph_edge_attributes"),
                src_size=n[ds], dst_size=n[hid], trainable_size=mc.encoder.get("trainable_size", 0))
            self.encoder[ds] = instantiate(_retarget(mc.encoder), _recursive_=False, in_channels_src=self.input_dim[ds],
                                           in_channels_dst=self.input_dim_latent, hidden_dim=self.num_channels,
                                           edge_dim=self.encoder_graph_provider[ds].edge_dim)
        self.processor_graph_provider = create_graph_provider(
            graph=edges[(hid, "to", hid)], edge_attributes=mc.processor.get("sub_graph_edge_attributes"),
            src_size=n[hid], dst_size=n[hid], trainable_size=mc.processor.get("trainable_size", 0))
        self.processor = instantiate(_retarget(mc.processor), _recursive_=False, num_channels=self.num_channels,
                                     edge_dim=self.processor_graph_provider.edge_dim)
        self.decoder_graph_provider, self.decoder = nn.ModuleDict(), nn.ModuleDict()
        for ds in self.dataset_names:
            self.decoder_graph_provider[ds] = create_graph_provider(
                graph=edges[(hid, "to", ds)], edge_attributes=mc.decoder.get("sub_graph_edge_attributes"),
                src_size=n[hid], dst_size=n[ds], trainable_size=mc.decoder.get("trainable_size", 0))
            self.decoder[ds] = instantiate(_retarget(mc.decoder), _recursive_=False, in_channels_src=self.num_channels,
                                           in_channels_dst=self.target_dim[ds], hidden_dim=self.num_channels,
                                           out_channels_dst=self.output_dim[ds], edge_dim=self.decoder_graph_provider[ds].edge_dim)

    def _build_boundings(self, cfgs) -> nn.ModuleDict:
        """models/base.py:94-97 / layers/bounding.py:312-400: one ModuleList per dataset, configuration order."""
        from ..layers.bounding import build_boundings_for

        out = nn.ModuleDict()
        for ds in self.dataset_names:
            stats = self.statistics.get(ds) if isinstance(self.statistics, dict) else self.statistics
            out[ds] = build_boundings_for(cfgs, self.data_indices[ds].model.output.name_to_index, stats,
                                          self.data_indices[ds].data.input.name_to_index)
        return out

    # -- glue (encoder_processor_decoder.py:98-163) ---------------------------------------------------------
    def _assemble_input(self, x: Tensor, batch_size: int, shard_sizes, group, ds: str, norm=None):
        """``norm``: the dataset's InputNormalizer when ``predict_step`` fuses it (x is then the RAW batch).  Returns
        (x_data_latent, x_skip, skip_is_raw)."""
        node_attr = self.node_attributes(ds, batch_size=batch_size)
        if shard_sizes is not None:
            node_attr = shard_tensor(node_attr, 0, shard_sizes, group)
        B, T, E, N, V = x.shape
        fusable = (_FUSED_INPUT and B == 1 and E == 1 and x.is_cuda and x.stride(4) == 1
                   and not (torch.is_grad_enabled() and (x.requires_grad or node_attr.requires_grad)))
        if fusable and (node_attr.dtype == x.dtype or (norm is not None and x.dtype == torch.float32)):
            # inference, one member: permute + cat + alignment zeros (+ the input normaliser as a column program) in ONE kernel
            width = T * V + node_attr.shape[1]
            if node_attr.dtype != torch.float32 and self._prepad(ds):
                width += (-width) % 64 if (-width) % 64 <= 16 and _PAD64 else (-width) % 8
            if norm is None:
                return ops.assemble_input(x[0, :, 0], node_attr, width), x[:, self._skip_step, ...], False
            mul, add = norm.column_program(V)
            return ops.assemble_input(x[0, :, 0], node_attr, width, mul, add, out_dtype=node_attr.dtype), x[:, self._skip_step, ...], True
        if norm is not None:  # not fusable: the normaliser's own kernel, then the generic path
            x = norm.transform(x, in_place=False).to(node_attr.dtype)
        x_skip = x[:, self._skip_step, ...]  # SkipConnection (layers/residual.py:60-81): last input step
        flat = x.permute(0, 2, 3, 1, 4).reshape(B * E * N, T * V)  # "(batch ensemble grid) (time vars)"
        cols = [flat, node_attr.to(flat.dtype)]
        width = flat.shape[1] + node_attr.shape[1]
        pad = (-width) % 64 if (-width) % 64 <= 16 and _PAD64 else (-width) % 8  # a K multiple of 64 takes the DMA-ring GEMM kernels
        if pad and flat.dtype != torch.float32 and self._prepad(ds):
            # 16-bit GEMM operand rows must be 16-byte aligned: the zero columns the embeddings of encoder (source) and decoder
            # (destination) would each append with a pad kernel are written by this cat instead (PaddedLinear takes such rows)
            key = (flat.shape[0], pad, flat.dtype, str(flat.device))
            if self._pad_zeros is None or self._pad_zeros[0] != key:
                self._pad_zeros = (key, torch.zeros((flat.shape[0], pad), dtype=flat.dtype, device=flat.device))
            cols.append(self._pad_zeros[1])
        return torch.cat(cols, dim=-1), x_skip, False

    def _prepad(self, ds: str) -> bool:
        """The GraphTransformer mappers embed their inputs through PaddedLinear, which accepts rows that already carry the
        alignment zeros; the GNN mappers (MLP embeddings) take the exact width."""
        return isinstance(self.encoder[ds], GraphTransformerBaseMapper) and isinstance(self.decoder[ds], GraphTransformerBaseMapper)

    def _hidden_attributes(self, batch_size: int) -> Tensor:
        """Node attributes of the hidden mesh; outside training (a static tensor) with the zero columns of the 16-byte row
        alignment already appended, once."""
        x = self.node_attributes(self._graph_name_hidden, batch_size=batch_size)
        # a handful of attribute columns: pad K to 64 (a static [N_hidden, 64] tensor) so that the hidden-node embedding takes the
        # DMA-ring kernels and can emit the row statistics of the encoder's destination LayerNorm
        pad = (-x.shape[1]) % 64 if _PAD64 and x.shape[1] < 64 else (-x.shape[1]) % 8
        if not pad or x.dtype == torch.float32 or (torch.is_grad_enabled() and x.requires_grad) or not all(self._prepad(ds) for ds in self.dataset_names):
            return x
        key = (x.data_ptr(), version(x), tuple(x.shape), x.dtype)
        if self._hidden_padded is None or self._hidden_padded[0] != key:
            self._hidden_padded = (key, torch.nn.functional.pad(x, (0, pad)), x)
        return self._hidden_padded[1]

    def _assemble_output(self, x_out: Tensor, x_skip: Tensor, batch_size: int, ensemble_size: int, dtype, ds: str, norm=None,
                         skip_is_raw: bool = False, denorm=None) -> Tensor:
        """``norm`` / ``skip_is_raw``: x_skip is the RAW input and is normalised inside the residual kernel; ``denorm``: the
        output InputNormalizer whose inverse transform is appended to the column program of the boundings."""
        N = x_out.shape[0] // (batch_size * ensemble_size)
        in_idx, out_idx = getattr(self, f"_in_idx_{ds}"), getattr(self, f"_out_idx_{ds}")
        col_map = getattr(self, f"_col_map_{ds}")
        fusable = (col_map is not None and batch_size == 1 and ensemble_size == 1 and self.n_step_output == 1 and x_out.is_cuda
                   and not (torch.is_grad_enabled() and (x_out.requires_grad or x_skip.requires_grad)))
        fused_norm = fusable and skip_is_raw and x_skip.dtype == dtype and dtype in (x_out.dtype, torch.float32)
        if skip_is_raw and not fused_norm:  # every path but the fused-normalisation kernel takes a NORMALISED skip
            x_skip, skip_is_raw = norm.transform(x_skip, in_place=False), False
        if fused_norm:
            mul, add = norm.column_program(x_skip.shape[-1])
            x_out = ops.assemble_output(x_out, x_skip.reshape(N, -1), col_map, mul, add).view(1, 1, 1, N, -1)
        elif fusable and x_out.dtype == dtype and x_skip.dtype == dtype:
            # one kernel for cast / residual on the prognostic columns (instead of clone + index_select + index_add_)
            x_out = ops.assemble_output(x_out, x_skip.reshape(N, -1), col_map).view(1, 1, 1, N, -1)
        else:
            x_out = x_out.reshape(batch_size, ensemble_size, N, self.n_step_output, -1).permute(0, 3, 1, 2, 4).to(dtype=dtype).clone()
            # SkipConnection._expand_time (layers/residual.py:53-57): the skip is repeated over the output steps
            skip = x_skip.unsqueeze(1).expand(-1, self.n_step_output, -1, -1, -1)
            x_out.index_add_(-1, out_idx, skip.index_select(-1, in_idx).to(dtype))
        if len(self.boundings[ds]) or denorm is not None:
            # all configured boundings (configuration order) and, fused, the output de-normalisation as ONE in-place column program
            from ..layers.bounding import apply_program_torch, program_tables

            if torch.is_grad_enabled() and x_out.requires_grad:
                x_out = apply_program_torch(x_out, [op for b in self.boundings[ds] for op in b.program()])
                if denorm is not None:
                    x_out = denorm.inverse_transform(x_out, in_place=False)
            else:
                # the column program and its device tables are built ONCE per (dataset, device, normaliser state): program() of
                # the normalised boundings reads a registered buffer (.tolist() = a device->host sync, illegal under hipGraph capture)
                key = (ds, str(x_out.device), None if denorm is None else (id(denorm), denorm._norm_mul._version, denorm._norm_add._version))
                if key not in self._bound_tables:
                    prog = [op for b in self.boundings[ds] for op in b.program()]
                    if denorm is not None:
                        prog += denorm.inverse_program(x_out.shape[-1])
                    self._bound_tables[key] = program_tables(prog, x_out.device)
                ops.bound_columns_(x_out, *self._bound_tables[key])
        return x_out

    @scoped_forward
    def forward(self, x: dict, *, model_comm_group=None, grid_shard_sizes: Optional[dict] = None, _fused_norm: Optional[dict] = None,
                **kwargs) -> dict:
        """``_fused_norm`` (set by ``predict_step`` only): {dataset: (input normaliser, output normaliser)} - x is then the RAW
        batch and the output is de-normalised."""
        names = list(x.keys())
        batch_size = x[names[0]].shape[0]
        ensemble_size = x[names[0]].shape[2]
        if ensemble_size != 1:
            # the reference's class fails on it as well (node attributes are repeated batch_size times against batch * ensemble *
            # grid input rows: torch.cat mismatch at encoder_processor_decoder.py:121); its ensemble model folds members into the batch
            raise ValueError(f"AnemoiModelEncProcDec: ensemble dimension {ensemble_size} != 1; fold the members into the batch dimension")
        in_out_sharded = {ds: grid_shard_sizes is not None and grid_shard_sizes.get(ds) is not None for ds in names}
        if model_comm_group is not None and comm_size(model_comm_group) > 1:
            assert batch_size == 1, "Only batch size of 1 is supported when model is sharded across GPUs"
            assert ensemble_size == 1, "Ensemble size per device must be 1 when model is sharded across GPUs"
        hid = self._graph_name_hidden
        x_hidden_latent = self._hidden_attributes(batch_size)
        shard_sizes_hidden = get_shard_sizes(x_hidden_latent, 0, model_comm_group)
        x_hidden_latent = shard_tensor(x_hidden_latent, 0, shard_sizes_hidden, model_comm_group)
        latents, skips, data_latents, data_shards = {}, {}, {}, {}
        # LayerNorm fold across the encoder / processor boundary (inference, one dataset, GraphTransformer on both sides): the
        # encoder's last GEMM leaves the row statistics the processor's first LayerNorm needs
        from ..layers.mapper import GraphTransformerForwardMapper
        from ..layers.processor import GraphTransformerProcessor

        chain_kw = ({"ln_chain": {"next_block": self.processor.proc[0]}} if len(names) == 1 and isinstance(self.encoder[names[0]], GraphTransformerForwardMapper)
                    and isinstance(self.processor, GraphTransformerProcessor) and model_comm_group is None else
                    ({"ln_chain": {}} if len(names) == 1 and isinstance(self.encoder[names[0]], GraphTransformerForwardMapper)
                     and isinstance(self.processor, GraphTransformerProcessor) else {}))
        for ds in names:
            shard_sizes_data = grid_shard_sizes[ds] if in_out_sharded[ds] else None
            norm_in, norm_out = (_fused_norm or {}).get(ds, (None, None))
            x_data_latent, x_skip, raw = self._assemble_input(x[ds], batch_size, shard_sizes_data, model_comm_group, ds, norm=norm_in)
            skips[ds], data_shards[ds] = (x_skip, raw, norm_in, norm_out), shard_sizes_data
            ea, ei, es = self.encoder_graph_provider[ds].get_edges(batch_size=batch_size, model_comm_group=model_comm_group)
            info = BipartiteGraphShardInfo(src_nodes=shard_sizes_data, dst_nodes=shard_sizes_hidden, edges=es)
            x_data_latent, x_latent = self.encoder[ds]((x_data_latent, x_hidden_latent.to(x_data_latent.dtype)), batch_size=batch_size,
                                                       shard_info=info, edge_attr=ea, edge_index=ei,
                                                       model_comm_group=model_comm_group, keep_x_dst_sharded=True, **chain_kw)
            data_latents[ds], latents[ds] = x_data_latent, x_latent
        x_latent = latents[names[0]] if len(names) == 1 else sum(latents.values())
        ea, ei, es = self.processor_graph_provider.get_edges(batch_size=batch_size, model_comm_group=model_comm_group)
        # the latent skip (:295-296) rides in the last processor block's last GEMM (GraphTransformer processor), else one add
        fuse_skip = self.latent_skip and isinstance(self.processor, GraphTransformerProcessor)
        # (one dataset, GraphTransformer decoder, unsharded inference: the decoder block's layer_norm_attention_src + k|v projection of the hidden
        # rows ride at the end of the last processor block's chain launch, behind the latent skip)
        from ..layers.mapper import GraphTransformerBackwardMapper

        dec0 = self.decoder[names[0]]
        hand_kv = (_TAIL_KV and fuse_skip and len(names) == 1 and model_comm_group is None and "ln_chain" in chain_kw and isinstance(dec0, GraphTransformerBackwardMapper)
                   and not torch.is_grad_enabled())
        x_latent_proc = self.processor(x=x_latent, batch_size=batch_size, shard_info=GraphShardInfo(nodes=shard_sizes_hidden, edges=es),
                                       edge_attr=ea, edge_index=ei, model_comm_group=model_comm_group, **chain_kw,
                                       **({"latent_skip": x_latent} if fuse_skip else {}), **({"after_last_block": dec0.proc} if hand_kv else {}))
        src_proj = None
        if hand_kv and chain_kw["ln_chain"].get("qkvs_x") is x_latent_proc:
            src_proj = (x_latent_proc, chain_kw["ln_chain"]["qkvs"])
        if self.latent_skip and not fuse_skip:
            if (x_latent_proc.is_cuda and x_latent_proc.dim() == 2 and x_latent_proc.shape == x_latent.shape and x_latent_proc.dtype == x_latent.dtype
                    and not (torch.is_grad_enabled() and (x_latent_proc.requires_grad or x_latent.requires_grad))):
                from ..layers.block import _identity_index  # inference (GNN processor): the add as one launch of this library, not a torch kernel

                x_latent_proc = ops.gather_add_rows(x_latent_proc, x_latent, _identity_index(x_latent))
            else:
                x_latent_proc = x_latent_proc + x_latent
        out = {}
        for ds in names:
            ea, ei, es = self.decoder_graph_provider[ds].get_edges(batch_size=batch_size, model_comm_group=model_comm_group)
            info = BipartiteGraphShardInfo(src_nodes=shard_sizes_hidden, dst_nodes=data_shards[ds], edges=es)
            x_out = self.decoder[ds]((x_latent_proc, data_latents[ds]), batch_size=batch_size, shard_info=info, edge_attr=ea,
                                     edge_index=ei, model_comm_group=model_comm_group, keep_x_dst_sharded=in_out_sharded[ds],
                                     **({"src_proj": src_proj} if src_proj is not None else {}))
            x_skip, raw, norm_in, norm_out = skips[ds]
            out[ds] = self._assemble_output(x_out, x_skip, batch_size, ensemble_size, x[ds].dtype, ds, norm=norm_in, skip_is_raw=raw,
                                            denorm=norm_out)
        return out

    # -- predict_step (models/base.py:303-391) ----------------------------------------------------------------------------
    @staticmethod
    def _sole_normalizer(procs, inverse: bool):
        """The InputNormalizer of a ``Processors`` chain that consists of nothing else (what the model edge can fuse), else None."""
        from ..preprocessing import InputNormalizer, Processors

        if isinstance(procs, InputNormalizer):
            return procs
        if isinstance(procs, Processors) and procs.inverse == inverse and len(procs.processors) == 1:
            only = next(iter(procs.processors.values()))
            if isinstance(only, InputNormalizer):
                return only
        return None

    def predict_step(self, batch: dict, pre_processors, post_processors, n_step_input: int, model_comm_group=None,
                     gather_out: bool = True, **kwargs) -> dict:
        """Pre-process, forward, post-process (reference models/base.py:303-391): ``batch[ds]`` is [batch, time, grid, vars] of
        RAW data.  When a dataset's pre- / post-processor chains are just the input normaliser, it is fused into the model
        edges: the transform runs inside the input assembly kernel (and on the skip connection inside the residual kernel),
        the inverse inside the output column program - no pass of its own over the data."""
        from ..distributed.primitives import gather_tensor

        with torch.no_grad():
            names = list(batch.keys())
            for ds in names:
                assert len(batch[ds].shape) == 4, (
                    f"The {ds} input tensor has an incorrect shape: expected a 4-dimensional tensor, got {batch[ds].shape}!")
            x = {ds: batch[ds][:, 0:n_step_input, None, ...] for ds in names}  # dummy ensemble dimension as 3rd index
            grid_shard_sizes = None
            if model_comm_group is not None:
                grid_shard_sizes = {ds: get_shard_sizes(x[ds], -2, model_comm_group) for ds in names}
                x = {ds: shard_tensor(x[ds], -2, grid_shard_sizes[ds], model_comm_group) for ds in names}
            fused = {}
            for ds in names:
                pre, post = self._sole_normalizer(pre_processors[ds], False), self._sole_normalizer(post_processors[ds], True)
                if pre is not None and post is not None and x[ds].is_cuda:
                    fused[ds] = (pre, post)
                    check = getattr(pre_processors[ds], "check_first_batch", None)
                    if check is not None and getattr(pre_processors[ds], "_awaiting_first_batch", False):
                        check(pre.transform(x[ds], in_place=False))  # the chain's one-off NaN check survives the fusion
                else:
                    x[ds] = pre_processors[ds](x[ds], in_place=False)
            y_hat = self.forward(x, model_comm_group=model_comm_group, grid_shard_sizes=grid_shard_sizes, _fused_norm=fused, **kwargs)
            for ds in names:
                if ds not in fused:
                    y_hat[ds] = post_processors[ds](y_hat[ds], in_place=False)
            if gather_out and model_comm_group is not None:
                for ds in names:
                    y_hat[ds] = gather_tensor(y_hat[ds], -2, grid_shard_sizes[ds], model_comm_group)
        return y_hat
