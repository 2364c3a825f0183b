"""Config / index builders for the benchmark and the tests: the nested ``model_config`` the reference's
AnemoiModelEncProcDec is instantiated from (training/src/anemoi/training/config/model/graphtransformer.yaml) and a
minimal ``data_indices`` object (models/tests/models/test_base_graph_model.py:33-56)."""
from types import SimpleNamespace


class IndexGroup(SimpleNamespace):
    def __len__(self):
        return len(self.full)


def make_data_indices(n_vars_in, n_prog):
    names = {f"v{i}": i for i in range(n_vars_in)}
    prog = list(range(n_prog))
    ns = SimpleNamespace(
        model=SimpleNamespace(
            input=IndexGroup(prognostic=prog, full=list(range(n_vars_in)), name_to_index=names),
            output=IndexGroup(prognostic=prog, full=prog, diagnostic=[], name_to_index={f"v{i}": i for i in prog}),
            _forcing=[],
        ),
        data=SimpleNamespace(input=SimpleNamespace(name_to_index=names)),
        name_to_index=names,
    )
    return {"data": ns}


def model_config(kind, num_channels, num_layers, num_heads, trainable, prefix="anemoi.models.layers"):
    """Same nested config the reference model is built from (tests/golden/make_golden.py); by default it even carries
    the REFERENCE's ``_target_`` strings, which the model retargets to the MI355X classes."""
    common = dict(cpu_offload=False, gradient_checkpointing=False, layer_kernels=None, trainable_size=trainable,
                  sub_graph_edge_attributes=["edge_length", "edge_dirs"])
    if kind == "gt":
        common.update(num_heads=num_heads, mlp_hidden_ratio=4, qk_norm=False, shard_strategy="edges",
                      graph_attention_backend="pyg", edge_pre_mlp=False)
        enc = dict(common, _target_=f"{prefix}.mapper.GraphTransformerForwardMapper", num_chunks=2)
        proc = dict(common, _target_=f"{prefix}.processor.GraphTransformerProcessor", num_chunks=1, num_layers=num_layers)
        dec = dict(common, _target_=f"{prefix}.mapper.GraphTransformerBackwardMapper", num_chunks=2, initialise_data_extractor_zero=False)
    else:
        common.update(mlp_extra_layers=0)
        enc = dict(common, _target_=f"{prefix}.mapper.GNNForwardMapper", num_chunks=1)
        proc = dict(common, _target_=f"{prefix}.processor.GNNProcessor", num_chunks=1, num_layers=num_layers)
        dec = dict(common, _target_=f"{prefix}.mapper.GNNBackwardMapper", num_chunks=1)
    return {
        "model": {
            "num_channels": num_channels,
            "trainable_parameters": {"data": trainable, "hidden": trainable, "data2hidden": trainable, "hidden2data": trainable, "hidden2hidden": trainable},
            "model": {"hidden_nodes_name": "hidden", "latent_skip": True},
            "encoder": enc, "processor": proc, "decoder": dec,
            "residual": {"_target_": "anemoi.models.layers.residual.SkipConnection", "step": -1},
            "bounding": [],
        }
    }


