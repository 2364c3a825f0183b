"""Shared builders for the module tests: construct this package's modules from the golden fixtures' configs."""
import torch

from anemoi_core_amd.layers.utils import load_layer_kernels


from anemoi_core_amd.models.configs import IndexGroup, make_data_indices, model_config  # noqa: E402,F401


def build_model_from_fixture(case):
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec

    cfg = case["cfg"]
    g = build_synthetic_graph(cfg["data_grid"], cfg["hidden_resolution"])
    model = AnemoiModelEncProcDec(
        model_config=model_config(cfg["kind"], cfg["num_channels"], cfg["num_layers"], cfg["num_heads"], cfg["trainable"]),
        data_indices=make_data_indices(cfg["n_vars"], cfg["n_vars"]), statistics={"data": None},
        n_step_input=cfg["n_step_input"], n_step_output=1, graph_data=g,
    ).eval()
    return model, g


def lk():
    return load_layer_kernels()
