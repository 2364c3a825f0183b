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
        data_indices=make_data_indices(cfg["n_vars"], cfg.get("n_prog", cfg["n_vars"])), statistics={"data": None},
        n_step_input=cfg["n_step_input"], n_step_output=cfg.get("n_step_output", 1), graph_data=g,
    ).eval()
    return model, g


def lk():
    return load_layer_kernels()


def indices_from_fixture(idx):
    """An IndexCollection-shaped object (what the reference's `data_indices[dataset]` offers to the model and to the
    normaliser) from the index tensors / name maps a fixture stores (tests/golden/make_golden.py: gen_edges)."""
    from types import SimpleNamespace

    grp = lambda full, n2i, **kw: IndexGroup(full=full, name_to_index=n2i, **kw)  # noqa: E731
    return SimpleNamespace(
        data=SimpleNamespace(input=grp(idx["data_input_full"], idx["data_input_name_to_index"]),
                             output=grp(idx["data_output_full"], idx["data_output_name_to_index"])),
        model=SimpleNamespace(input=grp(idx["model_input_full"], idx["model_input_name_to_index"], prognostic=idx["model_input_prognostic"]),
                              output=grp(idx["model_output_full"], idx["model_output_name_to_index"], prognostic=idx["model_output_prognostic"],
                                         diagnostic=[]),
                              _forcing=[]),
        name_to_index=idx["data_input_name_to_index"])
