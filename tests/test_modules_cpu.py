"""CPU (host-logic) tests of the nn.Module mirror: constructors accept the reference's keywords and the state_dict keys
and shapes equal the reference's (the golden fixtures hold the reference's own state_dicts) — the drop-in contract of
SURVEY.md §8(b).  No kernel runs here."""
import pytest
import torch

from anemoi_core_amd.layers.block import (
    GraphConvMapperBlock,
    GraphConvProcessorBlock,
    GraphTransformerMapperBlock,
    GraphTransformerProcessorBlock,
)
from anemoi_core_amd.layers.mapper import (
    GNNBackwardMapper,
    GNNForwardMapper,
    GraphTransformerBackwardMapper,
    GraphTransformerForwardMapper,
)
from anemoi_core_amd.layers.processor import GNNProcessor, GraphTransformerProcessor
from tests.helpers import build_model_from_fixture, lk


def assert_same_state_dict(module, ref_params):
    sd = module.state_dict()
    assert list(sd.keys()) == list(ref_params.keys()), set(sd) ^ set(ref_params)
    for k, v in ref_params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    module.load_state_dict(ref_params, strict=True)


@pytest.mark.parametrize("tag,cls", [("proc_qknorm", GraphTransformerProcessorBlock), ("proc", GraphTransformerProcessorBlock),
                                     ("map", GraphTransformerMapperBlock), ("map_qknorm_updsrc", GraphTransformerMapperBlock),
                                     ("gconv_proc", GraphConvProcessorBlock), ("gconv_proc_emb", GraphConvProcessorBlock),
                                     ("gconv_map", GraphConvMapperBlock), ("gconv_map_updsrc", GraphConvMapperBlock)])
def test_block_state_dicts_match_reference(golden, tag, cls):
    c = golden("blocks.pt")[tag]
    assert_same_state_dict(cls(layer_kernels=lk(), **c["cfg"]), c["params"])


@pytest.mark.parametrize("tag,cls", [("gt_processor", GraphTransformerProcessor), ("gt_forward_mapper", GraphTransformerForwardMapper),
                                     ("gt_backward_mapper", GraphTransformerBackwardMapper), ("gnn_processor", GNNProcessor),
                                     ("gnn_forward_mapper", GNNForwardMapper), ("gnn_backward_mapper", GNNBackwardMapper)])
def test_processor_mapper_state_dicts_match_reference(golden, tag, cls):
    c = golden("proc_mappers.pt")[tag]
    assert_same_state_dict(cls(**c["cfg"]), c["params"])  # the reference's own constructor kwargs, incl. backend "pyg"


@pytest.mark.parametrize("kind", ["gt", "gnn"])
def test_model_state_dict_matches_reference(golden, kind):
    c = golden("model_tiny.pt")[kind]
    model, _ = build_model_from_fixture(c)  # config carries the REFERENCE's _target_ strings
    assert_same_state_dict(model, c["params"])
    assert type(model.processor).__module__.startswith("anemoi_core_amd.")


def test_mapper_block_layer_norm_alias():
    blk = GraphTransformerMapperBlock(in_channels=64, hidden_dim=128, out_channels=64, num_heads=4, edge_dim=3, layer_kernels=lk())
    assert blk.layer_norm_attention_dest is blk.layer_norm_attention  # reference block.py:940-941
    sd = blk.state_dict()
    assert "layer_norm_attention.weight" in sd and "layer_norm_attention_dest.weight" in sd


def test_constructor_validation_matches_reference():
    with pytest.raises(ValueError, match="divisible by num_heads"):
        GraphTransformerProcessorBlock(in_channels=64, hidden_dim=64, out_channels=64, num_heads=5, edge_dim=3, layer_kernels=lk())
    with pytest.raises(AssertionError, match="divisible by the number of processor chunks"):
        GraphTransformerProcessor(num_layers=3, num_channels=64, num_chunks=2, num_heads=4, mlp_hidden_ratio=4, edge_dim=3)
    with pytest.raises(AssertionError, match="does not support out_channels_dst"):
        GraphTransformerForwardMapper(in_channels_src=3, in_channels_dst=3, hidden_dim=64, out_channels_dst=5, num_chunks=1,
                                      num_heads=4, mlp_hidden_ratio=4, edge_dim=3)
    with pytest.raises(NotImplementedError):
        from anemoi_core_amd.layers.utils import load_layer_kernels

        load_layer_kernels({"Linear": {"_target_": "torch.nn.Bilinear"}})


def test_attention_dropout_is_a_training_mode_feature_of_the_hip_path():
    """conv.py:94,145: dropout on the attention weights only acts in training mode; it runs in the HIP kernels
    (tests/test_attention_dropout_gpu.py), so a CPU tensor is refused like everywhere else."""
    from anemoi_core_amd.layers.conv import GraphTransformerConv

    conv = GraphTransformerConv(out_channels=8, dropout=0.1)
    assert conv.dropout == 0.1
    q = torch.zeros(3, 2, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        conv.train()(q, q, q, None, torch.zeros(2, 0, dtype=torch.long))
    with pytest.raises(ValueError):
        GraphTransformerConv(out_channels=8, dropout=1.5)


def test_forward_on_cpu_fails_loudly(golden):
    """No silent CPU fallback: the product modules refuse CPU tensors."""
    c = golden("blocks.pt")["proc"]
    blk = GraphTransformerProcessorBlock(layer_kernels=lk(), **c["cfg"]).eval()
    from anemoi_core_amd.distributed.shapes import GraphShardInfo

    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(c["x"], c["edge_attr"], c["edge_index"], GraphShardInfo(), 1, c["x"].shape[0])


def test_assemble_output_repeats_skip_over_output_steps():
    """n_step_output > 1: the skip connection is repeated over the output steps (reference layers/residual.py:53-57,
    models/encoder_processor_decoder.py:131-158): x_out[..., out_idx] += expand(x_skip)[..., in_idx]."""
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec
    from tests.helpers import make_data_indices, model_config

    g = build_synthetic_graph("o8", 2)
    T_out, V_in, V_prog = 2, 5, 3
    model = AnemoiModelEncProcDec(model_config=model_config("gt", 32, 1, 4, 2), data_indices=make_data_indices(V_in, V_prog),
                                  statistics={"data": None}, n_step_input=2, n_step_output=T_out, graph_data=g).eval()
    N = g.num_data
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 2, 1, N, V_in, generator=gen)
    dec_out = torch.randn(N, T_out * V_prog, generator=gen)  # "(batch ensemble grid) (time vars)"
    x_skip = x[:, -1, ...]
    got = model._assemble_output(dec_out, x_skip, 1, 1, torch.float32, "data")
    want = dec_out.view(1, 1, N, T_out, V_prog).permute(0, 3, 1, 2, 4).clone()
    want[..., list(range(V_prog))] += x_skip.unsqueeze(1).expand(-1, T_out, -1, -1, -1)[..., list(range(V_prog))]
    assert got.shape == (1, T_out, 1, N, V_prog)
    torch.testing.assert_close(got, want, rtol=0, atol=0)


def test_scoped_forward_finds_the_group_positionally_and_by_keyword(monkeypatch):
    """distributed.primitives.scoped_forward wraps a module's forward in forward_scope(model_comm_group) - the device-initiated
    wire's "one forward" marker - wherever the caller put the group; without a group nothing is entered."""
    import contextlib

    from anemoi_core_amd.distributed import primitives as P

    seen = []

    @contextlib.contextmanager
    def scope(group):
        seen.append(group)
        yield

    monkeypatch.setattr(P, "forward_scope", scope)

    class M:
        @P.scoped_forward
        def forward(self, x, batch_size, model_comm_group=None, flag=True):
            return x + 1

        @P.scoped_forward
        def kw_only(self, x, *, model_comm_group=None):
            return x + 2

    m = M()
    assert m.forward(1, 2) == 2 and seen == []
    assert m.forward(1, 2, "g0") == 2 and m.forward(1, 2, model_comm_group="g1") == 2 and m.kw_only(1, model_comm_group="g2") == 3
    assert seen == ["g0", "g1", "g2"] and m.kw_only(1) == 3 and len(seen) == 3
