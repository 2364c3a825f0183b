"""Model edges on the MI355X (scope row f4): the input normaliser as HIP kernels and fused into the model's input / output
assembly (`AnemoiModelEncProcDec.predict_step`), against the reference's own outputs (fixture tests/golden/edges.pt:
preprocessing/normalizer.py and models/base.py:303-391 run in the build container)."""
import pytest
import torch

from anemoi_core_amd import ops
from anemoi_core_amd.preprocessing import InputNormalizer, Processors
from tests.helpers import indices_from_fixture, model_config

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _normalizer(c):
    stats = {k: v.numpy().copy() for k, v in c["statistics"].items()}
    return InputNormalizer(config=c["data_config"]["normalizer"], data_indices=indices_from_fixture(c["indices"]), statistics=stats).to(DEV)


def test_normalizer_kernels_equal_reference_bitwise(golden):
    """fp32: x.mul_(mul).add_(add) and x.subtract_(add).div_(mul) are two IEEE roundings each - the kernel reproduces them."""
    c = golden("edges.pt")["normalizer"]
    nm = _normalizer(c)
    d = lambda k: c[k].to(DEV)  # noqa: E731
    assert torch.equal(nm.transform(d("x_all"), in_place=False).cpu(), c["t_all"])
    assert torch.equal(nm.inverse_transform(d("x_all"), in_place=False).cpu(), c["i_all"])
    assert torch.equal(nm.transform(d("x_in"), in_place=False).cpu(), c["t_in"])
    assert torch.equal(nm.inverse_transform(d("x_out"), in_place=False).cpu(), c["i_out"])
    sub = d("x_all")[..., c["data_index"]].contiguous()
    assert torch.equal(nm.transform(sub, in_place=False, data_index=c["data_index"]).cpu(), c["t_idx"])
    assert torch.equal(nm.inverse_transform(sub, in_place=False, data_index=c["data_index"]).cpu(), c["i_idx"])
    x = d("x_in").clone()
    assert nm.transform(x) is x and torch.equal(x.cpu(), c["t_in"])  # in place
    # 16-bit tensors: each of the two steps rounds to the tensor's dtype, as torch's in-place ops do
    xb = d("x_in").to(torch.bfloat16)
    mul, add = nm.column_program(xb.shape[-1])
    want = xb.clone().mul_(mul).add_(add)
    assert torch.equal(nm.transform(xb, in_place=False), want)


def _model_and_processors(c, dtype=torch.float32):
    from anemoi_core_amd.graphs.synthetic import build_synthetic_graph
    from anemoi_core_amd.models import AnemoiModelEncProcDec

    cfg = c["cfg"]
    g = build_synthetic_graph(cfg["data_grid"], cfg["hidden_resolution"])
    di = indices_from_fixture(c["indices"])
    stats = {k: v.numpy().copy() for k, v in c["statistics"].items()}
    model = AnemoiModelEncProcDec(model_config=model_config(cfg["kind"], cfg["num_channels"], cfg["num_layers"], cfg["num_heads"], cfg["trainable"]),
                                  data_indices={"data": di}, statistics={"data": stats}, n_step_input=cfg["n_step_input"], n_step_output=1,
                                  graph_data=g).eval()
    model.load_state_dict(c["params"], strict=True)
    nm = InputNormalizer(config=c["data_config"]["normalizer"], data_indices=di, statistics=stats).to(DEV)
    pre, post = Processors([["normalizer", nm]]), Processors([["normalizer", nm]], inverse=True)
    return model.to(DEV).to(dtype), pre, post


def test_predict_step_fused_normaliser_equals_reference(golden, monkeypatch):
    """models/base.py:303-391 on raw data: normalise -> forward -> de-normalise.  The fused path must not launch the
    normaliser's own kernel at all (its arithmetic rides in assemble_input / assemble_output / the column program)."""
    c = golden("edges.pt")["predict_step"]
    model, pre, post = _model_and_processors(c)
    batch = {"data": c["batch"].to(DEV)}
    unfused = model.predict_step(batch, {"data": _Opaque(pre)}, {"data": _Opaque(post)}, c["cfg"]["n_step_input"])["data"]

    def boom(*a, **k):
        raise AssertionError("the stand-alone normaliser kernel ran inside the fused predict_step")

    monkeypatch.setattr(ops, "affine_columns", boom)
    fused = model.predict_step(batch, {"data": pre}, {"data": post}, c["cfg"]["n_step_input"])["data"]
    assert fused.shape == c["out"].shape and fused.dtype == torch.float32
    scale = float(c["out"].abs().max())
    assert float((fused.cpu() - c["out"]).abs().max()) <= 2e-5 * max(1.0, scale)      # vs the REFERENCE's predict_step
    assert float((unfused.cpu() - c["out"]).abs().max()) <= 2e-5 * max(1.0, scale)
    assert float((fused - unfused).abs().max()) <= 2e-6 * max(1.0, scale)              # same arithmetic, fused or not
    # the input is untouched (in_place=False semantics of predict_step)
    assert torch.equal(batch["data"].cpu(), c["batch"])


class _Opaque(torch.nn.Module):
    """Hides a Processors chain from the fusion (any non-normaliser processor would do the same): forces the generic path."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, x, **kw):
        return self.inner(x, **kw)


def test_predict_step_bf16_model_on_fp32_data(golden):
    """fp32 batch into a bf16 model (the reference's autocast regime): the fused edges convert in-kernel, the residual and the
    output stay fp32 like the reference's (`x_out.to(dtype=x.dtype)`)."""
    c = golden("edges.pt")["predict_step"]
    model, pre, post = _model_and_processors(c, torch.bfloat16)
    out = model.predict_step({"data": c["batch"].to(DEV)}, {"data": pre}, {"data": post}, c["cfg"]["n_step_input"])["data"]
    assert out.dtype == torch.float32 and out.shape == c["out"].shape
    err = (out.cpu() - c["out"]).abs()
    scale = max(1.0, float(c["out"].abs().max()))
    assert float(err.max()) <= 6e-2 * scale and float(err.mean()) <= 1e-2 * scale


def test_predict_step_is_capturable_as_hip_graph(golden):
    """The fused edges do no host work per call (tables cached): predict_step replays from a hipGraph bit-identically."""
    c = golden("edges.pt")["predict_step"]
    model, pre, post = _model_and_processors(c)
    batch = {"data": c["batch"].to(DEV)}
    step = lambda: model.predict_step(batch, {"data": pre}, {"data": post}, c["cfg"]["n_step_input"])["data"]  # noqa: E731
    for _ in range(2):
        want = step().clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = step()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_predict_step_batch_of_two_equals_the_samples_one_by_one(golden):
    """Batch > 1 through predict_step (the fused model-edge kernels take batch 1 only; a batch goes through the generic
    assembly + the normaliser's own kernel): the samples of a batch do not interact, so the batch's output must be the
    per-sample outputs stacked - each of which is pinned to the REFERENCE's predict_step by the test above."""
    c = golden("edges.pt")["predict_step"]
    model, pre, post = _model_and_processors(c)
    b0 = c["batch"].to(DEV)
    b1 = (b0 * 0.75 + 0.1 * b0.roll(1, dims=2)).contiguous()  # a second, different raw sample (same variable ranges)
    n = c["cfg"]["n_step_input"]
    one = [model.predict_step({"data": b}, {"data": pre}, {"data": post}, n)["data"] for b in (b0, b1)]
    two = model.predict_step({"data": torch.cat([b0, b1], 0)}, {"data": pre}, {"data": post}, n)["data"]
    assert two.shape[0] == 2 and two.shape[1:] == one[0].shape[1:]
    scale = max(1.0, float(one[0].abs().max()))
    for i in range(2):
        assert float((two[i:i + 1] - one[i]).abs().max()) <= 2e-5 * scale, i
    assert float((one[0].cpu() - c["out"]).abs().max()) <= 2e-5 * scale
