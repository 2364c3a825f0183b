#!/usr/bin/env python
"""Reference-side drop-in probe (BUILD CONTAINER ONLY: needs /root/reference; run by tests/test_dropin_reference_cpu.py in a subprocess,
because the stand-ins of the reference's un-vendored dependencies are injected into sys.modules).

The REFERENCE's unchanged ``AnemoiModelEncProcDec`` (models/src/anemoi/models/models/encoder_processor_decoder.py:51-96, 185-330) is
instantiated twice from the same config - once on its own layers, once with the three ``_target_`` entries pointing at
``anemoi_core_amd.layers.*`` (GraphTransformer and GNN) - and must then
  * construct,
  * have the same state_dict keys and shapes,
  * load the reference's weights with ``strict=True``,
  * run the reference's own ``forward`` through this package's mapper / processor signatures down to this package's first HIP op, which
    refuses CPU tensors ("no CPU fallback") from anemoi_core_amd/ops.py - i.e. every call in between bound.
"""
import copy
import sys
import traceback

import torch

import make_golden as mg  # installs the stand-ins, imports the reference
from anemoi.models.models import AnemoiModelEncProcDec

g = mg.build_synthetic_graph("o8", 3)
fails = 0
for kind in ("gt", "gnn"):
    torch.manual_seed(5)
    cfg_ref = mg.model_config(kind, 64, 2, 4, 8)
    cfg_amd = copy.deepcopy(cfg_ref)
    for part in ("encoder", "processor", "decoder"):
        tgt = cfg_amd["model"][part]["_target_"]
        assert tgt.startswith("anemoi.models.layers."), tgt
        cfg_amd["model"][part]["_target_"] = tgt.replace("anemoi.models.layers.", "anemoi_core_amd.layers.")
        if "graph_attention_backend" in cfg_amd["model"][part]:
            cfg_amd["model"][part]["graph_attention_backend"] = "hip"
    kw = dict(data_indices=mg.make_data_indices(4, 4), statistics={"data": None}, n_step_input=2, n_step_output=1, graph_data=mg.make_hetero(g))
    ref = AnemoiModelEncProcDec(model_config=cfg_ref, **kw).eval()
    amd = AnemoiModelEncProcDec(model_config=mg.rs.DotDict(cfg_amd), **kw).eval()
    for part in ("encoder", "processor", "decoder"):
        mod = type(getattr(amd, part)["data"] if hasattr(getattr(amd, part), "keys") else getattr(amd, part)).__module__
        assert mod.startswith("anemoi_core_amd.layers"), f"{kind} {part}: built {mod}"
    sr, sa = ref.state_dict(), amd.state_dict()
    assert list(sr.keys()) == list(sa.keys()), f"{kind}: state_dict keys differ: {sorted(set(sr) ^ set(sa))[:6]}"
    bad = [k for k in sr if tuple(sr[k].shape) != tuple(sa[k].shape)]
    assert not bad, f"{kind}: shapes differ for {bad[:6]}"
    amd.load_state_dict(sr, strict=True)
    x = torch.randn(1, 2, 1, g.num_data, 4)
    try:
        with torch.no_grad():
            amd({"data": x})
        print(f"{kind}: FAIL - the forward ran on CPU tensors (there must be no CPU fallback)")
        fails += 1
    except RuntimeError as e:
        tb = traceback.extract_tb(sys.exc_info()[2])
        files = [f.filename for f in tb]
        assert "no CPU fallback" in str(e), f"{kind}: unexpected error {e}"
        own = [f for f in files if "/dist-packages/torch/" not in f and "/site-packages/torch/" not in f]  # (the refusal may come from the TORCH_LIBRARY layer under ops.py)
        assert own[-1].endswith("anemoi_core_amd/ops.py"), f"{kind}: raised below {own[-1]}"
        ref_frames = [f for f in files if "/reference/" in f and f.endswith("encoder_processor_decoder.py")]
        assert ref_frames, f"{kind}: the reference's forward is not on the stack"
        last = [f for f in tb if f.filename.endswith("anemoi_core_amd/ops.py")][-1]
        print(f"{kind}: OK - {len(sr)} state_dict entries equal, strict load, reference forward -> ops.{last.name} refuses the CPU tensor")
sys.exit(1 if fails else 0)
