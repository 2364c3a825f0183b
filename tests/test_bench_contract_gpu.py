"""The benchmark line's contract (driver prompt, section 4 "Measurement"): `python bench.py` prints ONE JSON line whose keys, units
and consistency relations are what the driver and the judge read.  Run here with few steps and a small CPU budget."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_contract():
    env = dict(os.environ, ANEMOI_BENCH_CPU_BUDGET_S="15")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ANEMOI_BENCH_TRANSPORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("forward nodes*channels/sec on O96 GraphTransformer") and d["unit"] == "nodes*channels/s"
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["dtype"] == "bf16" and d["vs_baseline"] is None and d["scaling"] in ("weak", "strong") and "synthetic" in d["data"]
    assert abs(d["value"] - 40320 * 512 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]  # value = N_data * channels / t_forward
    cfg = d["config"]
    assert "workload" in cfg and "O96" in cfg["workload"] and "16 layers x 512 ch" in cfg["workload"] and "model" not in cfg
    assert cfg["hip_graph"] is True and cfg["graph_equals_eager"] is True and cfg["graph_segments"] == 1
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] in (8000.0, 2500.0)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert rf["frac"] <= rf["frac_overhead_subtracted"]  # the raw brackets are the lower bound
    assert rf["traffic"] is None or rf["traffic"] > 0.5 * rf["algorithmic_bytes_per_launch"]
    # the dominant family's time is part of the step.  `avg_launch_us` is the RAW bracket (kernel + the gaps to its two markers, which
    # grow when the box's host cannot keep the launch queue backed up): compared after the calibrated marker overhead is taken off, and
    # with room for a slow host - the claim checked is "a part of the step", not a timing
    net_us = rf["calls_per_step"] * max(rf["avg_launch_us"] - rf["bracket_overhead_us"], 0.0)
    assert net_us <= 1.25 * d["ms_per_step"] * 1e3, (net_us, d["ms_per_step"], rf)
    gs = rf["gather_scatter"]
    assert gs["bound"] == "hbm" and 0.05 < gs["frac"] < 1.0
    cpu = d["cpu_baseline"]
    assert cpu["unit"] == d["unit"] and cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu
    assert cpu["timed_forwards"] >= 3 and cpu["physical_cores"] >= 1
    assert d["value"] > 50 * cpu["value"]  # a GPU forward in milliseconds against seconds on the host
    assert d["speedup_vs_gpu_eager"] > 2.0
