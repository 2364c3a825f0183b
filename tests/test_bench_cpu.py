"""Host-side checks of bench.py's command line (no GPU needed)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    repo = REPO
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=repo)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_config_presets(monkeypatch):
    sys.path.insert(0, REPO)
    import bench

    for name, want in (("o96", ("o96", 5, "gt")), ("o96-res6", ("o96", 6, "gt")), ("n320", ("n320", 6, "gt")), ("gnn", ("o96", 5, "gnn"))):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--config", name])
        a = bench.parse()
        assert (a.data_grid, a.hidden_res, a.kind) == want and a.gpus == 1 and a.layers == 16 and a.channels == 512
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "n320", "--hidden-res", "5"])  # explicit flags win
    assert bench.parse().hidden_res == 5
