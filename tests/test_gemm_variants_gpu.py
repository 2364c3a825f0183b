"""The GEMM entry point picks one of several kernels per call (320x256 big tile, 256/192/64x128 persistent ring with or
without the ping-pong wave schedule, the 8-wave split-K 64x128 tile for small M, 128^2 register-staged, generic).  The linear tests see whatever the cost model
picks for their shapes; here the same tests are repeated in a child process with the developer overrides that force
every ring-eligible shape through ONE variant, so that each kernel also meets edge tiles, tiny inputs and all epilogues."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


VARIANTS = {"bigtile-everywhere": {"ANEMOI_GEMM_BIG": "1"}, "ring-lockstep": {"ANEMOI_GEMM_BIG": "0", "ANEMOI_GEMM_PP": "0"},
            "ring-pingpong": {"ANEMOI_GEMM_BIG": "0", "ANEMOI_GEMM_PP": "1"}, "small-m-two-wave-tiles": {"ANEMOI_GEMM_SPLITWAVE": "0"}}


def test_linear_suite_with_forced_kernel_variants():
    """The four child processes run side by side on the one GPU (each is mostly interpreter start-up and small launches)."""
    cmd = [sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_kernels_gpu.py"), "-x", "-q", "-k", "linear", "-p", "no:cacheprovider"]
    procs = {name: subprocess.Popen(cmd, cwd=REPO, env={**os.environ, **env}, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for name, env in VARIANTS.items()}
    failed = {}
    for name, pr in procs.items():
        try:
            out, _ = pr.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            pr.kill()
            out = "timed out"
        if pr.returncode != 0:
            failed[name] = out[-3000:]
    assert not failed, failed
