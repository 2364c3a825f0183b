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


@pytest.mark.parametrize("env", [{"ANEMOI_GEMM_BIG": "1"}, {"ANEMOI_GEMM_BIG": "0", "ANEMOI_GEMM_PP": "0"}, {"ANEMOI_GEMM_BIG": "0", "ANEMOI_GEMM_PP": "1"},
                                 {"ANEMOI_GEMM_SPLITWAVE": "0"}],
                         ids=["bigtile-everywhere", "ring-lockstep", "ring-pingpong", "small-m-two-wave-tiles"])
def test_linear_suite_with_forced_kernel_variant(env):
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_kernels_gpu.py"), "-x", "-q", "-k", "linear",
                        "-p", "no:cacheprovider"], cwd=REPO, env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
