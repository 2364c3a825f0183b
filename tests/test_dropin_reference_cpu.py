"""The reference's own ``AnemoiModelEncProcDec`` on this package's layers (north_star: "so AnemoiModelEncProcDec is a drop-in",
models/src/anemoi/models/models/encoder_processor_decoder.py:51-96, 185-330): tests/golden/dropin_probe.py, run in a subprocess (it injects
stand-ins for the reference's un-vendored dependencies into sys.modules).  Build container only: skipped where /root/reference is absent."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/models/src/anemoi"), reason="needs the reference checkout (build container only)")
def test_reference_model_class_runs_on_this_packages_layers():
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "dropin_probe.py")], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "OMP_NUM_THREADS": "4"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "gt: OK" in r.stdout and "gnn: OK" in r.stdout, r.stdout
