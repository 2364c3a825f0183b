import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The oracle's CPU forwards are fastest at a few dozen threads (bench.py's sweep picks 32-64 of the GPU box's 256 logical CPUs); torch's
# default - every logical CPU - makes the full-size parity tests slower, not faster.  ANEMOI_TEST_THREADS overrides.
torch.set_num_threads(max(1, min(int(os.environ.get("ANEMOI_TEST_THREADS", "48")), os.cpu_count() or 1)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present and -m gpu was not requested explicitly.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
