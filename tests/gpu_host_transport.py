"""Kept for the tests' imports: the host-staged debug transport lives in the package (anemoi_core_amd/distributed/host_transport.py)."""
from anemoi_core_amd.distributed.host_transport import install  # noqa: F401
