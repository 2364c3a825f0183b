"""Module path mirror of the reference's ``anemoi.models.layers.normalization``."""
from .kernels import AutocastLayerNorm, LayerNorm  # noqa: F401
