"""Module path mirror of the reference's ``anemoi.models.layers.normalization``."""
from .kernels import AutocastLayerNorm, ConditionalLayerNorm, LayerNorm  # noqa: F401
