"""Validation workloads ("models") that run on the hand-written kernels."""
from .linear_step import LinearLayerStep  # noqa: F401
