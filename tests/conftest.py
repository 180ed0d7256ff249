"""Shared fixtures. GPU tests carry ``@pytest.mark.gpu`` and are deselected on CPU-only hosts
with ``-m "not gpu"``; everything else runs without a GPU, a cluster or Prometheus.

The reference has no tests; the tiers here follow that section's plan.
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / on the GPU box)")


class StepClock:
    """A clock tests advance by hand."""

    def __init__(self, t: float = 1000.0):
        self.t = t

    def __call__(self) -> float:
        return self.t

    def advance(self, dt: float) -> float:
        self.t += dt
        return self.t


@pytest.fixture
def clock():
    return StepClock()


@pytest.fixture
def demo_cfg():
    from k8s_llm_monitor_amd.config import demo_config
    return demo_config()
