"""A vLLM-compatible serving load generator on the MI355X kernels (BASELINE.json config 4).

``Engine`` (continuous batching over a paged KV cache, hipGraph decode), ``EngineMetrics``
(vLLM's Prometheus names + exact ground truth), ``LlmServer`` (``/metrics``, ``/v1/completions``),
``OpenLoopLoad`` (Poisson arrivals) and ``router`` (one engine process per GPU behind a router). Imports torch: the monitor service itself never imports
this package.
"""
from .engine import BlockAllocator, Engine, EngineConfig, Request  # noqa: F401
from .metrics import EngineMetrics, Histogram, exact_quantile  # noqa: F401
