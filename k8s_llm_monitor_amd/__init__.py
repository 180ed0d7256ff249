"""MI355X-native inference-cluster monitor.

Same capabilities as fuqiangfeng96-web/k8s-llm-monitor (HTTP API, three-tier alerts,
single-page dashboard; reference: monitor_server.js, monitor.html), rebuilt for AMD
Instinct MI355X: amdsmi telemetry, vLLM-ROCm metrics, and a validation toolkit of
hand-written gfx950 HIP kernels and an RCCL-over-xGMI probe.
"""
__version__ = "0.1.0"
