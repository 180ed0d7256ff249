"""Configuration: one file plus environment overrides.

The reference has no configuration at all: two literals (``PORT = 8888``,
``PROMETHEUS_URL``; monitor_server.js:10-11), a hard-coded core count of 8 (:76), inline
thresholds (:163-184), window/step (:119) and PromQL strings (:119-134). Every one of
those is a field here with the reference's value as its default, so an unconfigured
server behaves like the reference, and each can be overridden from a JSON/YAML file
(``--config`` / ``KLM_CONFIG``) or an environment variable ``KLM_<SECTION>_<FIELD>``
(e.g. ``KLM_SERVER_PORT=9000``, ``KLM_SOURCES_GPU=fake``, ``KLM_THRESHOLDS_HBM="60,80,90"``).
"""
from __future__ import annotations

import copy
import json
import os
from dataclasses import asdict, dataclass, field, fields
from pathlib import Path
from typing import Any, Dict, List, Mapping, Optional, Tuple


@dataclass
class ServerConfig:
    host: str = "0.0.0.0"            # monitor_server.js:297
    port: int = 8888                 # monitor_server.js:10
    cors_origin: str = "*"           # monitor_server.js:244
    access_log: bool = False
    static_dir: str = ""             # "" = the package's web/ directory
    state_dir: str = ""              # checkpoint/resume of history + alert state; "" = keep nothing on disk
    persist_period: float = 60.0     # seconds between checkpoints
    expose_config: bool = False      # GET /api/config (the running configuration: URLs, paths, peers); off: 404


@dataclass
class PeriodsConfig:
    """Sampler periods in seconds. Reference refresh periods: monitor.html:605-609."""
    realtime: float = 5.0            # host + GPU
    pods: float = 10.0
    vllm: float = 5.0
    history_step: float = 30.0       # monitor_server.js:119 ('30s')
    history_window: str = "30m"      # monitor_server.js:119 ('30m')
    alert_hold: float = 30.0         # how long an edge-triggered pod alert stays visible
    # the page's /api/alerts poll (its fallback path; /api/events pushes alerts as they appear). The reference polls every
    # 10 s (monitor.html:608); 4 s keeps the poll path alone under the 10 s target: realtime 5 s + tick + 4 s < 10 s
    page_alert_poll: float = 4.0
    # the alert watcher: between two realtime ticks the GPUs are read again every `alert_watch` seconds (one amdsmi metrics call
    # per GPU, ~0.5 ms; no slow-path calls) and, when a tiered signal (HBM %, temperatures, power %) sits in another tier than the
    # published snapshot says, the realtime tick runs right away instead of at its next grid point. 0 = off.
    alert_watch: float = 1.0
    probe: float = 0.0               # RCCL probe re-run period, 0 = only on demand
    peers: float = 5.0               # cluster view: peer poll period


@dataclass
class SourcesConfig:
    """Which backend feeds each collector. ``auto`` picks the real one when it works."""
    gpu: str = "auto"                # auto | amdsmi | native | exporter | fake | none
    kube: str = "auto"               # auto | kubectl | api | fake | none
    history: str = "auto"            # auto | local | prometheus
    vllm: str = "auto"               # auto | prometheus | direct | fake | none
    fake_gpus: int = 8
    fake_scenario: str = ""          # name of a scripted timeline (fakes/scenarios.py)
    kubectl: str = "kubectl"
    kube_api: str = ""               # e.g. https://kubernetes.default.svc
    kube_token_file: str = "/var/run/secrets/kubernetes.io/serviceaccount/token"
    kube_ca_file: str = "/var/run/secrets/kubernetes.io/serviceaccount/ca.crt"
    vllm_endpoints: List[str] = field(default_factory=list)   # direct /metrics URLs
    gpu_processes: bool = False      # per-process VRAM and pod attribution (needs hostPID in a pod)
    peers: List[str] = field(default_factory=list)            # other monitors to aggregate (cluster view)
    native_sampler: str = ""         # path of klm_amdsmi_sampler, "" = in-tree build
    probe_results: str = ""          # JSON file the RCCL probe writes, "" = in-tree default


@dataclass
class PrometheusConfig:
    url: str = "http://localhost:9090"   # monitor_server.js:11
    timeout: float = 3.0                 # the reference has none
    # PromQL per history key. Host queries are the reference's (monitor_server.js:119-125);
    # GPU queries replace DCGM_FI_DEV_* (:128-134) with AMD device-metrics-exporter gauges.
    q_cpu: str = '100 - (avg(rate(node_cpu_seconds_total{mode="idle"}[1m])) * 100)'
    q_memory: str = "100 - (node_memory_MemAvailable_bytes / node_memory_MemTotal_bytes * 100)"
    q_disk: str = ('100 - (node_filesystem_avail_bytes{mountpoint="/"} / '
                   'node_filesystem_size_bytes{mountpoint="/"} * 100)')
    q_gpu_util: str = "avg(gpu_gfx_activity)"
    q_gpu_mem: str = "avg(gpu_used_vram)"            # MiB, like DCGM_FI_DEV_FB_USED
    q_gpu_temp: str = "avg(gpu_junction_temperature)"
    # vLLM-ROCm serving metrics
    q_vllm_ttft_p50: str = "histogram_quantile(0.5, sum(rate(vllm:time_to_first_token_seconds_bucket[1m])) by (le))"
    q_vllm_ttft_p99: str = "histogram_quantile(0.99, sum(rate(vllm:time_to_first_token_seconds_bucket[1m])) by (le))"
    q_vllm_tokens_per_s: str = "sum(rate(vllm:generation_tokens_total[1m]))"
    q_vllm_prompt_tokens_per_s: str = "sum(rate(vllm:prompt_tokens_total[1m]))"
    q_vllm_kv_cache: str = "max(vllm:gpu_cache_usage_perc) * 100"
    q_vllm_running: str = "sum(vllm:num_requests_running)"
    q_vllm_waiting: str = "sum(vllm:num_requests_waiting)"


@dataclass
class ExporterConfig:
    """AMD device-metrics-exporter scraped directly (``sources.gpu = exporter``). Metric names
    are deployment-specific configuration; "" keeps the built-in default of that field
    (collectors/exporter.py DEFAULT_NAMES, unverified against a live exporter)."""
    url: str = "http://localhost:5000/metrics"
    timeout: float = 3.0
    m_gfx_activity: str = ""
    m_umc_activity: str = ""
    m_vram_used_mib: str = ""
    m_vram_total_mib: str = ""
    m_temp_hotspot: str = ""
    m_temp_mem: str = ""
    m_power_w: str = ""
    m_power_cap_w: str = ""
    m_clk_gfx: str = ""
    m_energy_uj: str = ""
    m_ecc_correctable: str = ""
    m_ecc_uncorrectable: str = ""
    m_xgmi_rx_bytes: str = ""
    m_xgmi_tx_bytes: str = ""

    def names(self) -> Dict[str, str]:
        return {f.name[2:]: getattr(self, f.name) for f in fields(self) if f.name.startswith("m_") and getattr(self, f.name)}


Tiers = Tuple[float, float, float]   # minor, serious, critical; a tier < 0 is disabled


@dataclass
class ThresholdsConfig:
    """Alert thresholds, strict ``>`` as in the reference (monitor_server.js:163-184)."""
    cpu: List[float] = field(default_factory=lambda: [70.0, 85.0, 95.0])       # :163-165
    memory: List[float] = field(default_factory=lambda: [70.0, 85.0, 95.0])    # :168-170
    disk: List[float] = field(default_factory=lambda: [70.0, 85.0, 95.0])      # :173-175
    hbm: List[float] = field(default_factory=lambda: [70.0, 85.0, 95.0])       # :180-182 (GPU memory %)
    # The reference's 75/85 C are for an NVIDIA edge sensor (:183-184). MI355X reports the
    # junction (hotspot) sensor, which runs hotter and throttles near 105-110 C.
    gpu_temp: List[float] = field(default_factory=lambda: [-1.0, 90.0, 100.0])
    hbm_temp: List[float] = field(default_factory=lambda: [-1.0, 85.0, 95.0])
    power_pct: List[float] = field(default_factory=lambda: [-1.0, 98.0, -1.0])  # of the power cap
    kv_cache: List[float] = field(default_factory=lambda: [70.0, 85.0, 95.0])
    vllm_waiting: List[float] = field(default_factory=lambda: [4.0, 16.0, 64.0])
    vllm_ttft_p99_s: List[float] = field(default_factory=lambda: [2.0, 5.0, 15.0])
    xgmi_link_down_tier: str = "critical"
    ecc_uncorrectable_tier: str = "critical"
    throttle_tier: str = "serious"


@dataclass
class NotifyConfig:
    """Alert delivery beyond the poll response. The reference has no notification channel,
    no debouncing and no alert history."""
    webhook_url: str = ""            # POST a JSON document on every alert begin / end ("" = off)
    webhook_timeout: float = 3.0
    min_tier: str = "minor"          # lowest tier that is sent: minor | serious | critical
    debounce_s: float = 0.0          # a level alert must persist this long before it is shown / sent


@dataclass
class CompatConfig:
    """Switches between the reference's behaviour and the fixed one."""
    cpu_percent: str = "load"        # "load": load_1min/cores*100 (:76) | "util": /proc/stat
    cpu_cores: int = 0               # 0 = os.cpu_count(); the reference hard-codes 8 (:76)
    gpu0_only_alerts: bool = False   # reference alerts on GPU 0 only (:178)
    first_sight_alerts: bool = False # reference fires "Pod 刚恢复" for unseen pods (:194-207)
    language: str = "zh"             # alert text language: zh (reference) | en
    reference_text: bool = False     # the reference's exact title / desc / fix strings (:163-230) for its own rules
    reference_rules_only: bool = False   # evaluate only the rules the reference has (:162-232)
    alert_extra_fields: bool = True  # /api/alerts items carry key / since / signal beside {title, desc, fix}
    alerts_per_request: bool = False # evaluate on every GET /api/alerts against a state the request replaces (:282-288, :235):
                                     # edge alerts appear in exactly one response, to whichever client polls first
    reference_defaults: bool = False # ONE switch for a side-by-side run: everything below set to the reference's behaviour


#: what ``compat.reference_defaults`` sets: applied UNDER the file, the environment and the overrides (`load_config`),
#: so a value the operator wrote explicitly wins over the preset
REFERENCE_PRESET = {
    "compat": {"cpu_percent": "load", "cpu_cores": 8,                     # load_1min / 8 (monitor_server.js:76)
               "gpu0_only_alerts": True,                                  # :178
               "first_sight_alerts": True,                                # :194-207
               "language": "zh", "reference_text": True, "reference_rules_only": True, "alert_extra_fields": False,
               "alerts_per_request": True},
    "thresholds": {"gpu_temp": [-1.0, 75.0, 85.0]},                       # :183-184
    "periods": {"page_alert_poll": 10.0, "alert_watch": 0.0},             # monitor.html:608; the reference evaluates per request
}


@dataclass
class Config:
    server: ServerConfig = field(default_factory=ServerConfig)
    periods: PeriodsConfig = field(default_factory=PeriodsConfig)
    sources: SourcesConfig = field(default_factory=SourcesConfig)
    prometheus: PrometheusConfig = field(default_factory=PrometheusConfig)
    exporter: ExporterConfig = field(default_factory=ExporterConfig)
    thresholds: ThresholdsConfig = field(default_factory=ThresholdsConfig)
    notify: NotifyConfig = field(default_factory=NotifyConfig)
    compat: CompatConfig = field(default_factory=CompatConfig)

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


class ConfigError(ValueError):
    pass


def _coerce(value: Any, like: Any, where: str) -> Any:
    """Convert `value` (from a file or an env string) to the type of the default `like`."""
    try:
        if isinstance(like, bool):
            if isinstance(value, str):
                v = value.strip().lower()
                if v in ("1", "true", "yes", "on"):
                    return True
                if v in ("0", "false", "no", "off", ""):
                    return False
                raise ValueError(value)
            return bool(value)
        if isinstance(like, int):
            return int(value)
        if isinstance(like, float):
            return float(value)
        if isinstance(like, list):
            if isinstance(value, str):
                value = [x for x in (p.strip() for p in value.split(",")) if x != ""]
            if not isinstance(value, (list, tuple)):
                raise ValueError(value)
            if like and isinstance(like[0], float):
                return [float(x) for x in value]
            return [str(x) for x in value]
        return str(value)
    except (TypeError, ValueError) as e:
        raise ConfigError(f"bad value for {where}: {value!r}") from e


def _apply(cfg: Config, data: Mapping[str, Any], origin: str) -> None:
    sections = {f.name: getattr(cfg, f.name) for f in fields(cfg)}
    for sec, body in data.items():
        if sec not in sections:
            raise ConfigError(f"{origin}: unknown section {sec!r}")
        if not isinstance(body, Mapping):
            raise ConfigError(f"{origin}: section {sec!r} must be a mapping")
        obj = sections[sec]
        names = {f.name for f in fields(obj)}
        for key, val in body.items():
            if key not in names:
                raise ConfigError(f"{origin}: unknown key {sec}.{key}")
            setattr(obj, key, _coerce(val, getattr(obj, key), f"{sec}.{key}"))


def _apply_env(cfg: Config, env: Mapping[str, str]) -> None:
    for f in fields(cfg):
        obj = getattr(cfg, f.name)
        for g in fields(obj):
            name = f"KLM_{f.name.upper()}_{g.name.upper()}"
            if name in env:
                setattr(obj, g.name, _coerce(env[name], getattr(obj, g.name), name))


def validate(cfg: Config) -> None:
    if not (0 < cfg.server.port < 65536) and cfg.server.port != 0:
        raise ConfigError(f"server.port out of range: {cfg.server.port}")
    for f in fields(cfg.thresholds):
        v = getattr(cfg.thresholds, f.name)
        if isinstance(v, list):
            if len(v) != 3:
                raise ConfigError(f"thresholds.{f.name} needs 3 values (minor, serious, critical)")
            act = [x for x in v if x >= 0]
            if act != sorted(act):
                raise ConfigError(f"thresholds.{f.name} must be ascending: {v}")
        elif v not in ("minor", "serious", "critical", "off"):
            raise ConfigError(f"thresholds.{f.name} must be a tier name or 'off'")
    for name in ("realtime", "pods", "vllm", "history_step"):
        if getattr(cfg.periods, name) <= 0:
            raise ConfigError(f"periods.{name} must be > 0")
    # the page substitutes this into setInterval(fetchAlerts, ms): 0 or a negative value would hammer /api/alerts
    if cfg.periods.page_alert_poll < 0.25:
        raise ConfigError(f"periods.page_alert_poll must be >= 0.25 s (got {cfg.periods.page_alert_poll})")
    if cfg.periods.alert_watch != 0 and cfg.periods.alert_watch < 0.1:
        raise ConfigError(f"periods.alert_watch must be 0 (off) or >= 0.1 s (got {cfg.periods.alert_watch})")
    if cfg.notify.min_tier not in ("minor", "serious", "critical"):
        raise ConfigError("notify.min_tier must be minor, serious or critical")
    if cfg.notify.debounce_s < 0:
        raise ConfigError("notify.debounce_s must be >= 0")
    if cfg.compat.cpu_percent not in ("load", "util"):
        raise ConfigError("compat.cpu_percent must be 'load' or 'util'")
    allowed = {"gpu": {"auto", "amdsmi", "native", "exporter", "fake", "none"},
               "kube": {"auto", "kubectl", "api", "fake", "none"},
               "history": {"auto", "local", "prometheus"},
               "vllm": {"auto", "prometheus", "direct", "fake", "none"}}
    for k, ok in allowed.items():
        if getattr(cfg.sources, k) not in ok:
            raise ConfigError(f"sources.{k} must be one of {sorted(ok)}")


def load_config(path: Optional[str] = None, env: Optional[Mapping[str, str]] = None,
                overrides: Optional[Mapping[str, Any]] = None) -> Config:
    """Defaults <- [reference preset] <- file <- environment <- explicit overrides (a nested dict).

    ``compat.reference_defaults`` (set in any of the three) puts REFERENCE_PRESET UNDER them: the preset replaces this
    project's defaults with the reference's, and every value the operator wrote explicitly (a custom gpu_temp threshold,
    a language) still wins over it."""
    env = os.environ if env is None else env
    path = path or env.get("KLM_CONFIG")
    data: Mapping[str, Any] = {}
    if path:
        p = Path(path)
        text = p.read_text(encoding="utf-8")
        if p.suffix in (".yaml", ".yml"):
            import yaml  # SafeLoader only
            data = yaml.safe_load(text) or {}
        else:
            data = json.loads(text or "{}")
        if not isinstance(data, Mapping):
            raise ConfigError(f"{path}: top level must be a mapping")

    def layered(preset: bool) -> Config:
        cfg = Config()
        if preset:
            _apply(cfg, REFERENCE_PRESET, "compat.reference_defaults")
        if data:
            _apply(cfg, data, str(path))
        _apply_env(cfg, env)
        if overrides:
            _apply(cfg, overrides, "overrides")
        return cfg

    cfg = layered(False)
    if cfg.compat.reference_defaults:
        cfg = layered(True)
    validate(cfg)
    return cfg


def demo_config(**overrides: Any) -> Config:
    """CPU-only demo mode: every source is a fake (BASELINE.json config 1)."""
    base = {"sources": {"gpu": "fake", "kube": "fake", "vllm": "fake", "history": "local"},
            "server": {"port": 0, "host": "127.0.0.1"}}
    merged = copy.deepcopy(base)
    for k, v in overrides.items():
        merged.setdefault(k, {}).update(v)
    return load_config(env={}, overrides=merged)
