"""The sampling loop: every source is polled on its own period by a background thread, the
alert engine runs once per tick, HTTP handlers only read snapshots.

Reference behaviour replaced: collectors run synchronously inside each request
(monitor_server.js:273-288), so every open browser re-runs ``df``, ``nvidia-smi`` and
``kubectl`` and ``/api/alerts`` mutates ``lastPodStates`` per client (SURVEY.md 2.7 item 2,
2.8 "blocking collectors" / "duplicate work"). Here cost is independent of viewer count,
a slow ``kubectl`` cannot stall GPU sampling (one thread per source) and the alert state
has one owner.
"""
from __future__ import annotations

import threading
import time
from typing import Any, Callable, Dict, List, Optional

from ..collectors import history as hist
from ..collectors import procmap
from ..collectors.gpu import GpuSource, open_gpu_source
from ..collectors.host import HostCollector
from ..collectors.kube import KubeSource, open_kube_source
from ..collectors.prometheus import PrometheusClient
from ..collectors.vllm import VllmSource, open_vllm_source
from ..config import Config
from ..utils.ringbuffer import HistoryStore
from ..utils.timefmt import parse_duration
from .alerts import THERMAL_REASONS, AlertBook, AlertState, check_alerts, empty_alerts, tier_of


class PeriodicTask:
    """Calls ``fn`` every ``period`` seconds on its own daemon thread, on a fixed time grid
    (no drift: a slow call skips grid points instead of shifting them). Keeps timing stats
    of the calls, including the thread's CPU time, which is what "sampling overhead" on the
    host side means."""

    def __init__(self, name: str, fn: Callable[[], Any], period: float):
        self.name, self.fn, self.period = name, fn, float(period)
        self._stop = threading.Event()
        self._th: Optional[threading.Thread] = None
        self.calls = 0
        self.errors = 0
        self.total_s = 0.0
        self.max_s = 0.0
        self.last_s = 0.0
        self.cpu_s = 0.0
        self.last_error: Optional[str] = None
        self.last_run_wall = 0.0

    def run_once(self) -> None:
        t0, c0 = time.perf_counter(), time.thread_time()
        try:
            self.fn()
        except Exception as e:      # a collector bug must not kill the loop
            self.errors += 1
            self.last_error = f"{type(e).__name__}: {e}"
        dt = time.perf_counter() - t0
        self.cpu_s += time.thread_time() - c0
        self.calls += 1
        self.total_s += dt
        self.last_s = dt
        self.max_s = max(self.max_s, dt)
        self.last_run_wall = time.time()

    def _loop(self) -> None:
        nxt = time.monotonic()
        while not self._stop.is_set():
            self.run_once()
            nxt += self.period
            now = time.monotonic()
            if nxt <= now:                               # overran: realign to the grid
                nxt = now + self.period - ((now - nxt) % self.period)
            self._stop.wait(max(0.0, nxt - now))

    def start(self) -> "PeriodicTask":
        if self._th is None:
            self._th = threading.Thread(target=self._loop, name=f"klm-{self.name}", daemon=True)
            self._th.start()
        return self

    def stop(self, timeout: float = 2.0) -> None:
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=timeout)
            self._th = None

    def stats(self) -> Dict[str, Any]:
        n = max(1, self.calls)
        return {"period_s": self.period, "calls": self.calls, "errors": self.errors,
                "mean_ms": round(self.total_s / n * 1e3, 4), "max_ms": round(self.max_s * 1e3, 4),
                "last_ms": round(self.last_s * 1e3, 4), "cpu_ms_per_call": round(self.cpu_s / n * 1e3, 4),
                "duty_percent": round(self.total_s / n / self.period * 100.0, 5) if self.period > 0 else None,
                "last_error": self.last_error}


class BackgroundPoller:
    """One GPU source polled in the background; used by ``bench.py`` to run the monitor's
    real telemetry path next to the load and by the validation phases to read it."""

    def __init__(self, src: GpuSource, period: float = 5.0):
        self.src = src
        self.latest: List[Dict[str, Any]] = []
        self.series: List[Dict[str, Any]] = []     # (t, per-GPU samples) kept for validation
        self.keep = 4096
        self._lock = threading.Lock()
        self.task = PeriodicTask("gpu-poller", self._tick, period)

    def _tick(self) -> None:
        s = self.src.sample()
        with self._lock:
            self.latest = s
            self.series.append({"t": time.time(), "gpus": s})
            if len(self.series) > self.keep:
                del self.series[: len(self.series) - self.keep]

    def start(self) -> "BackgroundPoller":
        self.task.start()
        return self

    def stop(self) -> None:
        self.task.stop()

    def snapshot(self) -> List[Dict[str, Any]]:
        with self._lock:
            return list(self.latest)

    def window(self, since: float) -> List[Dict[str, Any]]:
        with self._lock:
            return [s for s in self.series if s["t"] >= since]

    def describe(self) -> str:
        return f"{self.src.describe()} every {self.task.period:g}s"

    def stats(self) -> Dict[str, Any]:
        return self.task.stats()


class Monitor:
    """Application state: sources, periodic tasks, latest snapshot, history, alerts."""

    def __init__(self, cfg: Optional[Config] = None, *, host: Optional[HostCollector] = None,
                 gpu: Optional[GpuSource] = None, kube: Optional[KubeSource] = None,
                 vllm: Optional[VllmSource] = None, prom: Optional[PrometheusClient] = None,
                 clock: Callable[[], float] = time.time):
        self.cfg = cfg or Config()
        c = self.cfg
        self.clock = clock
        self.prom = prom or PrometheusClient(c.prometheus.url, c.prometheus.timeout)
        self.host = host or HostCollector(cores=c.compat.cpu_cores, cpu_percent=c.compat.cpu_percent)
        self.gpu = gpu or open_gpu_source(c.sources.gpu, c)
        self.kube = kube or open_kube_source(c.sources.kube, c)
        self.vllm = vllm or open_vllm_source(c.sources.vllm, c, self.prom)
        self.pod_resolver = procmap.PodResolver()

        window_s = parse_duration(c.periods.history_window)
        cap = int(window_s / max(0.05, min(c.periods.realtime, c.periods.history_step))) + 8
        self.store = HistoryStore(capacity=min(cap, 200000))
        self.local_history = hist.LocalHistory(self.store, c.periods.history_window, c.periods.history_step, now=clock)
        self.prom_history = hist.PrometheusHistory(self.prom, hist.prometheus_queries(c), c.periods.history_window,
                                                   f"{int(c.periods.history_step)}s")
        self._history_mode = c.sources.history
        self._prom_ok: Optional[bool] = None
        self._prom_checked = 0.0
        self._hist_cache = None
        self._hist_lock = threading.Lock()

        self._lock = threading.RLock()
        self._tick_lock = threading.Lock()
        from ..collectors.gpu import GpuNormalizer
        self._watch_norm = GpuNormalizer()
        self.watch_pokes = 0        # early realtime ticks the alert watcher asked for
        self._snap: Dict[str, Any] = {"host": {"cpu": {}, "memory": {}, "disk": {}}, "gpus": [], "pods": [],
                                      "vllm": {"available": False}, "alerts": empty_alerts(),
                                      "t_host": 0.0, "t_gpu": 0.0, "t_pods": 0.0, "t_vllm": 0.0, "t_alerts": 0.0}
        self._alert_state = AlertState()
        self._request_alert_state = AlertState()      # compat.alerts_per_request: the reference's lastPodStates (:157)
        self.version = 0            # bumped whenever the snapshot changes (HTTP body cache key)
        self.book = AlertBook(hold_s=c.periods.alert_hold, debounce_s=c.notify.debounce_s)
        self.notifier = None
        if c.notify.webhook_url:
            from .notify import WebhookNotifier
            self.notifier = WebhookNotifier(c.notify.webhook_url, c.notify.webhook_timeout, c.notify.min_tier)
            self.book.on_event = self.notifier
        self.listeners: List[Callable[[str, Any], None]] = []
        self.started = 0.0
        self.tasks: Dict[str, PeriodicTask] = {
            "realtime": PeriodicTask("realtime", self.tick_realtime, c.periods.realtime),
            "pods": PeriodicTask("pods", self.tick_pods, c.periods.pods),
            "vllm": PeriodicTask("vllm", self.tick_vllm, c.periods.vllm),
        }
        if (c.periods.alert_watch > 0 and c.periods.alert_watch < c.periods.realtime and not c.compat.alerts_per_request
                and getattr(self.gpu.reader, "supports_fast", False)):
            self.tasks["watch"] = PeriodicTask("watch", self.tick_watch, c.periods.alert_watch)
        from ..collectors.peers import PeerPoller
        self.peers = PeerPoller(c.sources.peers, timeout=min(2.0, c.periods.peers))
        if self.peers.peers:
            self.tasks["peers"] = PeriodicTask("peers", self.peers.poll, c.periods.peers)
        if c.periods.probe > 0:
            self.tasks["probe"] = PeriodicTask("probe", self.run_probe, c.periods.probe)
        self.state_dir = c.server.state_dir
        self.restored: Dict[str, Any] = {"history": False, "alert_state": False, "alert_book": False}
        if self.state_dir:
            self.tasks["persist"] = PeriodicTask("persist", self.checkpoint, max(1.0, c.server.persist_period))
            self.restore()

    # ---- checkpoint / resume (the reference keeps nothing: a restart clears lastPodStates and
    # re-fires its first-sight alerts, and history lives only in Prometheus)
    def _state_path(self, name: str) -> str:
        import os
        return os.path.join(self.state_dir, name)

    def checkpoint(self) -> None:
        import json
        import os
        if not self.state_dir:
            return
        os.makedirs(self.state_dir, exist_ok=True)
        self.store.save(self._state_path("history.json"))
        with self._lock:
            st = self._alert_state
            doc = {"saved": self.clock(), "pods": st.pods, "gpu_count": st.gpu_count,
                   "ecc_correctable": {str(k): v for k, v in st.ecc_correctable.items()},
                   "bad_pages_retired": {str(k): v for k, v in st.bad_pages_retired.items()},
                   "vllm_endpoints": st.vllm_endpoints, "evaluations": st.evaluations,
                   "book": self.book.to_state()}
        tmp = self._state_path(f"alert_state.json.tmp.{os.getpid()}")
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(doc, f)
        os.replace(tmp, self._state_path("alert_state.json"))

    def restore(self, max_age_s: float = 3600.0) -> None:
        import json
        self.restored["history"] = self.store.load(self._state_path("history.json"))
        try:
            with open(self._state_path("alert_state.json"), "r", encoding="utf-8") as f:
                d = json.load(f)
            if self.clock() - float(d.get("saved", 0.0)) <= max_age_s:       # a stale pod table would raise false edges
                self._alert_state = AlertState(pods={str(k): dict(v) for k, v in d["pods"].items()}, gpu_count=d.get("gpu_count"),
                                               ecc_correctable={int(k): int(v) for k, v in (d.get("ecc_correctable") or {}).items()},
                                               bad_pages_retired={int(k): int(v) for k, v in (d.get("bad_pages_retired") or {}).items()},
                                               vllm_endpoints=int(d.get("vllm_endpoints", 0)),
                                               evaluations=int(d.get("evaluations", 0)))
                self.restored["alert_state"] = True
                if isinstance(d.get("book"), dict):
                    # alerts that were live before the restart keep their `since`, log no second "begin" and post no second webhook
                    self.book.load_state(d["book"])
                    self.restored["alert_book"] = True
        except (OSError, ValueError, KeyError, TypeError, AttributeError):
            pass

    # ---- ticks -------------------------------------------------------------------------
    def tick_realtime(self) -> None:
        with self._tick_lock:                 # the realtime task and the alert watcher's early tick never run it concurrently
            self._tick_realtime()

    def _tick_realtime(self) -> None:
        host = self.host.collect()
        gpus = self.gpu.sample()
        if self.cfg.sources.gpu_processes and gpus:
            with self._lock:
                pods = self._snap["pods"]
            procmap.attribute(gpus, pods, self.pod_resolver)
        now = self.clock()
        with self._lock:
            self._snap["host"], self._snap["gpus"] = host, gpus
            self._snap["t_host"] = self._snap["t_gpu"] = now
            self.version += 1
            vllm = self._snap["vllm"]
        self.store.record(now, hist.history_values(host, gpus, vllm if vllm.get("available") else None))
        self._evaluate(now)
        self._emit("realtime", {"host": host, "gpus": gpus})

    # ---- the alert watcher ----------------------------------------------------------------
    def _tier_signature(self, gpus) -> Dict[str, Optional[str]]:
        """key -> tier of the tiered GPU signals (the rules of server/alerts.py that compare a value with three thresholds)."""
        th = self.cfg.thresholds
        sig: Dict[str, Optional[str]] = {}
        for g in gpus or []:
            k = f"gpu/{g.get('index', 0)}"
            used, total = g.get("memoryUsed"), g.get("memoryTotal")
            if isinstance(used, (int, float)) and isinstance(total, (int, float)) and total > 0:
                sig[k + "/hbm"] = tier_of(used / total * 100.0, th.hbm)
            if isinstance(g.get("temperature"), (int, float)):
                sig[k + "/temp"] = tier_of(float(g["temperature"]), th.gpu_temp)
            if isinstance(g.get("temperatureHbm"), (int, float)):
                sig[k + "/hbm_temp"] = tier_of(float(g["temperatureHbm"]), th.hbm_temp)
            pw, cap = g.get("powerW"), g.get("powerCapW")
            if isinstance(pw, (int, float)) and isinstance(cap, (int, float)) and cap > 0:
                sig[k + "/power"] = tier_of(pw / cap * 100.0, th.power_pct)
            # the level-triggered conditions that are a yes / no: power-cap throttling, thermal throttling, links down
            res = (g.get("throttle") or {}).get("residencyPercent") or {}
            sig[k + "/throttle_ppt"] = "on" if (isinstance(res.get("ppt"), (int, float)) and res["ppt"] > 50.0) else None
            sig[k + "/throttle_thm"] = "on" if any((res.get(r) or 0) > 0 for r in THERMAL_REASONS) else None
            down = sorted(str(l.get("link")) for l in (g.get("xgmi") or {}).get("links") or [] if l.get("state") == "down")
            sig[k + "/xgmi_down"] = ",".join(down) or None
        return sig

    def tick_watch(self) -> None:
        """Between two realtime ticks: one cheap read of the GPUs through the watcher's OWN normaliser (the published path's
        accumulator deltas are not touched); when a tiered signal sits in another tier than the published snapshot says, the
        realtime tick runs now. Threshold crossing -> alert is then bounded by `periods.alert_watch`, not `periods.realtime`."""
        reader = self.gpu.reader
        if not getattr(reader, "supports_fast", False):
            return
        try:
            quick = [self._watch_norm.normalize(r) for r in reader.read(fast=True)]
        except Exception:
            return                                      # the regular tick reports reader errors
        with self._lock:
            published = self._snap["gpus"]
            if not self._snap["t_gpu"]:
                return                                  # nothing published yet: the first regular tick is about to come
        if self._tier_signature(quick) != self._tier_signature(published):
            self.watch_pokes += 1
            self.tick_realtime()

    def tick_pods(self) -> None:
        pods = self.kube.pods()
        now = self.clock()
        with self._lock:
            self._snap["pods"], self._snap["t_pods"] = pods, now
            self.version += 1
        self._evaluate(now)
        self._emit("pods", pods)

    def tick_vllm(self) -> None:
        v = self.vllm.sample()
        now = self.clock()
        with self._lock:
            self._snap["vllm"], self._snap["t_vllm"] = v, now
            self.version += 1
        self._emit("vllm", v)

    def tick_all(self) -> None:
        """One synchronous pass over every source (tests, first paint before the threads run)."""
        self.tick_vllm()
        self.tick_pods()
        self.tick_realtime()

    def _evaluate(self, now: float) -> None:
        with self._lock:
            s = self._snap
            if not s["t_host"]:
                return                                       # nothing sampled yet
            pods_known = bool(s["t_pods"])
            # The first pods tick can take the kubectl / API-server timeout on a node whose API server is unreachable.
            # Host and GPU rules (HBM, temperature, ECC) do not wait for it: they are evaluated as soon as the realtime
            # sources have reported. Until the pod list is in, the pod half is left exactly as it was: the engine's pod
            # table is carried over unchanged, and the book keeps (neither ends nor re-announces) the pod alerts it
            # restored, instead of "ending" them on half a snapshot and announcing them again a moment later.
            # The same for the vLLM scrape: "an endpoint that answered before the restart no longer does" can only be
            # said once the scrape has run.
            vllm_known = bool(s["t_vllm"])
            st = self._alert_state
            if not vllm_known and st.vllm_endpoints:
                import dataclasses
                st = dataclasses.replace(st, vllm_endpoints=0)
            alerts, nxt = check_alerts(s["host"], s["gpus"], s["pods"] if pods_known else [], st, self.cfg,
                                       s["vllm"] if s["vllm"].get("available") else None)
            if not pods_known:
                nxt.pods = self._alert_state.pods
            if not vllm_known:
                nxt.vllm_endpoints = max(nxt.vllm_endpoints, self._alert_state.vllm_endpoints)
            self._alert_state = nxt
            shown = self.book.update(alerts, now, keep_prefixes=(() if pods_known else ("pod/",)) + (() if vllm_known else ("vllm/",)))
            before = {(a.get("key"), t) for t, xs in s["alerts"].items() for a in xs}
            s["alerts"], s["t_alerts"] = shown, now
            self.version += 1
            # pushed to /api/events when the visible set changed: a new alert, a tier change, or one that cleared
            changed = {(a.get("key"), t) for t, xs in shown.items() for a in xs} != before
        if changed:
            self._emit("alerts", shown)

    def _emit(self, kind: str, payload: Any) -> None:
        for fn in list(self.listeners):
            try:
                fn(kind, payload)
            except Exception:
                pass

    def run_probe(self) -> None:
        """Re-run the native RCCL probe (``periods.probe`` > 0) so the dashboard's curve stays fresh."""
        from ..parallel import probe
        probe.run_sweep(path=self.cfg.sources.probe_results, max_bytes=1 << 28, iters=10, warmup=3)

    def cluster(self) -> Dict[str, Any]:
        s = self.snapshot()
        local = {"host": s["host"], "gpus": s["gpus"], "pods": s["pods"], "vllm": s["vllm"], "alerts": self.alerts_public()}
        return self.peers.cluster(local)

    # ---- lifecycle ---------------------------------------------------------------------
    def start(self) -> "Monitor":
        self.started = self.clock()
        self.tick_all()
        for t in self.tasks.values():
            t.start()
        return self

    def stop(self) -> None:
        for t in self.tasks.values():
            t.stop()
        if self.state_dir:
            try:
                self.checkpoint()
            except OSError:
                pass
        if self.notifier is not None:
            self.notifier.flush(1.0)
            self.notifier.close()
        for s in (self.gpu, self.kube, self.vllm):
            try:
                s.close()
            except Exception:
                pass

    # ---- reads (HTTP handlers) -----------------------------------------------------------
    def get(self, key: str) -> Any:
        with self._lock:
            return self._snap[key]

    def snapshot(self) -> Dict[str, Any]:
        with self._lock:
            return dict(self._snap)

    def alerts_public(self) -> Dict[str, List[Dict[str, Any]]]:
        """The reference's shape: three lists of ``{title, desc, fix}`` (+ ``since``, ``key``, ``signal`` unless
        ``compat.alert_extra_fields`` is off). With ``compat.alerts_per_request`` the rules are evaluated here, per
        call, on the current snapshot against a state this call replaces: the reference's semantics exactly
        (monitor_server.js:282-288 and :235), including its first-sight and alert-stealing behaviour. The
        collectors still do not run per request."""
        keep = ("title", "desc", "fix", "since", "key", "signal") if self.cfg.compat.alert_extra_fields else ("title", "desc", "fix")
        with self._lock:
            if self.cfg.compat.alerts_per_request:
                s = self._snap
                a, self._request_alert_state = check_alerts(s["host"], s["gpus"], s["pods"], self._request_alert_state, self.cfg,
                                                            s["vllm"] if s["vllm"].get("available") else None)
            else:
                a = self._snap["alerts"]
            return {tier: [{k: x[k] for k in keep if k in x} for x in a[tier]] for tier in ("minor", "serious", "critical")}

    def prometheus_available(self) -> bool:
        now = time.monotonic()
        if self._prom_ok is None or now - self._prom_checked > 60.0:
            self._prom_ok = self.prom.reachable()
            self._prom_checked = now
        return bool(self._prom_ok)

    def history(self) -> Dict[str, Any]:
        mode = self._history_mode
        if mode == "auto":
            mode = "prometheus" if self.prometheus_available() else "local"
        if mode == "prometheus":
            # N open dashboards must not become 6N range queries: one result serves everybody for
            # a few seconds (the data has a 30 s step; the page asks every 30 s)
            ttl = min(5.0, self.cfg.periods.history_step / 2)
            now = time.monotonic()
            cache = self._hist_cache
            if cache is not None and now - cache[0] < ttl:
                return cache[1]
            # ONE refresh at a time, and nobody waits for it with a result in hand: with Prometheus slow (six queries
            # of up to `timeout` each) every other request gets the previous result instead of queueing on a lock
            # inside the executor, where a handful of dashboards used to occupy every thread
            if cache is not None and not self._hist_lock.acquire(blocking=False):
                return cache[1]
            if cache is None:
                self._hist_lock.acquire()
            try:
                cache = self._hist_cache
                if cache is not None and time.monotonic() - cache[0] < ttl:
                    return cache[1]
                out = self._history_from_prometheus()
                self._hist_cache = (time.monotonic(), out)
                return out
            finally:
                self._hist_lock.release()
        out = self.local_history.payload()
        out["source"] = "local"
        return out

    def _history_from_prometheus(self) -> Dict[str, Any]:
        out = self.prom_history.payload()
        # MI355X extras always come from the local ring (no PromQL configured for them)
        local = self.local_history.payload()
        for k, v in local.items():
            out.setdefault(k, v)
        out["source"] = "prometheus"
        return out

    def stats(self) -> Dict[str, Any]:
        return {"uptime_s": round(self.clock() - self.started, 3) if self.started else 0.0,
                "sources": {"gpu": self.gpu.describe(), "kube": self.kube.describe(), "vllm": self.vllm.describe(),
                            "history": self._history_mode, "prometheus": self.prom.url},
                "tasks": {k: t.stats() for k, t in self.tasks.items()},
                "collectors": {"host_ms": round(self.host.last_duration_s * 1e3, 4),
                               "gpu_ms": round(self.gpu.last_duration_s * 1e3, 4),
                               "kube_ms": round(self.kube.last_duration_s * 1e3, 4),
                               "vllm_ms": round(self.vllm.last_duration_s * 1e3, 4)},
                "errors": {"gpu": self.gpu.last_error, "kube": self.kube.last_error, "vllm": self.vllm.last_error,
                           "prometheus": self.prom.last_error},
                "history_points": {n: len(self.store.series(n)) for n in self.store.names()[:64]},
                "alert_evaluations": self._alert_state.evaluations,
                "alert_watch": {"period_s": self.cfg.periods.alert_watch if "watch" in self.tasks else 0.0, "early_ticks": self.watch_pokes},
                "state_dir": self.state_dir or None, "restored": self.restored,
                "webhook": self.notifier.stats() if self.notifier is not None else None}
