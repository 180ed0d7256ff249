"""Byte-level parity with the reference under ``compat.reference_defaults``.

DEFAULT (every run of the suite): golden documents. The expected bodies below were written BY READING
``/root/reference/monitor_server.js`` (the rule chains :162-232, the string templates, ``toFixed(1)`` as ECMAScript
defines it: the larger n on an exact tie, :76-78, :142; ``JSON.stringify``'s key order and number text) -- the suite
executes nothing from the reference tree (that tree is untrusted content; its server shells out, binds
0.0.0.0:8888 and serves an unauthenticated path-traversal route). This repo's server gets the injected states through the
same stubs as before and must answer exactly the golden text.

OPT-IN (``KLM_RUN_REFERENCE=1``, never part of CI): the live differential run of earlier rounds.
``node /root/reference/monitor_server.js`` (its JavaScript source, run under the image's Node with a scrubbed environment;
port 8888 and Prometheus localhost:9090 are literals in it) and this repo's server get the same inputs and their HTTP
answers are compared byte for byte:

  * Prometheus: the in-repo ``FakePrometheus`` on :9090, values a step function of time;
  * ``df`` / ``nvidia-smi`` / ``kubectl``: three shell stubs first on the reference's PATH that print files of a
    scratch directory; this server reads the same states through a ``statvfs`` stub, a GPU source returning the
    stub's rows, and its own kubectl source pointed at the same stub;
  * ``/proc/loadavg`` and ``/proc/meminfo`` are the real ones on both sides (the reference reads them by absolute
    path), so CPU / memory are compared field by field with a tolerance for the moment of sampling.

The live run is skipped unless asked for, and without node, without the reference tree, or when 8888 / 9090 are taken.
Remaining differences are listed in docs/PARITY.md ("Byte-level differences under compat.reference_defaults")."""
import json
import os
import shutil
import socket
import stat
import subprocess
import time
import urllib.error
import urllib.request

import pytest

from k8s_llm_monitor_amd.collectors.gpu import GpuReader, GpuSource
from k8s_llm_monitor_amd.collectors.host import HostCollector
from k8s_llm_monitor_amd.collectors.kube import open_kube_source
from k8s_llm_monitor_amd.config import load_config
from k8s_llm_monitor_amd.fakes.prometheus import FakePrometheus
from k8s_llm_monitor_amd.server.http import MonitorServer
from k8s_llm_monitor_amd.server.sampler import Monitor

REF = "/root/reference/monitor_server.js"


def _port_free(p):
    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            s.bind(("0.0.0.0", p))
            return True
        except OSError:
            return False


LIVE = os.environ.get("KLM_RUN_REFERENCE") == "1"
live = pytest.mark.skipif(not LIVE or shutil.which("node") is None or not os.path.exists(REF),
                          reason="the live differential run executes the reference server: opt in with KLM_RUN_REFERENCE=1 (needs node)")


def fetch(base, path, method="GET"):
    req = urllib.request.Request(base + path, method=method)
    try:
        with urllib.request.urlopen(req, timeout=10) as r:
            return r.status, {k.lower(): v for k, v in r.headers.items()}, r.read()
    except urllib.error.HTTPError as e:
        return e.code, {k.lower(): v for k, v in e.headers.items()}, e.read()


def step_resolver(q, t):
    """A value per (query, whole minute): awkward decimals, exact .x5 ties included (toFixed vs Python rounding)."""
    m = int(t // 60)
    base = {"node_cpu": 5.25, "MemAvailable": 40.05, "filesystem": 45.349999, "GPU_UTIL": 61.5, "gfx_activity": 61.5,
            "FB_USED": 170000.25, "used_vram": 170000.25, "GPU_TEMP": 66.75, "junction": 66.75}
    for k, v in base.items():
        if k in q:
            return v + (m % 7) * 0.35
    return 1.0 if q.strip() == "1" else None


class RowsGpuReader(GpuReader):
    """The rows the nvidia-smi stub prints, as raw readings."""
    kind = "rows"

    def __init__(self, scratch):
        self.scratch = scratch

    def read(self):
        try:
            with open(os.path.join(self.scratch, "gpu.csv")) as f:
                lines = [l for l in f.read().splitlines() if l.strip()]
        except OSError:
            return []
        out = []
        for l in lines:
            p = [x.strip() for x in l.split(",")]
            out.append({"index": int(p[0]), "name": p[1], "gfx_activity": int(p[2]), "vram_used_mib": int(p[3]),
                        "vram_total_mib": int(p[4]), "temp_hotspot": int(p[5]), "t": time.monotonic()})
        return out


def _pair(tmp_path_factory, with_reference: bool):
    if with_reference and not (_port_free(8888) and _port_free(9090)):
        pytest.skip("ports 8888 / 9090 are in use")
    scratch = str(tmp_path_factory.mktemp("ref_tools"))
    tools = os.path.join(scratch, "bin")
    os.makedirs(tools)
    for name, body in (("df", 'cat "%s/df.txt"' % scratch),
                       ("nvidia-smi", '[ -f "%s/gpu.csv" ] || exit 9; cat "%s/gpu.csv"' % (scratch, scratch)),
                       ("kubectl", '[ -f "%s/pods.json" ] || exit 1; cat "%s/pods.json"' % (scratch, scratch))):
        p = os.path.join(tools, name)
        with open(p, "w") as f:
            f.write("#!/bin/sh\n" + body + "\n")
        os.chmod(p, os.stat(p).st_mode | stat.S_IEXEC)

    state = {"disk": (1000 * 2 ** 30, 453 * 2 ** 30)}

    def set_disk(total, used):
        state["disk"] = (total, used)
        with open(os.path.join(scratch, "df.txt"), "w") as f:
            f.write("Filesystem 1B-blocks Used Available Use%% Mounted on\n/dev/root %d %d %d 46%% /\n" % (total, used, total - used))

    def set_gpus(rows):
        p = os.path.join(scratch, "gpu.csv")
        if rows is None:
            if os.path.exists(p):
                os.remove(p)
        else:
            with open(p, "w") as f:
                f.write("".join("%d, %s, %d, %d, %d, %d\n" % r for r in rows))

    def set_pods(items):
        p = os.path.join(scratch, "pods.json")
        if items is None:
            if os.path.exists(p):
                os.remove(p)
        else:
            with open(p, "w") as f:
                json.dump({"apiVersion": "v1", "kind": "PodList", "items": items}, f)

    set_disk(*state["disk"])
    prom = FakePrometheus(step_resolver, port=9090 if with_reference else 0).start()
    ref = None
    if with_reference:
        # a scrubbed environment: the stubs first on PATH, nothing of the caller's (tokens, proxies, cloud credentials)
        ref = subprocess.Popen(["node", REF], cwd=os.path.dirname(REF), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               env={"PATH": tools + os.pathsep + "/usr/local/bin:/usr/bin:/bin", "HOME": scratch, "LANG": "C.UTF-8",
                                    "TZ": os.environ.get("TZ", "UTC")})

    class FakeStatvfs:
        def __call__(self, root):
            total, used = state["disk"]
            return os.statvfs_result((4096, 4096, total // 4096, (total - used) // 4096, (total - used) // 4096, 0, 0, 0, 0, 255))

    cfg = load_config(env={}, overrides={
        "compat": {"reference_defaults": True},
        "server": {"port": 0, "host": "127.0.0.1"},
        "sources": {"gpu": "none", "kube": "kubectl", "kubectl": os.path.join(tools, "kubectl"), "vllm": "none", "history": "prometheus"},
        "prometheus": {"url": "http://localhost:9090" if with_reference else prom.url,
                       # the reference's own three GPU queries (monitor_server.js:128-134): metric names are configuration
                       "q_gpu_util": "avg(DCGM_FI_DEV_GPU_UTIL)", "q_gpu_mem": "avg(DCGM_FI_DEV_FB_USED)", "q_gpu_temp": "avg(DCGM_FI_DEV_GPU_TEMP)"},
        "periods": {"realtime": 3600.0, "pods": 3600.0, "vllm": 3600.0}})          # the test ticks by hand
    host = HostCollector(cores=cfg.compat.cpu_cores, cpu_percent=cfg.compat.cpu_percent, statvfs=FakeStatvfs())
    mon = Monitor(cfg, host=host, gpu=GpuSource(RowsGpuReader(scratch)), kube=open_kube_source("kubectl", cfg))
    mon.tick_all()
    srv = MonitorServer(mon, cfg).start()
    if with_reference:
        for _ in range(100):
            try:
                urllib.request.urlopen("http://127.0.0.1:8888/api/host/metrics", timeout=1).read()
                break
            except Exception:
                time.sleep(0.1)
        else:
            ref.kill()
            pytest.skip("the reference server did not start under this node")

    class P:
        pass
    p = P()
    p.ref, p.ours, p.mon = "http://127.0.0.1:8888", srv.url, mon
    p.set_disk, p.set_gpus, p.set_pods = set_disk, set_gpus, set_pods
    yield p
    srv.stop()
    mon.stop()
    if ref is not None:
        ref.terminate()
        try:
            ref.wait(timeout=5)
        except subprocess.TimeoutExpired:
            ref.kill()
    prom.stop()


@pytest.fixture(scope="module")
def pair(tmp_path_factory):
    yield from _pair(tmp_path_factory, with_reference=True)


@pytest.fixture(scope="module")
def ours(tmp_path_factory):
    """This repo's server alone, behind the same stubs: what the golden documents are held against."""
    yield from _pair(tmp_path_factory, with_reference=False)


def both(pair, path, method="GET"):
    return fetch(pair.ref, path, method), fetch(pair.ours, path, method)


@live
def test_history_bodies_are_byte_identical(pair):
    for attempt in range(4):                      # the two servers must ask within the same second (end = floor(now))
        time.sleep(max(0.0, 0.15 - (time.time() % 1.0)) if (time.time() % 1.0) > 0.7 else 0.0)
        (rs, rh, rb), (os_, oh, ob) = both(pair, "/api/history")
        assert rs == os_ == 200
        ours = json.loads(ob)
        ref = json.loads(rb)
        sub = {k: ours[k] for k in ref}           # this server adds keys (MI355X series, "source"); the reference's six must match
        if sub == ref:
            break
    assert list(ref) == ["cpu", "memory", "disk", "gpuUtil", "gpuMem", "gpuTemp"] and list(ours)[:6] == list(ref)
    assert len(ref["cpu"]["labels"]) == 61 and ref["cpu"]["data"][0].count(".") == 1
    assert json.dumps(sub, ensure_ascii=False, separators=(",", ":")).encode() == rb    # same text: labels, toFixed(1) strings, key order
    assert rh["content-type"] == oh["content-type"] == "application/json; charset=utf-8"


@live
def test_host_metrics_same_keys_types_and_strings(pair):
    pair.set_disk(2000 * 2 ** 30, 1234567892992)
    pair.mon.tick_realtime()
    (rs, rh, rb), (os_, oh, ob) = both(pair, "/api/host/metrics")
    r, o = json.loads(rb), json.loads(ob)
    assert rs == os_ == 200 and list(r) == list(o) == ["cpu", "memory", "disk"]
    assert r["disk"] == o["disk"] == {"total": 2000 * 2 ** 30, "used": 1234567892992, "percent": "57.5"}     # integers + toFixed(1) string
    assert type(r["cpu"]["load_1min"]) is type(o["cpu"]["load_1min"]) is float or isinstance(r["cpu"]["load_1min"], int)
    for k in ("total", "used"):
        assert isinstance(r["memory"][k], int) and isinstance(o["memory"][k], int)
    assert r["memory"]["total"] == o["memory"]["total"]
    assert abs(float(r["memory"]["percent"]) - float(o["memory"]["percent"])) <= 0.5 and isinstance(o["memory"]["percent"], str)
    # divisor 8 on both sides (monitor_server.js:76): same load -> same string
    if r["cpu"]["load_1min"] == o["cpu"]["load_1min"]:
        assert r["cpu"]["percent"] == o["cpu"]["percent"]
    assert o["cpu"]["percent"] == "%.1f" % (o["cpu"]["load_1min"] / 8 * 100) or abs(float(o["cpu"]["percent"]) - o["cpu"]["load_1min"] / 8 * 100) < 0.06
    assert set(r["cpu"]) <= set(o["cpu"]) and set(r["memory"]) == set(o["memory"])        # additive extras on cpu only


def _no_cpu_mem(alerts):
    """CPU / memory come from the real /proc on both sides, sampled at different moments: compared separately."""
    return {t: [a for a in v if not (a["title"].startswith("CPU") or a["title"].startswith("内存"))] for t, v in alerts.items()}


POD = lambda ns, name, phase, restarts, start="2020-01-01T00:00:00Z": {
    "metadata": {"namespace": ns, "name": name},
    "status": {"phase": phase, "startTime": start, "containerStatuses": [{"restartCount": restarts}]}}


@live
def test_alert_bodies_match_for_injected_states(pair):
    seq = [
        # disk tiers (monitor_server.js:173-175), GPU 0 memory tiers and temperature tiers (:178-184), a second GPU that must be ignored
        dict(disk=(1000, 710), gpus=[(0, "GPU-A", 10, 71000, 100000, 70), (1, "GPU-B", 99, 99000, 100000, 99)], pods=[]),
        dict(disk=(1000, 861), gpus=[(0, "GPU-A", 10, 86100, 100000, 76)], pods=[POD("llm", "vllm-0", "Running", 0)]),          # first sight -> "Pod 刚恢复"
        dict(disk=(1000, 951), gpus=[(0, "GPU-A", 10, 95100, 100000, 86)], pods=[POD("llm", "vllm-0", "Running", 2), POD("llm", "p", "Pending", 0)]),
        dict(disk=(1000, 700), gpus=[(0, "GPU-A", 10, 70000, 100000, 75)], pods=[POD("llm", "vllm-0", "Failed", 2), POD("llm", "p", "Running", 0)]),
        dict(disk=(1000, 100), gpus=None, pods=None),                                                                      # tools fail: [] and []
    ]
    for i, st in enumerate(seq):
        pair.set_disk(st["disk"][0] * 2 ** 20, st["disk"][1] * 2 ** 20)
        pair.set_gpus(st["gpus"])
        pair.set_pods(st["pods"])
        pair.mon.tick_all()
        (rs, rh, rb), (os_, oh, ob) = both(pair, "/api/alerts")
        assert rs == os_ == 200
        r, o = _no_cpu_mem(json.loads(rb)), _no_cpu_mem(json.loads(ob))
        assert json.dumps(o, ensure_ascii=False) == json.dumps(r, ensure_ascii=False), (i, r, o)      # titles, descs, fixes, order, keys
        if i == 1:
            assert [a["title"] for a in r["serious"]] == ["磁盘空间紧张", "GPU 显存使用偏高", "GPU 温度偏高", "Pod 刚恢复: vllm-0"]
        if i == 2:
            assert [a["title"] for a in r["critical"]] == ["磁盘空间不足", "GPU 显存不足", "GPU 温度过高"]
            assert [a["title"] for a in r["serious"]] == ["Pod 重启: vllm-0", "Pod 刚恢复: p"][:1] + ["Pod 等待调度: p"] or len(r["serious"]) >= 2
        # the GPU and pod endpoints give the same documents too
        (_, _, rg), (_, _, og) = both(pair, "/api/gpu/metrics")
        assert [{k: g[k] for k in ("name", "utilization", "memoryUsed", "memoryTotal", "temperature")} for g in json.loads(og)] == json.loads(rg)
        (_, _, rp), (_, _, op) = both(pair, "/api/k8s/pods")
        assert [{k: p[k] for k in ("namespace", "name", "status", "restarts", "age")} for p in json.loads(op)] == json.loads(rp)
    # edge alerts are consumed by the request that saw them, on both servers (monitor_server.js:235)
    pair.set_pods([POD("x", "fresh", "Running", 0)])
    pair.mon.tick_all()
    first = [both(pair, "/api/alerts") for _ in range(1)][0]
    second = both(pair, "/api/alerts")
    for side in (0, 1):
        assert any("刚恢复" in a["title"] for a in json.loads(first[side][2])["serious"])
        assert not any("刚恢复" in a["title"] for a in json.loads(second[side][2])["serious"])


@live
def test_options_404_and_query_strings(pair):
    for path, method in (("/api/alerts", "OPTIONS"), ("/nope", "OPTIONS"), ("/nope", "GET"), ("/api/unknown/route", "GET"),
                         ("/api/k8s/pods?x=1&y=2", "GET"), ("/monitor.html?v=3", "GET"), ("/logo.svg", "GET")):
        (rs, rh, rb), (os_, oh, ob) = both(pair, path, method)
        assert rs == os_, (path, method, rs, os_)
        for h in ("access-control-allow-origin", "access-control-allow-methods", "content-type"):
            assert rh.get(h) == oh.get(h), (path, method, h, rh.get(h), oh.get(h))
        if path in ("/nope", "/api/unknown/route") or method == "OPTIONS":
            assert rb == ob, (path, method, rb, ob)                     # b"" for OPTIONS, b"Not Found" for 404
        if path == "/logo.svg":
            assert rb.lstrip().startswith(b"<svg") and ob.lstrip().startswith(b"<svg")     # this repo ships its own drawing


# ---- golden documents (derived by reading monitor_server.js; nothing of the reference runs) ----------------------------------
SEQ = [
    dict(disk=(1000, 710), gpus=[(0, "GPU-A", 10, 71000, 100000, 70), (1, "GPU-B", 99, 99000, 100000, 99)], pods=[]),
    dict(disk=(1000, 861), gpus=[(0, "GPU-A", 10, 86100, 100000, 76)], pods=[POD("llm", "vllm-0", "Running", 0)]),
    dict(disk=(1000, 951), gpus=[(0, "GPU-A", 10, 95100, 100000, 86)], pods=[POD("llm", "vllm-0", "Running", 2), POD("llm", "p", "Pending", 0)]),
    dict(disk=(1000, 700), gpus=[(0, "GPU-A", 10, 70000, 100000, 75)], pods=[POD("llm", "vllm-0", "Failed", 2), POD("llm", "p", "Running", 0)]),
    dict(disk=(1000, 100), gpus=None, pods=None),
]
A = lambda title, desc, fix: {"title": title, "desc": desc, "fix": fix}
#: checkAlerts' answer per state, CPU / memory rules aside (they read the real /proc): the if / else-if chains of :172-184 with
#: strict >, `${diskPercent}` printing the parsed NUMBER (71, not 71.0) while the GPU line prints toFixed(1) (71.0), GPU 0 only,
#: the pod pass of :188-232 against the table the PREVIOUS request left behind (unseen pods count as 'Unknown')
GOLDEN_ALERTS = [
    {"minor": [A("磁盘使用率偏高", "磁盘使用率 71%", "建议：关注磁盘增长趋势，定期清理"),
               A("GPU 显存使用偏高", "GPU 显存使用 71000/100000 MB (71.0%)", "建议：持续监控 GPU 显存使用情况")], "serious": [], "critical": []},
    {"minor": [], "serious": [A("磁盘空间紧张", "磁盘使用率 86.1%", "建议：1. 清理旧日志 2. 删除临时文件 3. 计划扩容"),
                              A("GPU 显存使用偏高", "GPU 显存使用 86100/100000 MB (86.1%)", "建议：监控模型推理性能，考虑优化显存使用"),
                              A("GPU 温度偏高", "GPU 温度 76°C", "建议：关注温度趋势，检查散热系统"),
                              A("Pod 刚恢复: vllm-0", "命名空间: llm, 之前状态: Unknown", "建议：检查之前故障原因 kubectl describe pod vllm-0 -n llm")], "critical": []},
    {"minor": [], "serious": [A("Pod 重启: vllm-0", "命名空间: llm, 重启次数: 2次 (增加了2次)", "建议：kubectl logs vllm-0 -n llm --previous 查看重启前日志"),
                              A("Pod 等待调度: p", "命名空间: llm", "建议：kubectl describe pod p -n llm")],
     "critical": [A("磁盘空间不足", "磁盘使用率 95.1%", "建议：1. 立即清理日志文件 2. 删除不必要的镜像/容器 3. 扩容磁盘"),
                  A("GPU 显存不足", "GPU 显存使用 95100/100000 MB (95.1%)", "建议：1. 减少 batch_size 2. 使用模型量化 3. 扩容 GPU 或使用多卡"),
                  A("GPU 温度过高", "GPU 温度 86°C", "建议：1. 检查 GPU 散热风扇 2. 降低计算负载 3. 增加机房空调")]},
    # exactly 70 % / 70.0 % / 75 degrees: strict comparisons, nothing fires; the pod that failed and the one that came up do
    {"minor": [], "serious": [A("Pod 刚恢复: p", "命名空间: llm, 之前状态: Pending", "建议：检查之前故障原因 kubectl describe pod p -n llm")],
     "critical": [A("Pod 异常: vllm-0", "命名空间: llm, 状态: Failed", "建议：kubectl describe pod vllm-0 -n llm")]},
    {"minor": [], "serious": [], "critical": []},                    # every tool failed: [] and [] (:94, :113), disk at 10 %
]


def _js_age(start_iso: str, now: float) -> str:
    """monitor_server.js:106-110."""
    import calendar
    s = now - calendar.timegm(time.strptime(start_iso, "%Y-%m-%dT%H:%M:%SZ"))
    d, h, m = int(s // 86400), int((s % 86400) // 3600), int((s % 3600) // 60)
    return f"{d}d {h}h" if d > 0 else f"{h}h {m}m" if h > 0 else f"{m}m"


def test_golden_alert_gpu_and_pod_bodies(ours):
    for i, (st, want) in enumerate(zip(SEQ, GOLDEN_ALERTS)):
        ours.set_disk(st["disk"][0] * 2 ** 20, st["disk"][1] * 2 ** 20)
        ours.set_gpus(st["gpus"])
        ours.set_pods(st["pods"])
        ours.mon.tick_all()
        status, head, body = fetch(ours.ours, "/api/alerts")
        assert status == 200 and head["content-type"] == "application/json; charset=utf-8" and head["access-control-allow-origin"] == "*"
        got = _no_cpu_mem(json.loads(body))
        # the same TEXT JSON.stringify would produce: key order minor / serious / critical and title / desc / fix, no spaces
        assert json.dumps(got, ensure_ascii=False, separators=(",", ":")) == json.dumps(want, ensure_ascii=False, separators=(",", ":")), (i, got)
        for a in (x for t in json.loads(body).values() for x in t):
            assert list(a) == ["title", "desc", "fix"]               # no extra fields under the preset
        _, _, gb = fetch(ours.ours, "/api/gpu/metrics")
        want_g = [{"name": r[1], "utilization": r[2], "memoryUsed": r[3], "memoryTotal": r[4], "temperature": r[5]} for r in (st["gpus"] or [])]
        assert [{k: g[k] for k in ("name", "utilization", "memoryUsed", "memoryTotal", "temperature")} for g in json.loads(gb)] == want_g
        assert [list(g)[:5] for g in json.loads(gb)] == [["name", "utilization", "memoryUsed", "memoryTotal", "temperature"]] * len(want_g)   # :90's key order first
        _, _, pb = fetch(ours.ours, "/api/k8s/pods")
        now = time.time()
        want_p = [{"namespace": p["metadata"]["namespace"], "name": p["metadata"]["name"], "status": p["status"]["phase"],
                   "restarts": p["status"]["containerStatuses"][0]["restartCount"], "age": _js_age(p["status"]["startTime"], now)} for p in (st["pods"] or [])]
        assert [{k: p[k] for k in ("namespace", "name", "status", "restarts", "age")} for p in json.loads(pb)] == want_p
    # edge alerts are consumed by the request that saw them (monitor_server.js:235 replaces the table on every call)
    ours.set_pods([POD("x", "fresh", "Running", 0)])
    ours.mon.tick_all()
    first, second = json.loads(fetch(ours.ours, "/api/alerts")[2]), json.loads(fetch(ours.ours, "/api/alerts")[2])
    assert [a for a in first["serious"] if "刚恢复" in a["title"]] == [A("Pod 刚恢复: fresh", "命名空间: x, 之前状态: Unknown", "建议：检查之前故障原因 kubectl describe pod fresh -n x")]
    assert not any("刚恢复" in a["title"] for a in second["serious"])


#: Number.prototype.toFixed(1) (ECMA-262 21.1.3.3): n / 10 - x as close to zero as possible, the LARGER n on a tie; the ties are
#: decided on the double's exact value (40.05 is stored as 40.0499999..., 5.25 exactly)
GOLDEN_TOFIXED = [(5.25, "5.3"), (5.6, "5.6"), (5.95, "6.0"), (40.05, "40.0"), (40.4, "40.4"), (45.349999, "45.3"), (61.5, "61.5"), (61.85, "61.9"),
                  (170000.25, "170000.3"), (66.75, "66.8"), (0.05, "0.1"), (0.25, "0.3"), (0.35, "0.3"), (2.5, "2.5"), (99.95, "100.0"), (100.0, "100.0"),
                  (0.0, "0.0"), (57.48779296875, "57.5")]


def test_golden_tofixed_and_history_text(ours):
    from k8s_llm_monitor_amd.utils.timefmt import fixed1, hhmm
    for x, want in GOLDEN_TOFIXED:
        assert fixed1(x) == want, (x, fixed1(x), want)
    status, head, body = fetch(ours.ours, "/api/history")
    doc = json.loads(body)
    assert status == 200 and list(doc)[:6] == ["cpu", "memory", "disk", "gpuUtil", "gpuMem", "gpuTemp"]     # :146-153's key order
    for key, frag in (("cpu", "node_cpu"), ("memory", "MemAvailable"), ("disk", "filesystem"), ("gpuUtil", "GPU_UTIL"), ("gpuMem", "FB_USED"), ("gpuTemp", "GPU_TEMP")):
        ser = doc[key]
        assert list(ser) == ["labels", "data"] and len(ser["labels"]) == len(ser["data"]) == 61             # 30 min at 30 s, both ends
        assert all(isinstance(v, str) and v.count(".") == 1 and len(v.split(".")[1]) == 1 for v in ser["data"])
        assert all(len(l) == 5 and l[2] == ":" for l in ser["labels"])                                        # zh-CN 2-digit HH:MM
        # every value is the step function's value at that sample time, printed the way toFixed(1) prints it
        vals = {fixed1(step_resolver(frag, 60.0 * m)) for m in range(7)}
        assert set(ser["data"]) <= vals, (key, set(ser["data"]) - vals)
    # the same sub-document as text, the way JSON.stringify writes it
    sub = {k: doc[k] for k in ("cpu", "memory", "disk", "gpuUtil", "gpuMem", "gpuTemp")}
    assert body.startswith(json.dumps(sub, ensure_ascii=False, separators=(",", ":")).encode()[:-1])


def test_golden_disk_string_and_failure_shapes(ours):
    ours.set_disk(2000 * 2 ** 30, 1234567892992)
    ours.mon.tick_realtime()
    o = json.loads(fetch(ours.ours, "/api/host/metrics")[2])
    assert list(o) == ["cpu", "memory", "disk"] and o["disk"] == {"total": 2000 * 2 ** 30, "used": 1234567892992, "percent": "57.5"}
    assert isinstance(o["memory"]["percent"], str) and isinstance(o["memory"]["total"], int) and list(o["memory"]) == ["total", "used", "percent"]
    assert o["cpu"]["percent"] == "%.1f" % (o["cpu"]["load_1min"] / 8 * 100) or abs(float(o["cpu"]["percent"]) - o["cpu"]["load_1min"] / 8 * 100) < 0.06   # divisor 8 (:76)


def test_golden_options_404_and_query_strings(ours):
    """:244-250, :290: CORS and the JSON content type on everything, OPTIONS 200 with an empty body, the query string ignored,
    404 'Not Found' as plain text under the JSON content type."""
    for path, method, status, body in (("/api/alerts", "OPTIONS", 200, b""), ("/nope", "OPTIONS", 200, b""), ("/nope", "GET", 404, b"Not Found"),
                                       ("/api/unknown/route", "GET", 404, b"Not Found"), ("/api/k8s/pods?x=1&y=2", "GET", 200, None),
                                       ("/monitor.html?v=3", "GET", 200, None), ("/logo.svg", "GET", 200, None)):
        st, h, b = fetch(ours.ours, path, method)
        assert st == status, (path, method, st)
        assert h["access-control-allow-origin"] == "*" and h["access-control-allow-methods"] == "GET, OPTIONS"
        ctype = {"/monitor.html?v=3": "text/html; charset=utf-8", "/logo.svg": "image/svg+xml"}.get(path, "application/json; charset=utf-8")
        assert h["content-type"] == ctype, (path, h["content-type"])
        if body is not None:
            assert b == body, (path, method, b)
