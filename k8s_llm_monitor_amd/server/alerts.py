"""Three-tier alert engine: minor 轻微 / serious 严重 / critical 紧急.

Reference: ``checkAlerts`` (monitor_server.js:159-238). Kept from it:
  * tier names and the response shape ``{"minor": [...], "serious": [...], "critical": [...]}``
    with ``{title, desc, fix}`` items;
  * strict ``>`` comparisons at 70 / 85 / 95 % for CPU, memory, disk and GPU memory
    (:163-182), at most one alert per signal, at its highest tier (if / else-if chains);
  * pod rules: ``Pending`` -> serious (:229), ``Failed``/``Error`` -> critical (:227),
    not-Running -> Running since the last evaluation -> serious (:201), restart count rose
    -> serious (:210); pod alerts carry a ready-to-paste kubectl command (:205, :214, :228);
  * a host collector failure gives NaN percentages, so no host alert fires (:162, :80).

Changed on purpose:
  * pure function: the previous pod states are an explicit argument and the new ones are
    returned; the module-level ``lastPodStates`` (:157) is gone. The sampler owns the state
    and evaluates once per tick, so concurrent clients no longer steal each other's
    edge-triggered alerts;
  * first-sight pods raise no "just recovered" alert unless ``compat.first_sight_alerts``
    (the reference seeds unseen pods with status ``Unknown``, :194-196);
  * every GPU is checked, not only GPU 0 (:178), unless ``compat.gpu0_only_alerts``;
  * MI355X rules added: junction / HBM temperature, power vs cap, uncorrectable ECC,
    xGMI link down, thermal / power throttling, a GPU dropping off the bus; vLLM rules:
    KV-cache usage, waiting queue depth, time to first token; container waiting reasons
    (CrashLoopBackOff, ImagePullBackOff...) that the reference cannot see.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

TIERS = ("minor", "serious", "critical")

Alert = Dict[str, Any]
Alerts = Dict[str, List[Alert]]


@dataclass
class AlertState:
    """Everything the engine remembers between two evaluations."""
    pods: Dict[str, Dict[str, Any]] = field(default_factory=dict)   # "ns/name" -> {status, restarts, age}
    gpu_count: Optional[int] = None
    ecc_correctable: Dict[int, int] = field(default_factory=dict)
    bad_pages_retired: Dict[int, int] = field(default_factory=dict)   # GPU index -> retired HBM pages at the last evaluation
    vllm_endpoints: int = 0          # most vLLM endpoints ever seen answering (0 = never seen: no "down" alert)
    evaluations: int = 0


def empty_alerts() -> Alerts:
    return {"minor": [], "serious": [], "critical": []}


def tier_of(value: float, tiers: Sequence[float]) -> Optional[str]:
    """Highest tier whose threshold `value` strictly exceeds; a negative threshold is off.
    NaN compares false everywhere, like the reference's parseFloat(undefined)."""
    if value is None or isinstance(value, float) and math.isnan(value):
        return None
    hit = None
    for name, thr in zip(TIERS, tiers):
        if thr is not None and thr >= 0 and value > thr:
            hit = name
    return hit


def _pf(x: Any) -> float:
    """JavaScript parseFloat on the API's own string/number fields."""
    try:
        return float(x)
    except (TypeError, ValueError):
        return float("nan")


def _fmt(x: float, nd: int = 1) -> str:
    return f"{x:.{nd}f}"


# ---- message catalogue -----------------------------------------------------------------
# zh wording is written for this project (same meaning as the reference's canned strings,
# monitor_server.js:163-184); en is an addition.
_MSG = {
    "zh": {
        "cpu": {"critical": ("CPU 使用率过高", "建议：1. 用 top 找出异常进程 2. 为节点增加 CPU 核心 3. 优化或迁移高占用服务"),
                "serious": ("CPU 使用率偏高", "建议：1. 查看占用最高的进程 2. 评估升级 CPU 或增加实例 3. 排查异常流量"),
                "minor": ("CPU 使用率轻度偏高", "建议：持续观察，必要时扩容")},
        "cpu_desc": "当前 CPU 使用率 {v}%",
        "memory": {"critical": ("内存使用率过高", "建议：1. 释放页缓存 echo 3 > /proc/sys/vm/drop_caches 2. 重启内存占用大的服务 3. 扩容内存"),
                   "serious": ("内存使用率偏高", "建议：1. 排查内存泄漏 2. 增加 Swap 3. 规划内存扩容"),
                   "minor": ("内存使用率轻度偏高", "建议：关注内存趋势，必要时扩容")},
        "memory_desc": "当前内存使用率 {v}%",
        "disk": {"critical": ("磁盘空间不足", "建议：1. 立即清理日志 2. 删除无用镜像/容器 3. 扩容磁盘"),
                 "serious": ("磁盘空间紧张", "建议：1. 清理旧日志 2. 删除临时文件 3. 计划扩容"),
                 "minor": ("磁盘使用率偏高", "建议：关注增长趋势，定期清理")},
        "disk_desc": "磁盘使用率 {v}%",
        "hbm": {"critical": ("GPU{g} HBM 显存不足", "建议：1. 减小 batch_size / max_num_seqs 2. 降低 gpu_memory_utilization 或启用量化 (FP8/MXFP4) 3. 增加张量并行卡数"),
                "serious": ("GPU{g} HBM 显存使用偏高", "建议：关注 KV cache 占用与推理延迟，评估显存优化"),
                "minor": ("GPU{g} HBM 显存使用偏高", "建议：持续观察 HBM 使用趋势")},
        "hbm_desc": "HBM 使用 {u}/{t} MB ({p}%)",
        "gpu_temp": {"critical": ("GPU{g} 结温过高", "建议：1. 检查液冷/风冷与进风温度 2. 降低计算负载或功耗上限 3. 检查机房空调"),
                     "serious": ("GPU{g} 结温偏高", "建议：关注温度趋势，检查散热系统")},
        "gpu_temp_desc": "结温 (junction) {v}°C",
        "hbm_temp": {"critical": ("GPU{g} HBM 温度过高", "建议：降低显存带宽负载，检查散热；HBM 过热会触发降频"),
                     "serious": ("GPU{g} HBM 温度偏高", "建议：关注 HBM 温度趋势")},
        "hbm_temp_desc": "HBM 温度 {v}°C",
        "power": {"serious": ("GPU{g} 功耗接近上限", "建议：功耗触顶会导致降频，检查 power cap 设置与负载分布"),
                  "minor": ("GPU{g} 功耗偏高", "建议：持续观察"),
                  "critical": ("GPU{g} 功耗超限", "建议：检查供电与 power cap")},
        "power_desc": "功耗 {w}W / 上限 {c}W ({p}%)",
        "ecc_ue": ("GPU{g} 出现不可纠正 ECC 错误", "不可纠正 ECC 错误 {n} 个", "建议：1. 迁移该卡上的负载 2. amd-smi bad-pages / RAS 检查 3. 联系硬件更换"),
        "ecc_ce": ("GPU{g} 可纠正 ECC 错误增加", "可纠正 ECC 错误累计 {n} 个 (新增 {d})", "建议：持续观察，增长过快时安排检修"),
        "bad_pages": ("GPU{g} HBM 坏页隔离", "已隔离 {n} 页，待隔离 {p} 页，无法隔离 {u} 页", "建议：1. amd-smi bad-pages 查看地址 2. 待隔离页在驱动重载后生效，安排维护窗口 3. 数量持续增长时更换"),
        "xgmi": ("GPU{g} xGMI 链路断开", "链路 {l} 状态 down{peer}", "建议：1. amd-smi xgmi 检查链路 2. RCCL 集合通信将绕行或失败，尽快隔离该节点"),
        "pcie": ("GPU{g} PCIe 链路出现重传", "PCIe replay 新增 {d} 次 (累计 {n})，L0 恢复 {r} 次", "建议：1. 检查 PCIe 插槽 / 线缆 / riser 2. lspci -vv 查看链路速率是否降级 3. 持续增长时安排检修"),
        "throttle_thm": ("GPU{g} 温度降频中", "降频原因: {r}", "建议：检查散热，降低负载；性能数据在降频期间不可信"),
        "throttle_ppt": ("GPU{g} 功耗受限降频", "功耗限制驻留 {p}%", "建议：满载时属正常；若影响吞吐可调整 power cap 或错峰"),
        "vllm_down": ("vLLM 服务不可达", "{n}/{m} 个 vLLM 端点有响应", "建议：1. kubectl get pods 查看推理 Pod 2. 查看 Pod 日志 (OOM / HIP 错误) 3. 检查 /metrics 端口与网络策略"),
        "gpu_lost": ("GPU 数量减少", "检测到 {n} 张 GPU，之前为 {m} 张", "建议：1. dmesg 查看 amdgpu 错误 2. amd-smi list 3. 必要时重置或下线节点"),
        "pod_failed": ("Pod 异常: {name}", "命名空间: {ns}, 状态: {st}", "建议：kubectl describe pod {name} -n {ns}"),
        "pod_pending": ("Pod 等待调度: {name}", "命名空间: {ns}", "建议：kubectl describe pod {name} -n {ns}"),
        "pod_waiting": ("Pod 容器异常: {name}", "命名空间: {ns}, 原因: {reason}", "建议：kubectl describe pod {name} -n {ns} && kubectl logs {name} -n {ns} --previous"),
        "pod_recovered": ("Pod 刚恢复: {name}", "命名空间: {ns}, 之前状态: {prev}", "建议：检查之前的故障原因 kubectl describe pod {name} -n {ns}"),
        "pod_restart": ("Pod 重启: {name}", "命名空间: {ns}, 重启次数: {n}次 (增加了{d}次)", "建议：kubectl logs {name} -n {ns} --previous 查看重启前日志"),
        "kv": {"critical": ("vLLM KV cache 即将耗尽", "建议：1. 降低 max_num_seqs / max_model_len 2. 增加副本或张量并行 3. 启用 FP8 KV cache"),
               "serious": ("vLLM KV cache 使用偏高", "建议：关注排队与抢占 (preemption)，评估扩容"),
               "minor": ("vLLM KV cache 使用偏高", "建议：持续观察")},
        "kv_desc": "KV cache 使用率 {v}%",
        "waiting": {"critical": ("vLLM 请求严重排队", "建议：立即扩容推理副本或限流"),
                    "serious": ("vLLM 请求排队", "建议：评估扩容，检查单请求长度"),
                    "minor": ("vLLM 出现排队", "建议：持续观察")},
        "waiting_desc": "等待中的请求 {v} 个 (运行中 {r} 个)",
        "ttft": {"critical": ("vLLM 首 token 延迟过高", "建议：检查排队、prefill 长度与 GPU 降频"),
                 "serious": ("vLLM 首 token 延迟偏高", "建议：检查排队与 prefill 负载"),
                 "minor": ("vLLM 首 token 延迟轻度偏高", "建议：持续观察")},
        "ttft_desc": "TTFT p99 {v}s",
    },
}
_MSG["en"] = {
    "cpu": {"critical": ("CPU usage critical", "Find runaway processes with top; add cores; move heavy services"),
            "serious": ("CPU usage high", "Check the top consumers; consider more instances; look for abnormal traffic"),
            "minor": ("CPU usage elevated", "Keep watching; scale out if it persists")},
    "cpu_desc": "CPU usage {v}%",
    "memory": {"critical": ("Memory usage critical", "Drop caches (echo 3 > /proc/sys/vm/drop_caches); restart large services; add memory"),
               "serious": ("Memory usage high", "Look for leaks; add swap; plan a memory upgrade"),
               "minor": ("Memory usage elevated", "Watch the trend")},
    "memory_desc": "memory usage {v}%",
    "disk": {"critical": ("Disk almost full", "Clean logs now; remove unused images/containers; grow the disk"),
             "serious": ("Disk space tight", "Clean old logs and temp files; plan expansion"),
             "minor": ("Disk usage elevated", "Watch growth; clean regularly")},
    "disk_desc": "disk usage {v}%",
    "hbm": {"critical": ("GPU{g} HBM nearly exhausted", "Lower batch size / max_num_seqs; lower gpu_memory_utilization or quantise (FP8/MXFP4); add tensor-parallel GPUs"),
            "serious": ("GPU{g} HBM usage high", "Watch KV-cache use and latency; consider memory optimisations"),
            "minor": ("GPU{g} HBM usage elevated", "Watch the HBM trend")},
    "hbm_desc": "HBM {u}/{t} MB ({p}%)",
    "gpu_temp": {"critical": ("GPU{g} junction temperature critical", "Check cooling and inlet temperature; reduce load or power cap"),
                 "serious": ("GPU{g} junction temperature high", "Watch the trend; check cooling")},
    "gpu_temp_desc": "junction {v}°C",
    "hbm_temp": {"critical": ("GPU{g} HBM temperature critical", "Reduce memory-bandwidth load; check cooling; HBM overheating throttles the GPU"),
                 "serious": ("GPU{g} HBM temperature high", "Watch the HBM temperature")},
    "hbm_temp_desc": "HBM {v}°C",
    "power": {"serious": ("GPU{g} power at the cap", "Power capping lowers clocks; check the cap and load placement"),
              "minor": ("GPU{g} power elevated", "Keep watching"),
              "critical": ("GPU{g} power above the cap", "Check power delivery and the cap")},
    "power_desc": "power {w}W of {c}W cap ({p}%)",
    "ecc_ue": ("GPU{g} uncorrectable ECC errors", "{n} uncorrectable ECC error(s)", "Drain the GPU; check amd-smi bad-pages / RAS; schedule replacement"),
    "ecc_ce": ("GPU{g} correctable ECC errors rising", "{n} correctable ECC errors so far (+{d})", "Keep watching; service the GPU if the rate grows"),
    "bad_pages": ("GPU{g} HBM pages retired", "{n} page(s) retired, {p} pending, {u} unreservable", "amd-smi bad-pages lists them; pending pages retire at the next driver reload: plan a window; replace the GPU if the count keeps growing"),
    "xgmi": ("GPU{g} xGMI link down", "link {l} is down{peer}", "Check amd-smi xgmi; RCCL collectives will reroute or fail; isolate the node"),
    "pcie": ("GPU{g} PCIe link replays", "{d} new PCIe replay(s) ({n} so far), {r} L0 recoveries", "Check the slot / cable / riser; lspci -vv for a downgraded link; service the node if the count keeps growing"),
    "throttle_thm": ("GPU{g} thermal throttling", "reasons: {r}", "Check cooling, reduce load; performance numbers are unreliable while throttled"),
    "throttle_ppt": ("GPU{g} power-limited", "power-limit residency {p}%", "Normal at full load; raise the cap or spread load if throughput matters"),
    "vllm_down": ("vLLM endpoint unreachable", "{n} of {m} vLLM endpoint(s) answering", "kubectl get pods for the serving pods; read their logs (OOM / HIP errors); check the /metrics port and network policy"),
    "gpu_lost": ("GPU count dropped", "{n} GPU(s) visible, {m} before", "Check dmesg for amdgpu errors; amd-smi list; reset or drain the node"),
    "pod_failed": ("Pod failed: {name}", "namespace: {ns}, status: {st}", "kubectl describe pod {name} -n {ns}"),
    "pod_pending": ("Pod pending: {name}", "namespace: {ns}", "kubectl describe pod {name} -n {ns}"),
    "pod_waiting": ("Pod container failing: {name}", "namespace: {ns}, reason: {reason}", "kubectl describe pod {name} -n {ns} && kubectl logs {name} -n {ns} --previous"),
    "pod_recovered": ("Pod just recovered: {name}", "namespace: {ns}, previous status: {prev}", "Find the earlier failure: kubectl describe pod {name} -n {ns}"),
    "pod_restart": ("Pod restarted: {name}", "namespace: {ns}, restarts: {n} (+{d})", "kubectl logs {name} -n {ns} --previous"),
    "kv": {"critical": ("vLLM KV cache nearly exhausted", "Lower max_num_seqs / max_model_len; add replicas or TP; use FP8 KV cache"),
           "serious": ("vLLM KV cache usage high", "Watch queueing and preemption; consider scaling"),
           "minor": ("vLLM KV cache usage elevated", "Keep watching")},
    "kv_desc": "KV cache usage {v}%",
    "waiting": {"critical": ("vLLM requests backing up badly", "Scale replicas or rate-limit now"),
                "serious": ("vLLM requests queueing", "Consider scaling; check request lengths"),
                "minor": ("vLLM queue forming", "Keep watching")},
    "waiting_desc": "{v} waiting request(s) ({r} running)",
    "ttft": {"critical": ("vLLM time to first token critical", "Check queueing, prefill length and GPU throttling"),
             "serious": ("vLLM time to first token high", "Check queueing and prefill load"),
             "minor": ("vLLM time to first token elevated", "Keep watching")},
    "ttft_desc": "TTFT p99 {v}s",
}

# The reference's own strings, byte for byte (monitor_server.js:163-184 host and GPU, :203-230 pods): an API output
# table that clients may key on, selected by ``compat.reference_text`` (part of ``compat.reference_defaults``).
# GPU titles carry no index there because only GPU 0 is checked (:178).
_MSG["ref"] = dict(_MSG["zh"], **{
    "cpu": {"critical": ("CPU 使用率过高", "建议：1. 检查是否有异常进程 2. 考虑扩容 CPU 核心数 3. 优化占用高的服务"),
            "serious": ("CPU 使用率偏高", "建议：1. 查看占用最高的进程 2. 考虑升级 CPU 或增加实例 3. 检查是否有DDOS攻击"),
            "minor": ("CPU 使用率轻度偏高", "建议：持续监控，必要时考虑扩容")},
    "cpu_desc": "当前 CPU 使用率 {v}%",
    "memory": {"critical": ("内存使用率过高", "建议：立即释放缓存 echo 3 > /proc/sys/vm/drop_caches 2. 重启占用内存大的服务 3. 扩容内存"),
               "serious": ("内存使用率偏高", "建议：1. 检查内存泄漏 2. 增加 Swap 3. 考虑扩容内存"),
               "minor": ("内存使用率轻度偏高", "建议：关注内存趋势，必要时扩容")},
    "memory_desc": "当前内存使用率 {v}%",
    "disk": {"critical": ("磁盘空间不足", "建议：1. 立即清理日志文件 2. 删除不必要的镜像/容器 3. 扩容磁盘"),
             "serious": ("磁盘空间紧张", "建议：1. 清理旧日志 2. 删除临时文件 3. 计划扩容"),
             "minor": ("磁盘使用率偏高", "建议：关注磁盘增长趋势，定期清理")},
    "disk_desc": "磁盘使用率 {v}%",
    "hbm": {"critical": ("GPU 显存不足", "建议：1. 减少 batch_size 2. 使用模型量化 3. 扩容 GPU 或使用多卡"),
            "serious": ("GPU 显存使用偏高", "建议：监控模型推理性能，考虑优化显存使用"),
            "minor": ("GPU 显存使用偏高", "建议：持续监控 GPU 显存使用情况")},
    "hbm_desc": "GPU 显存使用 {u}/{t} MB ({p}%)",
    "gpu_temp": {"critical": ("GPU 温度过高", "建议：1. 检查 GPU 散热风扇 2. 降低计算负载 3. 增加机房空调"),
                 "serious": ("GPU 温度偏高", "建议：关注温度趋势，检查散热系统")},
    "gpu_temp_desc": "GPU 温度 {v}°C",
    "pod_failed": ("Pod 异常: {name}", "命名空间: {ns}, 状态: {st}", "建议：kubectl describe pod {name} -n {ns}"),
    "pod_pending": ("Pod 等待调度: {name}", "命名空间: {ns}", "建议：kubectl describe pod {name} -n {ns}"),
    "pod_recovered": ("Pod 刚恢复: {name}", "命名空间: {ns}, 之前状态: {prev}", "建议：检查之前故障原因 kubectl describe pod {name} -n {ns}"),
    "pod_restart": ("Pod 重启: {name}", "命名空间: {ns}, 重启次数: {n}次 (增加了{d}次)", "建议：kubectl logs {name} -n {ns} --previous 查看重启前日志"),
})

#: container waiting reasons -> tier
WAITING_REASONS = {"CrashLoopBackOff": "critical", "ImagePullBackOff": "serious", "ErrImagePull": "serious",
                   "CreateContainerConfigError": "serious", "CreateContainerError": "serious",
                   "InvalidImageName": "serious", "RunContainerError": "critical", "OOMKilled": "critical"}
THERMAL_REASONS = ("prochot", "socket_thm", "vr_thm", "hbm_thm")


def _add(alerts: Alerts, tier: str, key: str, title: str, desc: str, fix: str, edge: bool = False, **extra) -> None:
    a: Alert = {"title": title, "desc": desc, "fix": fix, "key": key, "tier": tier}
    if edge:
        a["edge"] = True
    a.update(extra)
    alerts[tier].append(a)


def _tiered(alerts: Alerts, msgs, name: str, key: str, value: float, tiers: Sequence[float], desc: str, g: str = "") -> None:
    tier = tier_of(value, tiers)
    if tier is None or tier not in msgs[name]:
        return
    title, fix = msgs[name][tier]
    _add(alerts, tier, key, title.format(g=g), desc, fix, signal=name, value=value)


def check_alerts(host: Mapping[str, Any], gpus: Sequence[Mapping[str, Any]], pods: Iterable[Mapping[str, Any]],
                 state: Optional[AlertState] = None, cfg=None,
                 vllm: Optional[Mapping[str, Any]] = None) -> Tuple[Alerts, AlertState]:
    """Evaluate every rule once. Returns (alerts, next_state); `state` is not modified."""
    from ..config import Config
    cfg = cfg or Config()
    th, compat = cfg.thresholds, cfg.compat
    msgs = _MSG["ref"] if getattr(compat, "reference_text", False) else _MSG.get(compat.language, _MSG["zh"])
    ref_only = bool(getattr(compat, "reference_rules_only", False))     # only the rules of monitor_server.js:162-232
    state = state or AlertState()
    alerts = empty_alerts()

    # ---- host (monitor_server.js:162-175)
    cpu = _pf((host.get("cpu") or {}).get("percent"))
    _tiered(alerts, msgs, "cpu", "host/cpu", cpu, th.cpu, msgs["cpu_desc"].format(v=_js_num(cpu)))
    mem = _pf((host.get("memory") or {}).get("percent"))
    _tiered(alerts, msgs, "memory", "host/memory", mem, th.memory, msgs["memory_desc"].format(v=_js_num(mem)))
    disk = _pf((host.get("disk") or {}).get("percent"))
    _tiered(alerts, msgs, "disk", "host/disk", disk, th.disk, msgs["disk_desc"].format(v=_js_num(disk)))

    # ---- GPUs (monitor_server.js:177-185, GPU 0 only there)
    gpus = list(gpus)
    many = len(gpus) > 1
    checked = gpus[:1] if compat.gpu0_only_alerts else gpus
    next_ecc: Dict[int, int] = dict(state.ecc_correctable)
    next_bad: Dict[int, int] = dict(state.bad_pages_retired)
    for gpu in checked:
        idx = gpu.get("index", 0)
        g = f" {idx}" if many else ""
        k = f"gpu/{idx}"
        used, total = gpu.get("memoryUsed"), gpu.get("memoryTotal")
        if isinstance(used, (int, float)) and isinstance(total, (int, float)) and total > 0:
            pct = used / total * 100.0
            _tiered(alerts, msgs, "hbm", f"{k}/hbm", pct, th.hbm,
                    msgs["hbm_desc"].format(u=used, t=total, p=_fmt(pct)), g)
        temp = gpu.get("temperature")
        if isinstance(temp, (int, float)):
            _tiered(alerts, msgs, "gpu_temp", f"{k}/temp", float(temp), th.gpu_temp,
                    msgs["gpu_temp_desc"].format(v=_js_num(temp)), g)
        if ref_only:
            continue
        thbm = gpu.get("temperatureHbm")
        if isinstance(thbm, (int, float)):
            _tiered(alerts, msgs, "hbm_temp", f"{k}/hbm_temp", float(thbm), th.hbm_temp,
                    msgs["hbm_temp_desc"].format(v=_js_num(thbm)), g)
        pw, cap = gpu.get("powerW"), gpu.get("powerCapW")
        if isinstance(pw, (int, float)) and isinstance(cap, (int, float)) and cap > 0:
            pp = pw / cap * 100.0
            _tiered(alerts, msgs, "power", f"{k}/power", pp, th.power_pct,
                    msgs["power_desc"].format(w=_fmt(pw, 0), c=_fmt(cap, 0), p=_fmt(pp)), g)
        ecc = gpu.get("ecc") or {}
        ue = ecc.get("uncorrectable")
        if isinstance(ue, int) and ue > 0 and th.ecc_uncorrectable_tier in TIERS:
            t, d, f = msgs["ecc_ue"]
            _add(alerts, th.ecc_uncorrectable_tier, f"{k}/ecc_ue", t.format(g=g), d.format(n=ue), f, signal="ecc")
        ce = ecc.get("correctable")
        if isinstance(ce, int):
            before = state.ecc_correctable.get(idx)
            if before is not None and ce > before:
                t, d, f = msgs["ecc_ce"]
                _add(alerts, "minor", f"{k}/ecc_ce", t.format(g=g), d.format(n=ce, d=ce - before), f, edge=True, signal="ecc")
            next_ecc[idx] = ce
        bp = gpu.get("badPages")
        if isinstance(bp, dict):
            n_bp, p_bp, u_bp = (bp.get(x) or 0 for x in ("retired", "pending", "unreservable"))
            t, d, f = msgs["bad_pages"]
            if p_bp or u_bp:
                # memory that is known bad and still in use (waiting for the next driver reload, or impossible to take out):
                # level-triggered, serious
                _add(alerts, "serious", f"{k}/bad_pages", t.format(g=g), d.format(n=n_bp, p=p_bp, u=u_bp), f, signal="ras")
            else:
                # retired pages are history the driver has dealt with: say so when the count GROWS (the table shows the count)
                before = state.bad_pages_retired.get(idx)
                if before is not None and n_bp > before:
                    _add(alerts, "minor", f"{k}/bad_pages", t.format(g=g), d.format(n=n_bp, p=0, u=0), f, edge=True, signal="ras")
            next_bad[idx] = n_bp
        if th.xgmi_link_down_tier in TIERS:
            for link in (gpu.get("xgmi") or {}).get("links") or []:
                if link.get("state") == "down":
                    t, d, f = msgs["xgmi"]
                    peer = f" -> {link['peer']}" if link.get("peer") else ""
                    _add(alerts, th.xgmi_link_down_tier, f"{k}/xgmi/{link.get('link')}", t.format(g=g),
                         d.format(l=link.get("link"), peer=peer), f, signal="xgmi")
        perr = (gpu.get("pcie") or {}).get("errors") or {}
        rep = perr.get("replay") or {}
        if isinstance(rep.get("delta"), int) and rep["delta"] > 0:
            t, d, f = msgs["pcie"]
            _add(alerts, "minor", f"{k}/pcie", t.format(g=g),
                 d.format(d=rep["delta"], n=rep.get("count"), r=(perr.get("l0Recovery") or {}).get("count", 0)), f, edge=True, signal="pcie")
        thr = gpu.get("throttle") or {}
        res = thr.get("residencyPercent") or {}
        thermal = [r for r in THERMAL_REASONS if (res.get(r) or 0) > 0]
        if thermal and th.throttle_tier in TIERS:
            t, d, f = msgs["throttle_thm"]
            _add(alerts, th.throttle_tier, f"{k}/throttle_thm", t.format(g=g), d.format(r=", ".join(thermal)), f, signal="throttle")
        ppt = res.get("ppt")
        if isinstance(ppt, (int, float)) and ppt > 50.0:
            t, d, f = msgs["throttle_ppt"]
            _add(alerts, "minor", f"{k}/throttle_ppt", t.format(g=g), d.format(p=_fmt(ppt)), f, signal="throttle")
    if not ref_only and state.gpu_count is not None and len(gpus) < state.gpu_count:
        t, d, f = msgs["gpu_lost"]
        _add(alerts, "critical", "gpu/lost", t, d.format(n=len(gpus), m=state.gpu_count), f, signal="gpu_lost")
    # the count only ever ratchets up, so the alert stays level-triggered until the GPU is back
    next_gpu_count = max(len(gpus), state.gpu_count or 0)

    # ---- pods (monitor_server.js:187-232)
    cur_pods: Dict[str, Dict[str, Any]] = {}
    for pod in pods:
        ns, name = pod.get("namespace"), pod.get("name")
        status = pod.get("status")
        restarts = pod.get("restarts") or 0
        key = f"{ns}/{name}"
        cur_pods[key] = {"status": status, "restarts": restarts, "age": pod.get("age")}
        last = state.pods.get(key)
        seen = last is not None
        if last is None:
            last = {"status": "Unknown", "restarts": 0, "age": "0m"}     # :194-196
        fmt = dict(name=name, ns=ns)
        if last["status"] != "Running" and status == "Running" and (seen or compat.first_sight_alerts):
            t, d, f = msgs["pod_recovered"]
            _add(alerts, "serious", f"pod/{key}/recovered", t.format(**fmt), d.format(prev=last["status"], **fmt),
                 f.format(**fmt), edge=True, signal="pod")
        if restarts > last["restarts"] and restarts > 0 and (seen or compat.first_sight_alerts):
            t, d, f = msgs["pod_restart"]
            _add(alerts, "serious", f"pod/{key}/restart", t.format(**fmt),
                 d.format(n=restarts, d=restarts - last["restarts"], **fmt), f.format(**fmt), edge=True, signal="pod")
        if status in ("Failed", "Error"):
            t, d, f = msgs["pod_failed"]
            _add(alerts, "critical", f"pod/{key}/failed", t.format(**fmt), d.format(st=status, **fmt), f.format(**fmt), signal="pod")
        elif status == "Pending":
            t, d, f = msgs["pod_pending"]
            _add(alerts, "serious", f"pod/{key}/pending", t.format(**fmt), d.format(**fmt), f.format(**fmt), signal="pod")
        reason = pod.get("reason")
        if not ref_only and reason in WAITING_REASONS and status not in ("Failed", "Error"):
            t, d, f = msgs["pod_waiting"]
            _add(alerts, WAITING_REASONS[reason], f"pod/{key}/waiting", t.format(**fmt),
                 d.format(reason=reason, **fmt), f.format(**fmt), signal="pod")

    # ---- vLLM serving metrics (no reference rule)
    # an endpoint that answered once and no longer does is an outage, not "no data": level-triggered,
    # critical when nothing answers, serious when some replicas are gone (direct scraping counts them)
    if ref_only:
        vllm = None
    n_ep = 0
    if vllm:
        e = vllm.get("endpoints")
        n_ep = int(e) if isinstance(e, (int, float)) and e > 0 else 1
    if not ref_only and state.vllm_endpoints and n_ep < state.vllm_endpoints:
        t, d, f = msgs["vllm_down"]
        _add(alerts, "critical" if n_ep == 0 else "serious", "vllm/down", t, d.format(n=n_ep, m=state.vllm_endpoints), f,
             signal="vllm_down")
    next_vllm_endpoints = max(n_ep, state.vllm_endpoints)
    if vllm:
        kv = vllm.get("kvCachePercent")
        if isinstance(kv, (int, float)):
            _tiered(alerts, msgs, "kv", "vllm/kv", float(kv), th.kv_cache, msgs["kv_desc"].format(v=_fmt(kv)))
        w = vllm.get("waiting")
        if isinstance(w, (int, float)):
            _tiered(alerts, msgs, "waiting", "vllm/waiting", float(w), th.vllm_waiting,
                    msgs["waiting_desc"].format(v=_js_num(w), r=_js_num(vllm.get("running") or 0)))
        ttft = vllm.get("ttftP99")
        if isinstance(ttft, (int, float)):
            _tiered(alerts, msgs, "ttft", "vllm/ttft", float(ttft), th.vllm_ttft_p99_s, msgs["ttft_desc"].format(v=_fmt(ttft, 2)))

    nxt = AlertState(pods=cur_pods, gpu_count=next_gpu_count, ecc_correctable=next_ecc, bad_pages_retired=next_bad,
                     vllm_endpoints=next_vllm_endpoints, evaluations=state.evaluations + 1)
    return alerts, nxt


def _js_num(x: Any) -> str:
    """Number -> text the way a JS template literal prints it (85 not 85.0, 85.3 stays)."""
    if isinstance(x, float):
        if math.isnan(x):
            return "NaN"
        if x == int(x) and abs(x) < 1e15:
            return str(int(x))
        return repr(x)
    return str(x)


class AlertBook:
    """Holds edge-triggered alerts visible for a while and tracks when each alert began.

    The reference shows an edge alert in exactly one response and to exactly one client
    (SURVEY 2.7 item 2). Here the sampler evaluates once per tick and the book keeps edge
    alerts for ``hold_s`` seconds; every alert gets ``since`` (epoch seconds of first sight).
    """

    def __init__(self, hold_s: float = 30.0, debounce_s: float = 0.0):
        self.hold_s = hold_s
        self.debounce_s = debounce_s     # level alerts younger than this are not shown yet
        self.on_event = None             # callable(event: dict) for begin / end (notifier hook)
        self._announced: Dict[str, str] = {}      # key -> tier it was last announced at
        self._since: Dict[str, float] = {}
        self._held: Dict[str, Tuple[float, Alert]] = {}
        self.history: List[Dict[str, Any]] = []     # bounded log of alert begin/end events
        self.max_history = 500

    def update(self, alerts: Alerts, now: float, keep_prefixes: Sequence[str] = ()) -> Alerts:
        """`keep_prefixes`: key prefixes whose source has not reported yet (the sampler passes "pod/" until the first
        pod list is in): known alerts under them are neither ended nor re-announced by this evaluation."""
        out = empty_alerts()
        live = set()
        if keep_prefixes:
            live.update(k for k in self._since if k.startswith(tuple(keep_prefixes)))
        for tier in TIERS:
            for a in alerts[tier]:
                key = a.get("key") or a["title"]
                live.add(key)
                if key not in self._since:
                    self._since[key] = now
                a = dict(a, since=self._since[key])
                if a.get("edge"):
                    if key not in self._announced:
                        self._announced[key] = tier
                        self._log("begin", a, now)
                    self._held[key] = (now + self.hold_s, a)
                elif now - self._since[key] >= self.debounce_s:
                    if key not in self._announced:
                        self._log("begin", a, now)
                    elif self._announced[key] != tier:          # e.g. HBM minor -> critical
                        self._log("change", dict(a, previous_tier=self._announced[key]), now)
                    self._announced[key] = tier
                    out[tier].append(a)
        for key in [k for k in self._since if k not in live and k not in self._held]:
            if key in self._announced:
                self._log("end", {"key": key, "tier": self._announced.pop(key)}, now)
            del self._since[key]
        for key, (until, a) in list(self._held.items()):
            if now >= until:
                del self._held[key]
                if key not in live:
                    self._since.pop(key, None)
                    self._announced.pop(key, None)
                continue
            out[a["tier"]].append(a)
        return out

    # ---- checkpoint / resume: a restart must not re-announce what was already announced (webhook, history, `since`)
    def to_state(self) -> Dict[str, Any]:
        return {"announced": dict(self._announced), "since": dict(self._since),
                "held": {k: [until, a] for k, (until, a) in self._held.items()}, "history": list(self.history)}

    def load_state(self, d: Mapping[str, Any]) -> None:
        self._announced = {str(k): str(v) for k, v in (d.get("announced") or {}).items()}
        self._since = {str(k): float(v) for k, v in (d.get("since") or {}).items()}
        self._held = {str(k): (float(v[0]), dict(v[1])) for k, v in (d.get("held") or {}).items()}
        self.history = [dict(e) for e in (d.get("history") or [])][-self.max_history:]

    def _log(self, what: str, a: Mapping[str, Any], now: float) -> None:
        ev = {"t": now, "event": what, "key": a.get("key"), "tier": a.get("tier"), "title": a.get("title"),
              "desc": a.get("desc"), "fix": a.get("fix")}
        if a.get("previous_tier"):
            ev["previous_tier"] = a["previous_tier"]
        self.history.append(ev)
        if self.on_event is not None:
            try:
                self.on_event(ev)
            except Exception:
                pass
        if len(self.history) > self.max_history:
            del self.history[: len(self.history) - self.max_history]
