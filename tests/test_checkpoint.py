"""Checkpoint / resume of history and alert state (the reference keeps
nothing on disk, so a restart empties the charts' local source and re-fires first-sight alerts)."""
import json
import os

from k8s_llm_monitor_amd.collectors.gpu import GpuSource
from k8s_llm_monitor_amd.config import demo_config
from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader
from k8s_llm_monitor_amd.fakes.kube import FakeKube
from k8s_llm_monitor_amd.server.sampler import Monitor


def make(clock, state_dir, kube="steady"):
    cfg = demo_config(server={"state_dir": str(state_dir)}, periods={"history_step": 5.0})
    return Monitor(cfg, gpu=GpuSource(FakeGpuReader(2, clock=clock)), kube=FakeKube(kube, clock=clock), clock=clock)


def test_history_and_alert_state_survive_a_restart(tmp_path, clock):
    m1 = make(clock, tmp_path, kube="pending")
    for _ in range(4):
        m1.tick_all()
        clock.advance(5)
    assert len(m1.history()["cpu"]["data"]) == 4
    assert [a["key"] for a in m1.alerts_public()["serious"]] == ["pod/llm/vllm-llama3-405b-tp8-0/pending"]
    m1.stop()                                                     # writes the checkpoint
    assert os.path.exists(tmp_path / "history.json") and os.path.exists(tmp_path / "alert_state.json")

    m2 = make(clock, tmp_path, kube="steady")                     # the pending pod is now gone, the others still run
    assert m2.restored == {"history": True, "alert_state": True, "alert_book": True}
    assert len(m2.history()["cpu"]["data"]) == 4                  # charts keep their window
    assert "llm/vllm-llama3-8b-7f9c6d5b8-2xkzv" in m2._alert_state.pods
    m2.tick_all()
    assert m2.alerts_public()["serious"] == []                    # known pods: no first-sight noise, no stale alert
    assert m2.stats()["restored"]["history"] is True
    m2.stop()


def test_restored_pod_table_detects_a_restart_that_happened_while_down(tmp_path, clock):
    m1 = make(clock, tmp_path)
    m1.tick_all()
    m1.stop()
    from k8s_llm_monitor_amd.fakes.kube import base_pods

    def bumped(t):
        pods = base_pods()
        pods[0].restarts += 2
        return pods

    m2 = make(clock, tmp_path, kube=bumped)
    m2.tick_all()
    assert [a["key"] for a in m2.alerts_public()["serious"]] == ["pod/kube-system/coredns-6799fbcd5-x7k2p/restart"]
    m2.stop()


def test_stale_or_corrupt_state_is_ignored(tmp_path, clock):
    m1 = make(clock, tmp_path)
    m1.tick_all()
    m1.stop()
    clock.advance(7200)                                            # older than an hour: pod table not trusted
    m2 = make(clock, tmp_path)
    assert m2.restored["alert_state"] is False and m2._alert_state.pods == {}
    m2.stop()
    (tmp_path / "alert_state.json").write_text("{broken")
    (tmp_path / "history.json").write_text("[]")
    m3 = make(clock, tmp_path)
    assert m3.restored == {"history": False, "alert_state": False, "alert_book": False}
    m3.tick_all()
    m3.stop()
    json.loads((tmp_path / "alert_state.json").read_text())       # rewritten cleanly


def test_no_state_dir_writes_nothing(tmp_path, clock, monkeypatch):
    monkeypatch.chdir(tmp_path)
    cfg = demo_config()
    m = Monitor(cfg, gpu=GpuSource(FakeGpuReader(1, clock=clock)), kube=FakeKube(clock=clock), clock=clock)
    m.tick_all()
    m.stop()
    assert "persist" not in m.tasks and os.listdir(tmp_path) == []


def test_live_alert_is_not_announced_twice_across_a_restart(tmp_path, clock):
    """a restart used to log a fresh `begin` for every level-triggered alert (a second webhook
    post), reset `since` to the restart time and lose /api/alerts/history."""
    from k8s_llm_monitor_amd.fakes.gpu import GpuState

    def hot(t, i):
        return GpuState(hbm_frac=0.97 if i == 0 else 0.30)

    def mk():
        cfg = demo_config(server={"state_dir": str(tmp_path)}, periods={"history_step": 5.0})
        m = Monitor(cfg, gpu=GpuSource(FakeGpuReader(2, scenario=hot, clock=clock)), kube=FakeKube("steady", clock=clock), clock=clock)
        posted = []
        m.book.on_event = posted.append                    # what WebhookNotifier would be handed
        return m, posted

    m1, posted1 = mk()
    t_begin = clock()
    m1.tick_all()
    clock.advance(5)
    m1.tick_all()
    crit = m1.alerts_public()["critical"]
    assert [a["key"] for a in crit] == ["gpu/0/hbm"] and crit[0]["since"] == t_begin
    assert [(e["event"], e["key"]) for e in posted1] == [("begin", "gpu/0/hbm")]
    m1.stop()

    clock.advance(30)                                       # down for half a minute, the GPU is still full
    m2, posted2 = mk()
    assert m2.restored["alert_book"] is True
    m2.tick_all()
    clock.advance(5)
    m2.tick_all()
    crit = m2.alerts_public()["critical"]
    assert [a["key"] for a in crit] == ["gpu/0/hbm"] and crit[0]["since"] == t_begin     # began before the restart
    assert posted2 == []                                    # no second begin, no second webhook post
    assert [(e["event"], e["key"]) for e in m2.book.history] == [("begin", "gpu/0/hbm")]  # history survived
    m2.stop()

    # and an alert that ended while the monitor was down is closed on the first evaluation
    def cool(t, i):
        return GpuState(hbm_frac=0.30)
    cfg = demo_config(server={"state_dir": str(tmp_path)}, periods={"history_step": 5.0})
    m3 = Monitor(cfg, gpu=GpuSource(FakeGpuReader(2, scenario=cool, clock=clock)), kube=FakeKube("steady", clock=clock), clock=clock)
    m3.tick_all()
    assert m3.alerts_public()["critical"] == [] and [e["event"] for e in m3.book.history] == ["begin", "end"]
    m3.stop()


def test_gpu_alerts_do_not_wait_for_a_slow_kube_source(tmp_path, clock):
    """the first pods tick can block for the kubectl / API-server timeout; host and GPU rules (HBM, temperature,
    ECC) must fire meanwhile, and a restored pod alert must neither end nor be announced again before the pod list is in."""
    from k8s_llm_monitor_amd.fakes.gpu import GpuState

    m1 = make(clock, tmp_path, kube="pending")
    m1.tick_all()
    clock.advance(5)
    m1.tick_all()
    assert [a["key"] for a in m1.alerts_public()["serious"]] == ["pod/llm/vllm-llama3-405b-tp8-0/pending"]
    m1.stop()

    def hot(t, i):
        return GpuState(hbm_frac=0.97 if i == 0 else 0.30)

    cfg = demo_config(server={"state_dir": str(tmp_path)}, periods={"history_step": 5.0})
    m2 = Monitor(cfg, gpu=GpuSource(FakeGpuReader(2, scenario=hot, clock=clock)), kube=FakeKube("pending", clock=clock), clock=clock)
    events = []
    m2.book.on_event = events.append
    pods_before = dict(m2._alert_state.pods)
    assert pods_before and m2.get("t_pods") == 0.0
    m2.tick_realtime()                                      # the pods tick is still stuck in its first call
    assert [a["key"] for a in m2.alerts_public()["critical"]] == ["gpu/0/hbm"]          # not suppressed
    assert m2._alert_state.pods == pods_before              # the restored pod table is untouched
    assert [(e["event"], e["key"]) for e in events] == [("begin", "gpu/0/hbm")]          # the pod alert was not "ended"
    m2.tick_pods()                                          # ... and now the pod list arrives: same alert, same `since`, no new begin
    assert [a["key"] for a in m2.alerts_public()["serious"]] == ["pod/llm/vllm-llama3-405b-tp8-0/pending"]
    assert [(e["event"], e["key"]) for e in events] == [("begin", "gpu/0/hbm")]
    m2.stop()


def test_sse_is_told_when_an_alert_clears(clock):
    from k8s_llm_monitor_amd.fakes.gpu import GpuState
    level = {"hbm": 0.97}
    cfg = demo_config()
    m = Monitor(cfg, gpu=GpuSource(FakeGpuReader(1, scenario=lambda t, i: GpuState(hbm_frac=level["hbm"]), clock=clock)),
                kube=FakeKube("steady", clock=clock), clock=clock)
    pushed = []
    m.listeners.append(lambda kind, payload: pushed.append((kind, sorted(a["key"] for t in payload.values() for a in t))) if kind == "alerts" else None)
    m.tick_all()
    clock.advance(5)
    m.tick_all()                                            # unchanged: nothing pushed again
    level["hbm"] = 0.30
    clock.advance(5)
    m.tick_all()
    assert pushed == [("alerts", ["gpu/0/hbm"]), ("alerts", [])]
    m.stop()
