"""Cluster view: two live monitors, one aggregating the other over HTTP."""
import json
import time
import urllib.request

from k8s_llm_monitor_amd.collectors.peers import PeerPoller, digest
from k8s_llm_monitor_amd.config import demo_config
from k8s_llm_monitor_amd.server.http import MonitorServer
from k8s_llm_monitor_amd.server.sampler import Monitor


def test_digest_of_a_summary():
    s = {"host": {"cpu": {"percent": "12.5"}, "memory": {"percent": "40.0"}, "disk": {}},
         "gpus": [{"index": 0, "memoryUsed": 100, "memoryTotal": 400, "utilization": 50, "gfxActivityAvg": 60.0, "temperature": 70, "powerW": 500.0},
                  {"index": 1, "memoryUsed": 300, "memoryTotal": 400, "utilization": 80, "gfxActivityAvg": None, "temperature": 90, "powerW": 700.0}],
         "pods": [{}, {}], "alerts": {"minor": [1], "serious": [], "critical": [1, 2]}, "vllm": {"available": True, "tokensPerSec": 123.0}}
    d = digest(s)
    assert d["gpus"] == 2 and d["hbmPercent"] == 50.0 and d["gfxActivityMean"] == 70.0 and d["temperatureMax"] == 90
    assert d["powerW"] == 1200.0 and d["cpuPercent"] == 12.5 and d["diskPercent"] is None and d["pods"] == 2
    assert d["alerts"] == {"minor": 1, "serious": 0, "critical": 2} and d["vllmTokensPerSec"] == 123.0
    assert digest({})["gpus"] == 0


def test_dead_peer_is_reported_not_raised():
    p = PeerPoller(["http://127.0.0.1:9"], timeout=0.2)
    p.poll()
    c = p.cluster({"gpus": [], "alerts": {}})
    assert c["totals"]["nodes"] == 2 and c["totals"]["nodesDown"] == 1 and c["nodes"][1]["ok"] is False and c["nodes"][1]["error"]


def test_two_monitors_one_cluster():
    cfg_b = demo_config(periods={"realtime": 0.2}, sources={"fake_gpus": 4, "fake_scenario": "hbm_ramp"})
    mon_b = Monitor(cfg_b).start()
    srv_b = MonitorServer(mon_b, cfg_b).start()
    cfg_a = demo_config(periods={"realtime": 0.2, "peers": 0.2}, sources={"peers": [srv_b.url, "http://127.0.0.1:9"]})
    mon_a = Monitor(cfg_a).start()
    srv_a = MonitorServer(mon_a, cfg_a).start()
    try:
        time.sleep(0.8)
        c = json.loads(urllib.request.urlopen(srv_a.url + "/api/cluster", timeout=3).read())
        assert [n["ok"] for n in c["nodes"]] == [True, True, False]
        assert c["nodes"][0]["gpus"] == 8 and c["nodes"][1]["gpus"] == 4 and c["totals"]["gpus"] == 12
        assert c["totals"]["hbmTotalMiB"] == 12 * 294896 and c["totals"]["nodesDown"] == 1
        assert len(c["nodes"][1]["perGpu"]) == 4 and c["nodes"][1]["url"] == srv_b.url
        assert "peers" in mon_a.tasks and "peers" not in mon_b.tasks
    finally:
        srv_a.stop(); mon_a.stop(); srv_b.stop(); mon_b.stop()
