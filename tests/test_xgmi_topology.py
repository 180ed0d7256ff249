"""xGMI link -> peer mapping from amdsmi's topology calls.

CPU tests drive ``read_xgmi_topology`` with a stub ``amdsmi`` module whose answers have the shapes captured
on a real MI355X (profiles/r2_amdsmi_topo_mi355x.json: eight link slots, the own slot ``X``, far ends that the
container cannot see reported as the all-ones PCI address, ``amdsmi_gpu_xgmi_error_status`` NOT_SUPPORTED)."""
import json
import os
import subprocess
import time
import types

import pytest

from k8s_llm_monitor_amd.collectors.gpu import (GpuNormalizer, GpuSource, _valid_bdf, read_xgmi_topology,
                                                 topology_to_raw)
from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader
from k8s_llm_monitor_amd.server.alerts import AlertState, check_alerts
from k8s_llm_monitor_amd.server.http import xgmi_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNSEEN = "ffffffffffff:ff:1f.7"


class H:
    """A processor handle as the bindings hand it out (ctypes.c_void_p-like)."""
    def __init__(self, v):
        self.value = v


def stub_amdsmi(n, bdfs, link_bdfs, hive=77, nearest=None, fail=()):
    """link_bdfs[i][slot] = far-end address the link table reports for GPU i."""
    m = types.SimpleNamespace()
    handles = [H(0x1000 + i) for i in range(n)]
    idx = {h.value: i for i, h in enumerate(handles)}
    m.AmdSmiLinkType = types.SimpleNamespace(AMDSMI_LINK_TYPE_XGMI=2)

    def guard(name):
        if name in fail:
            raise RuntimeError("AMDSMI_STATUS_NOT_SUPPORTED")
    def xgmi_info(h):
        guard("xgmi_info")
        return {"xgmi_lanes": 0, "xgmi_hive_id": hive, "xgmi_node_id": idx[h.value], "index": 0}
    def link_metrics(h):
        guard("link_metrics")
        i = idx[h.value]
        return {"num_links": 0, "links": [{"bdf": link_bdfs[i][s], "bit_rate": 38, "max_bandwidth": 608, "link_type": 2,
                                            "read": 0 if s == i else 1000 + s, "write": 0 if s == i else 2000 + s} for s in range(8)]}
    def nearest_fn(h, t):
        guard("nearest")
        return {"processor_list": [handles[j].value for j in (nearest[idx[h.value]] if nearest else [])]}
    def link_type(a, b):
        return {"hops": 1, "type": 2}
    m.amdsmi_get_xgmi_info, m.amdsmi_get_link_metrics = xgmi_info, link_metrics
    m.amdsmi_get_link_topology_nearest, m.amdsmi_topo_get_link_type = nearest_fn, link_type
    return m, handles


def test_unseen_address_is_not_a_peer():
    assert not _valid_bdf(UNSEEN) and not _valid_bdf("") and not _valid_bdf(None) and not _valid_bdf("N/A")
    assert _valid_bdf("0000:8b:00.0")


def test_peers_named_by_the_link_table_on_a_full_hive():
    bdfs = [f"0000:{0x10 + i:02x}:00.0" for i in range(8)]
    # slot order deliberately NOT the GPU order: slot s of GPU i ends at GPU (i + s) % 8
    table = [[UNSEEN if s == 0 else bdfs[(i + s) % 8] for s in range(8)] for i in range(8)]
    smi, hs = stub_amdsmi(8, bdfs, table, nearest=[[j for j in range(8) if j != i] for i in range(8)])
    topo = read_xgmi_topology(smi, hs, bdfs)
    assert len(topo) == 8
    for i, t in enumerate(topo):
        assert [l["peer_index"] for l in t["links"]] == [None] + [(i + s) % 8 for s in range(1, 8)]
        assert {l["source"] for l in t["links"][1:]} == {"link_metrics"} and t["links"][0]["source"] is None
        assert t["note"] is None and t["hive_id"] == 77 and t["node_id"] == i and t["nearest"] == [j for j in range(8) if j != i]
        assert t["pair_type"][(i + 1) % 8] == {"hops": 1, "type": 2} and all(l["max_gbps"] == 608 for l in t["links"])


def test_peers_inferred_from_node_ids_when_the_table_names_nobody():
    bdfs = [f"0000:{0x20 + i:02x}:00.0" for i in range(4)]
    smi, hs = stub_amdsmi(4, bdfs, [[UNSEEN] * 8 for _ in range(4)])
    topo = read_xgmi_topology(smi, hs, bdfs)
    for i, t in enumerate(topo):
        got = [l["peer_index"] for l in t["links"]]
        assert got[:4] == [None if s == i else s for s in range(4)] and got[4:] == [None] * 4     # slots 4..7 end outside this process
        assert {l["source"] for l in t["links"] if l["peer_index"] is not None} == {"xgmi_node_id"}
        assert t["links"][(i + 1) % 4]["peer_bdf"] == bdfs[(i + 1) % 4] and t["note"] is None


def test_single_visible_gpu_reports_why_peers_are_unnamed():
    """What the one-GPU gpurun box returns (profiles/r2_amdsmi_topo_mi355x.json)."""
    with open(os.path.join(ROOT, "profiles", "r2_amdsmi_topo_mi355x.json")) as f:
        real = json.load(f)["gpu0"]
    assert all(l["bdf"] == UNSEEN for l in real["link_metrics_a"]["links"]) and real["xgmi_link_status"]["status"][0] == "X"
    assert "NOT_SUPPORTED" in real["xgmi_error_status"] and real["nearest_AMDSMI_LINK_TYPE_XGMI"] == {"processor_list": []}
    smi, hs = stub_amdsmi(1, [real["bdf"]], [[l["bdf"] for l in real["link_metrics_a"]["links"]]])
    t = read_xgmi_topology(smi, hs, [real["bdf"]])[0]
    assert all(l["peer_index"] is None and l["peer_bdf"] is None for l in t["links"]) and len(t["links"]) == 8
    assert "not visible" in t["note"]
    raw = topology_to_raw(t)
    assert raw["xgmi_peers"] == [None] * 8 and raw["xgmi_link_max_gbps"] == [608.0] * 8


def test_failing_topology_calls_leave_a_reason_not_an_exception():
    smi, hs = stub_amdsmi(2, ["0000:01:00.0", "0000:02:00.0"], [[UNSEEN] * 8] * 2, fail=("xgmi_info", "link_metrics", "nearest"))
    topo = read_xgmi_topology(smi, hs, ["0000:01:00.0", "0000:02:00.0"])
    assert all(t["links"] == [] and "xgmi_info" in t["note"] and "link_metrics" in t["note"] for t in topo)


def test_normaliser_api_and_matrix_carry_the_peers(clock):
    src = GpuSource(FakeGpuReader(8, scenario="allreduce", clock=clock))
    src.sample()
    clock.advance(5.0)
    gpus = src.sample()
    g3 = gpus[3]
    assert g3["xgmi"]["peersKnown"] and g3["xgmi"]["nodeId"] == 3 and isinstance(g3["xgmi"]["hiveId"], str) and g3["xgmi"]["errorStatus"] == "no_errors"
    assert [l["peerIndex"] for l in g3["xgmi"]["links"]] == [0, 1, 2, 4, 5, 6, 7]
    assert all(l["peer"] == f"0000:{0x10 + l['peerIndex']:02x}:00.0" and l["peerSource"] == "link_metrics" and l["maxGBps"] == 76.0
               for l in g3["xgmi"]["links"])
    m = xgmi_matrix(gpus)
    assert m["peersKnown"] and m["indices"] == list(range(8)) and all(m["matrix"][i][i] is None for i in range(8))
    assert all(m["matrix"][i][j]["state"] == "up" and m["matrix"][i][j]["txGBps"] > 0 for i in range(8) for j in range(8) if i != j)
    assert m["linkImbalance"] == pytest.approx(1.0, abs=0.05)


def test_link_down_alert_names_the_peer(clock, demo_cfg):
    src = GpuSource(FakeGpuReader(8, scenario="xgmi_down", clock=clock))
    gpus = None
    for _ in range(30):
        clock.advance(10.0)
        gpus = src.sample()
        if any(l["state"] == "down" for g in gpus for l in g["xgmi"]["links"]):
            break
    down = [(g["index"], l) for g in gpus for l in g["xgmi"]["links"] if l["state"] == "down"]
    assert down, "the scenario takes a link down"
    alerts, _ = check_alerts({"cpu": {}, "memory": {}, "disk": {}}, gpus, [], AlertState(), demo_cfg)
    x = [a for tier in alerts.values() for a in tier if a.get("signal") == "xgmi"]
    assert x and all("-> 0000:" in a["desc"] for a in x)
    gi, l = down[0]
    cellm = xgmi_matrix(gpus)["matrix"][gi][l["peerIndex"]]
    assert cellm["state"] == "down" and cellm["link"] == l["link"]


def test_native_sampler_selftest_resolves_peers():
    from k8s_llm_monitor_amd.build import artifact
    exe = artifact("klm_amdsmi_sampler")
    if not exe.exists():
        pytest.skip("native sampler not built")
    r = subprocess.run([str(exe), "--selftest"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 0
    g0, g1 = json.loads(r.stdout)["gpus"]
    assert g0["xgmi_peer_index"] == [None, 1] + [None] * 6 and g1["xgmi_peer_index"] == [0] + [None] * 7
    assert g0["xgmi_peers"][1] == g1["bdf"] and g0["xgmi_peer_source"][1] == "link_metrics" and g0["xgmi_error_status"] == "not_supported"
    assert g0["xgmi_hive_id"] == 6319504153631384352 and g1["xgmi_node_id"] == 1 and g0["xgmi_link_max_gbps"] == [608] * 8
    s = GpuNormalizer().normalize(g0)
    assert [l["peerIndex"] for l in s["xgmi"]["links"]] == [1] + [None] * 6 and s["xgmi"]["peersKnown"]   # own slot dropped


@pytest.mark.gpu
def test_real_topology_names_peers_or_says_why():
    """On real hardware every link slot either names a visible GPU or the record says why none does."""
    from k8s_llm_monitor_amd.build import artifact
    from k8s_llm_monitor_amd.collectors.gpu import open_gpu_source
    src = open_gpu_source("amdsmi")
    try:
        topo = src.reader.topology
        assert len(topo) >= 1
        for t in topo:
            assert len(t["links"]) == 8 and all(l["max_gbps"] and l["max_gbps"] >= 100 for l in t["links"])
            named = [l for l in t["links"] if l["peer_index"] is not None]
            assert named or (t["note"] and ("not visible" in t["note"] or "no link slot" in t["note"])), t
            assert t["hive_id"] is not None and t["node_id"] is not None
        src.sample()
        time.sleep(1.0)
        g = src.sample()[0]
        x = g["xgmi"]
        assert x["errorStatus"] in ("not_supported", "no_errors", "error", "multiple_errors")
        assert len(x["links"]) >= 7 and all("peer" in l and "peerIndex" in l and l["maxGBps"] >= 12.5 for l in x["links"])
        if len(topo) == 1:
            assert x["peersKnown"] is False and "not visible" in x["topologyNote"]
        # the native sampler resolves the same thing
        r = subprocess.run([str(artifact("klm_amdsmi_sampler")), "--once"], capture_output=True, text=True, timeout=30)
        raw = json.loads(r.stdout.strip().splitlines()[-1])["gpus"][0]
        assert raw["xgmi_peer_index"] == [l["peer_index"] for l in topo[0]["links"]]
        assert raw["xgmi_hive_id"] == topo[0]["hive_id"] and raw["xgmi_error_status"] == x["errorStatus"]
    finally:
        src.close()


def test_counter_unit_check_tells_kib_from_kb():
    from k8s_llm_monitor_amd.validation.phases import xgmi_unit_check
    want = 7.0 / 4 * (1 << 28) * 40
    assert xgmi_unit_check(want, want)["verdict"] == "KiB (x1024)"                       # dashboard assumption holds
    assert xgmi_unit_check(want * 1.024, want)["verdict"] == "KB (x1000)"                # counts were 1000-byte units
    assert "neither" in xgmi_unit_check(want * 2.0, want)["verdict"] and xgmi_unit_check(0, want)["verdict"].startswith("no traffic")
