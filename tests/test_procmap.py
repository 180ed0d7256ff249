"""GPU process -> pod attribution (which pod holds the HBM)."""
from k8s_llm_monitor_amd.collectors.gpu import GpuSource
from k8s_llm_monitor_amd.collectors.procmap import PodResolver, attribute, pod_uid_from_cgroup
from k8s_llm_monitor_amd.config import demo_config
from k8s_llm_monitor_amd.fakes.gpu import FakeGpuReader
from k8s_llm_monitor_amd.fakes.kube import FakeKube, fake_uid
from k8s_llm_monitor_amd.server.sampler import Monitor

UID = "0f3a9c1e-7b2d-4e5f-8a6b-1c2d3e4f5a6b"
CGROUPFS = f"0::/kubepods/burstable/pod{UID}/3f5c2d8a9b\n"
SYSTEMD = ("0::/kubepods.slice/kubepods-burstable.slice/kubepods-burstable-pod" + UID.replace("-", "_") +
           ".slice/cri-containerd-abc123.scope\n")
V1 = f"12:memory:/kubepods/pod{UID}/abc\n11:cpu,cpuacct:/kubepods/pod{UID}/abc\n"


def test_pod_uid_from_cgroup_forms():
    assert pod_uid_from_cgroup(CGROUPFS) == UID and pod_uid_from_cgroup(SYSTEMD) == UID and pod_uid_from_cgroup(V1) == UID
    assert pod_uid_from_cgroup("0::/user.slice/user-1000.slice/session-3.scope\n") is None and pod_uid_from_cgroup("") is None


def test_attribute_sums_per_pod_and_caches():
    reads = []

    def read(pid):
        reads.append(pid)
        if pid == 99:
            raise FileNotFoundError(pid)
        return CGROUPFS if pid in (10, 11) else "0::/system.slice/sshd.service\n"

    r = PodResolver(read)
    pods = [{"namespace": "llm", "name": "vllm-0", "uid": UID.upper()}]
    gpus = [{"index": 0, "processList": [{"pid": 10, "name": "python3", "vramBytes": 100}, {"pid": 11, "name": "python3", "vramBytes": 50},
                                         {"pid": 12, "name": "rocm-smi", "vramBytes": 7}, {"pid": 99, "name": "gone", "vramBytes": 1}]},
            {"index": 1, "processList": None}]
    attribute(gpus, pods, r)
    assert gpus[0]["podsHbm"][0] == {"owner": "llm/vllm-0", "vramBytes": 150}
    assert {x["owner"] for x in gpus[0]["podsHbm"][1:]} == {"host:rocm-smi", "host:gone"}
    assert gpus[0]["processList"][0]["pod"] == "llm/vllm-0" and gpus[0]["processList"][2]["pod"] is None
    assert "podsHbm" not in gpus[1]
    attribute(gpus, pods, r)
    assert reads.count(10) == 1                      # cached
    attribute([{"index": 0, "processList": [{"pid": 10, "name": "p", "vramBytes": 1}]}], [], r)
    assert 11 not in r._cache                        # dead pids are forgotten
    assert attribute([{"index": 0, "processList": [{"pid": 10, "vramBytes": 1}]}], [], r)[0]["podsHbm"][0]["owner"] == f"pod:{UID}"


def test_monitor_joins_processes_with_pods(clock):
    cfg = demo_config(sources={"gpu_processes": True})
    mon = Monitor(cfg, gpu=GpuSource(FakeGpuReader(2, clock=clock)), kube=FakeKube(clock=clock), clock=clock)
    uid = fake_uid("llm", "vllm-llama3-70b-tp8-0")
    mon.pod_resolver = PodResolver(lambda pid: f"0::/kubepods/besteffort/pod{uid}/x\n")
    mon.tick_all()
    g = mon.get("gpus")[1]
    assert g["podsHbm"][0]["owner"] == "llm/vllm-llama3-70b-tp8-0" and g["podsHbm"][0]["vramBytes"] > 100 << 30
    mon.stop()
