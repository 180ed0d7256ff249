#!/usr/bin/env python3
"""What ONE GPU of a tensor-parallel llama3-405b job does per decode step, measured on one MI355X.

The box has one GPU, so the other seven ranks are absent: this rank holds its real shard (1/8 of the heads, FFN columns
and vocabulary: about 50 GB of bf16 weights) and runs the real decode step, hipGraph and all, with every all-reduce /
all-gather launched on a ONE-rank mesh communicator (the kernel, its flag barrier and its launch are there; the peer
traffic and the wait for peers are not). The step time is therefore this GPU's compute + launch share of a TP step: a
lower bound of the real step, to which 2 x layers mesh all-reduces of [batch, 16384] bf16 add their xGMI latency.
Writes gpurun_out/tp_shard_bench.json.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_llm_monitor_amd.models.llama import PRESETS, LlamaModel  # noqa: E402
from k8s_llm_monitor_amd.parallel import xgmi  # noqa: E402
from k8s_llm_monitor_amd.parallel.tp import SingleGroup  # noqa: E402
from k8s_llm_monitor_amd.serving import Engine, EngineConfig  # noqa: E402
from k8s_llm_monitor_amd.serving import bench as sb  # noqa: E402


class AbsentPeers(SingleGroup):
    """Rank 0 of `world`, peers absent: collectives run the mesh kernels on a one-rank communicator."""

    def __init__(self, world: int, dev):
        self.rank, self.world, self.backend = 0, world, "xgmi (one-rank loopback)"
        self.comm = xgmi.XgmiComm(0, 1, dev, lambda b: [b])
        self.calls = 0

    def all_reduce(self, x):
        self.calls += 1
        return self.comm.all_reduce(x)

    def all_gather(self, x):
        return self.comm.all_gather(x)

    def describe(self):
        return {"world": self.world, "backend": self.backend}


CASES = ((1, None), (8, 4), (8, 16), (16, 4), (16, 16), (64, None))


def main() -> int:
    global CASES
    if len(sys.argv) > 4 and sys.argv[4] == "small-batch":
        CASES = ((1, 0), (1, 16), (2, 0), (2, 16), (4, 0), (4, 16))
    elif len(sys.argv) > 4 and sys.argv[4].startswith("batches="):
        CASES = tuple((int(b), None) for b in sys.argv[4][8:].split(","))
    name = sys.argv[1] if len(sys.argv) > 1 else "llama3-405b"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tag = f"{name}_tp{world}"
    weights = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tp = AbsentPeers(world, dev) if world > 1 else SingleGroup()
    t0 = time.time()
    model = LlamaModel(PRESETS[name], dev, tp=tp)
    if weights == "fp8":
        model.enable_fp8_decode()
    print(f"shard of {name} / tp{world}: {model.weight_bytes() / 1e9:.1f} GB in {time.time() - t0:.1f} s", flush=True)
    eng = Engine(EngineConfig(model=name, device="cuda:0", max_num_seqs=128, max_model_len=4096,
                              kv_blocks=int(os.environ.get("KLM_KV_BLOCKS", "20000"))), model=model)
    rows = []
    for b, lim in CASES:
        if lim != model.skinny_limit:
            model.skinny_limit = lim          # batch 8 / 16 on either path: library GEMMs (4) or the skinny kernel (16)
            eng._graphs.clear()
        r = sb.decode_step_ms(eng, b, 1024, iters=10)
        r["path"] = "skinny" if model.skinny(b) else ("mid" if model.mid(b) else "library")
        r["allreduce_launches_per_step"] = 2 * PRESETS[name].layers
        r["allreduce_bytes_each"] = eng.buckets[[x >= b for x in eng.buckets].index(True)] * PRESETS[name].hidden * 2
        rows.append(r)
        print(json.dumps(r), flush=True)
    out = {"model": name, "tp_world": world, "weights": weights, "shard_weight_bytes": model.weight_bytes(),
           "local_shapes": {"n_q": model.lcfg.n_q, "n_kv": model.lcfg.n_kv, "ffn": model.lcfg.ffn, "vocab": model.lcfg.vocab},
           "note": "one rank of the job, peers absent: compute + launches of a TP decode step, no xGMI wait", "decode": rows,
           "mesh_status": tp.comm.status() if world > 1 else 0}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/tp_shard_bench_{tag}_{weights}.json", "w") as f:
        json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
