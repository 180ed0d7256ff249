"""Host-side cost of the serving engine's per-step control message between tensor-parallel ranks (gloo, CPU): the round-2 form
(all_gather_object of the whole scheduler digest: a pickle of one tuple per running request, two collectives) next to the
round-3 form (one all-gather of four int64). Run: torchrun --nproc-per-node 2 scripts/exp/tp_sync_cost.py [out.json]"""
import json, os, sys, time
import torch
import torch.distributed as dist

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
out = {}
for running in (1, 16, 64, 256):
    digest = (0, (), tuple((i, 100 + i) for i in range(running)), 1234, 20000)
    for name in ("all_gather_object_digest", "all_gather_4_int64"):
        def once():
            if name == "all_gather_object_digest":
                box = [None] * world
                dist.all_gather_object(box, (False, digest, (False, []) if rank == 0 else None))
                return [m[1] != box[0][1] for m in box]
            t = torch.tensor([0, hash(digest) & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64)
            box = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(box, t)
            return [b.tolist()[1] != box[0].tolist()[1] for b in box]
        for _ in range(50):
            once()
        dist.barrier()
        t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            once()
        out.setdefault(str(running), {})[name + "_us"] = round((time.perf_counter() - t0) / n * 1e6, 1)
if rank == 0:
    doc = {"ranks": world, "backend": "gloo (CPU, loopback)", "us_per_step_by_running_requests": out,
           "note": "per engine step, on the critical path of every decode launch"}
    print(json.dumps(doc, indent=1))
    if len(sys.argv) > 1:
        json.dump(doc, open(sys.argv[1], "w"), indent=1)
dist.barrier()
dist.destroy_process_group()
