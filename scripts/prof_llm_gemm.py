"""rocprofv3 workload: the serving engine's instantiations of the 256 x 256 ring GEMM at 256 rows -- K-split fp32 partials
(Llama-3-8B w_down: N 4096, K 14336) and the SwiGLU tile epilogue (Llama-3-70B gate_up: N 57344, K 8192) -- next to
hipBLASLt on the same shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_llm_monitor_amd.ops import llm
dev = torch.device("cuda:0")
it = int(os.environ.get("KLM_PROF_ITERS", "10"))
g = torch.Generator(device=dev); g.manual_seed(5)
def rnd(*s):
    return (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
x1, w1 = rnd(256, 14336), rnd(4096, 14336)
x2, w2 = rnd(256, 8192), rnd(57344, 8192)
w2t, w1t = w2.t(), w1.t()
for fn in (lambda: llm.mid_linear(x1, w1, kind="none"), lambda: torch.matmul(x1, w1t),
           lambda: llm.mid_linear(x2, w2, kind="swiglu"), lambda: torch.matmul(x2, w2t)):
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
print("done")
