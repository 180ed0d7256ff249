import ctypes, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libdir = os.path.join(root, "k8s_llm_monitor_amd", "_native", "lib")
names = sys.argv[1:] or ["base", "skipb1", "skipa1", "skipboth"]
libs = {}
for n in names:
    l = ctypes.CDLL(os.path.join(libdir, f"libklm_exp_{n}.so"))
    l.klm_gemm_bf16.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    libs[n] = l
dev = torch.device("cuda:0"); N = 8192
a = (torch.rand(N, N, device=dev) * 2 - 1).to(torch.bfloat16); b = (torch.rand(N, N, device=dev) * 2 - 1).to(torch.bfloat16)
c = torch.empty(N, N, device=dev, dtype=torch.bfloat16)
variant = int(os.environ.get("KLM_VARIANT", "1"))
def run(l, iters):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(iters):
        rc = l.klm_gemm_bf16(a.data_ptr(), b.data_ptr(), c.data_ptr(), N, N, N, variant, s)
        assert rc == 0, rc
for l in libs.values(): run(l, 50)
torch.cuda.synchronize()
res = {n: [] for n in names}
for rep in range(6):                      # interleaved A/B in one process
    for n in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(libs[n], 20); e0.record(); run(libs[n], 200); e1.record(); torch.cuda.synchronize()
        res[n].append(2.0 * N ** 3 * 200 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
for n in names:
    v = sorted(res[n]); print(f"{n:10s} median {v[len(v)//2]:.1f} TF  min {v[0]:.1f} max {v[-1]:.1f}", flush=True)
