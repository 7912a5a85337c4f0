import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()
T = 440 * 197
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters
for name, (m, n, k, epi) in {"qkv": (T, 2304, 768, 0), "out": (T, 768, 768, 0), "fc1": (T, 3072, 768, 1), "fc2": (T, 768, 3072, 0)}.items():
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g)
    lib.dgvit_set_gemm_bf16_tile(256257)
    timeit(lambda: F.op_gemm_bf16(epi, x, w, bias=bias))
    for gm in (8, 4):
        for ph in (8, 1, 2, 4, 16, 8, 1):
            lib.dgvit_set_gemm_bf16_group_m(gm + 1000 * ph)
            ms = timeit(lambda: F.op_gemm_bf16(epi, x, w, bias=bias))
            print(f"{name} group_m {gm:3d} phases {ph:2d}  {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:8.1f} TF", flush=True)
    lib.dgvit_set_gemm_bf16_group_m(8)
