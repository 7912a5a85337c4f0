"""Is the bf16 ring GEMM held back by the clock (DVFS)?  Same launch on random vs all-zero operands (zeros draw far less MFMA
power: MI355X_MICROARCH.md 'DVFS give-back').  python tools/bf16_power_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402

dgvit_amd.load_library()


def timeit(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


for (m, n, k) in [(50432, 2304, 768), (50432, 768, 3072), (8192, 8192, 8192)]:
    for name, gen in (("random", lambda *s: torch.randn(*s, device="cuda")), ("zeros", lambda *s: torch.zeros(*s, device="cuda"))):
        x, w = gen(m, k).to(torch.bfloat16), gen(n, k).to(torch.bfloat16)
        ms = timeit(lambda: F.op_gemm_bf16(0, x, w))
        print(f"{m}x{n}x{k} {name:7s} {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s", flush=True)
