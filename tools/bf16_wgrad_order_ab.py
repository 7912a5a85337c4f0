"""bf16 weight-gradient GEMMs of config 5 (dW = dY^T X over T = 86 680 token rows, ring kernel, split-K): virtual tiles k-slice major
against tile major (diagnostic library, interleaved).  python tools/bf16_wgrad_order_ab.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 440
lib = dgvit_amd.diagnostic_library().__enter__()
T = B * 197
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


for name, (mo, ko) in {"qkv": (2304, 768), "to_out": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}.items():
    dy = torch.randn(T, mo, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(T, ko, device="cuda", generator=g).to(torch.bfloat16)
    res, outs = [], {}
    for rep in range(3):
        for on in (1, 0):
            lib.dgvit_set_gemm_wgrad_slice_major(on)
            outs[on] = F.op_wgrad_bf16(dy, x)
            res.append((on, timeit(lambda: F.op_wgrad_bf16(dy, x))))
    lib.dgvit_set_gemm_wgrad_slice_major(1)
    same = all(torch.equal(a, b) for a, b in zip(outs[1], outs[0]) if a is not None)
    print(f"{name:7s} dW ({mo}, {ko}) over {T} rows: " + "  ".join(f"slice-major={on}: {us:7.1f} us" for on, us in res) + f"  identical={same}", flush=True)
