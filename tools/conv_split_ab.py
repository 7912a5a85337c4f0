"""CNN critic feature stack (conv1-3 + pool) forward at small batches: in-launch k-slices of the implicit-GEMM convolutions on / off
(diagnostic library), interleaved.  python tools/conv_split_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402

lib = dgvit_amd.diagnostic_library().__enter__()
g = torch.Generator().manual_seed(0)
shapes = [(16, 1, 5, 5), (16,), (64, 16, 5, 5), (64,), (256, 64, 5, 5), (256,)]
params = [(torch.randn(*s, generator=g) * 0.1).cuda() for s in shapes]


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


for B in (1, 8, 32, 64, 128):
    img = torch.rand(B, 128, 160, generator=g).cuda()
    with torch.no_grad():
        fn = dgvit_amd.GraphedStep(lambda: F.cnn_features(img, params), warmup=2)
        res = []
        for rep in range(2):
            for on in (1, 0):
                lib.dgvit_set_gemm_split(on)
                fn = dgvit_amd.GraphedStep(lambda: F.cnn_features(img, params), warmup=2)
                res.append((on, timeit(fn)))
        lib.dgvit_set_gemm_split(1)
    print(f"B={B:4d}  " + "  ".join(f"split={on}: {us:7.1f} us" for on, us in res), flush=True)
