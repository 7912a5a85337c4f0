"""One bf16 GEMM shape, a few launches (for rocprofv3 --pmc passes).  python tools/bf16_gemm_one.py M N K tile epi [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402

M, N, K, tile, epi = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
res = torch.randn(M, N, device="cuda", generator=g) if epi == 2 else None
lib.dgvit_set_gemm_bf16_tile(tile)
for _ in range(iters):
    y = F.op_gemm_bf16(epi, x, w, bias=bias, res=res)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
