#!/usr/bin/env python3
"""Launch one GEMM shape a few times (for rocprofv3 --pmc passes). usage: gemm_one.py layout epi M N K hint [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
layout, epi, m, n, k, hint = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = "cuda"
if layout == 0:
    A, B = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
elif layout == 1:
    A, B = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)
else:
    A, B = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
aux = torch.randn(m, n, device=dev) if epi in (2, 4) else None
lib.dgvit_set_gemm_tile(hint)
for _ in range(reps):
    F.op_gemm(layout, epi, A, B, m, n, k, aux=aux, want_c2=(epi == 1))
torch.cuda.synchronize()
