#!/usr/bin/env python3
"""fp32 GEMM on square problems per tile choice, random and all-zero operands (clock / power check).  GPU box only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F

lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
dev = "cuda"
HINTS = [128128032, 128128016, 64064032, 64128016]
for n in (4096, 8192):
    for zeros in (0, 1):
        A = torch.zeros(n, n, device=dev) if zeros else torch.randn(n, n, device=dev)
        B = torch.zeros(n, n, device=dev) if zeros else torch.randn(n, n, device=dev)
        row = []
        for layout in (0, 1, 2):
            for hint in HINTS:
                lib.dgvit_set_gemm_tile(hint)
                ts = []
                for rep in range(4):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(3):
                        F.op_gemm(layout, 0, A, B, n, n, n)
                    e.record()
                    torch.cuda.synchronize()
                    if rep:
                        ts.append(s.elapsed_time(e) / 3)
                lib.dgvit_set_gemm_tile(0)
                ms = sorted(ts)[len(ts) // 2]
                row.append(f"L{layout}/{hint}: {2.0 * n ** 3 / ms / 1e9:6.1f}")
        print(f"n={n} {'zeros ' if zeros else 'random'} | " + " | ".join(row), flush=True)
