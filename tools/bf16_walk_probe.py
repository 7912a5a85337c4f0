#!/usr/bin/env python3
"""bf16 ring GEMM on the config-5 shapes (B = 440: T = 86680 rows) for several walk-group heights, interleaved in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 440
T, D, I, M = B * 197, 768, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
shapes = {"qkv": (T, 3 * I, D, 0), "out": (T, D, I, 0), "fc1": (T, M, D, 1), "fc2": (T, D, M, 0)}
groups = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,4,8,16,32".split(","))]
for name, (m, n, k, epi) in shapes.items():
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g)
    res = {gm: [] for gm in groups}
    for rnd in range(4):
        for gm in groups:
            lib.dgvit_set_gemm_bf16_group_m(gm)
            for _ in range(2):
                F.op_gemm_bf16(epi, x, w, bias=bias)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                F.op_gemm_bf16(epi, x, w, bias=bias)
            e.record(); torch.cuda.synchronize()
            res[gm].append(s.elapsed_time(e) / 5)
    lib.dgvit_set_gemm_bf16_group_m(8)
    print(name, (m, n, k), " ".join(f"G{gm}: {2.0*m*n*k/sorted(v)[len(v)//2]/1e9:6.0f}TF" for gm, v in res.items()), flush=True)
