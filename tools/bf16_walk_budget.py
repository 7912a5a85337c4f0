#!/usr/bin/env python3
"""Stream GEMM on the config-5 shapes (B = 440: T = 86680 rows): the tile walk's L2 budget (column blocks), interleaved in
one process.  budget 0 = the round-3 walk (groups of 8 row panels, column by column).

    python tools/bf16_walk_budget.py [batch] [budgets_kb comma list]          timing sweep (median of 5 rounds x 5 launches)
    python tools/bf16_walk_budget.py pmc <budget_kb> [batch] [reps]          `reps` launches per shape with one budget, for rocprofv3 --pmc
                                                                              (tools/pmc_walk.py groups the dispatches by shape)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402

lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
pmc = len(sys.argv) > 1 and sys.argv[1] == "pmc"
if pmc:
    budget = int(sys.argv[2])
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 440
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
else:
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 440
    budgets = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1024,1600,2048,2560,3072").split(",")]
T, D, I, M = B * 197, 768, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
shapes = {"qkv": (T, 3 * I, D, 0), "out": (T, D, I, 0), "fc1": (T, M, D, 1), "fc2": (T, D, M, 0)}
for name, (m, n, k, epi) in shapes.items():
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g)
    if pmc:
        lib.dgvit_set_gemm_bf16_l2_budget_kb(budget)
        for _ in range(reps):
            F.op_gemm_bf16(epi, x, w, bias=bias)
        torch.cuda.synchronize()
        continue
    res = {b: [] for b in budgets}
    for rnd in range(5):
        for b in budgets:
            lib.dgvit_set_gemm_bf16_l2_budget_kb(b)
            for _ in range(2):
                F.op_gemm_bf16(epi, x, w, bias=bias)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                F.op_gemm_bf16(epi, x, w, bias=bias)
            e.record()
            torch.cuda.synchronize()
            res[b].append(s.elapsed_time(e) / 5)
    print(name, (m, n, k), " ".join(f"L2={b}KB: {sorted(v)[len(v)//2]*1e3:6.1f}us {2.0*m*n*k/sorted(v)[len(v)//2]/1e9:5.0f}TF" for b, v in res.items()), flush=True)
lib.dgvit_set_gemm_bf16_l2_budget_kb(2048)
print("done")
