#!/usr/bin/env python3
"""In-CU efficiency probe: grids of exactly 1, 2, 3, 4 workgroups per CU (no tile-quantisation loss)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
dev = "cuda"
def timeit(layout, m, n, k, hint, reps=5, inner=5):
    A = torch.randn(m, k, device=dev)
    B = torch.randn(n, k, device=dev) if layout == 0 else torch.randn(k, n, device=dev)
    lib.dgvit_set_gemm_tile(hint)
    ts = []
    for r in range(reps + 1):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            F.op_gemm(layout, 0, A, B, m, n, k)
        e.record(); torch.cuda.synchronize()
        if r: ts.append(s.elapsed_time(e) / inner)
    lib.dgvit_set_gemm_tile(0)
    ts.sort(); return ts[len(ts)//2]
CFGS = [("128x128x32", 128128032, 128, 128), ("128x128x16", 128128016, 128, 128), ("64x64x32", 64064032, 64, 64),
        ("64x128x32", 64128032, 64, 128), ("64x128x16", 64128016, 64, 128), ("128x64x32", 128064032, 128, 64)]
for name, hint, bm, bn in CFGS:
    for K in (256, 2048):
        for per_cu in (2, 8):
            tiles = 256 * per_cu
            m, n = tiles * bm, bn
            ms = timeit(0, m, n, K, hint)
            tf = 2.0 * m * n * K / ms / 1e9
            print(f"tile {name:12s} K={K:4d} wg/CU={per_cu:2d} ({tiles} tiles)  {ms*1e3:8.1f} us  {tf:6.1f} TF  ({tf/157.3*100:4.1f}% of peak)", flush=True)
