#!/usr/bin/env python3
"""fp32 GEMM throughput against workgroups per CU (dynamic-LDS padding caps the residency): does a K = 256 shape still gain
from the 5th resident workgroup, i.e. is its per-tile prologue / epilogue hidden by co-resident workgroups or not?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
dev = "cuda"
CASES = [("qkv fwd NT 25600x1536x256 64x128x16", 0, 0, 25600, 1536, 256, 64128016, 30720),
         ("fc1 fwd NT 25600x2048x256 64x128x16 gelu2", 0, 1, 25600, 2048, 256, 64128016, 30720),
         ("fc2 fwd NT 24576x256x2048 64x64x32", 0, 0, 24576, 256, 2048, 64064032, 36864)]
for name, layout, epi, m, n, k, hint, lds in CASES:
    A = torch.randn(m, k, device=dev)
    B = torch.randn(n, k, device=dev)
    bias = torch.randn(n, device=dev)
    lib.dgvit_set_gemm_tile(hint)
    row = []
    for per_cu in (1, 2, 3, 4, 5):
        want = 160 * 1024 // per_cu                      # LDS per workgroup that leaves room for exactly per_cu workgroups
        pad = max(0, want - lds - 256) if per_cu < 5 else 0
        lib.dgvit_set_gemm_lds_pad(pad)
        ts = []
        for r in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                F.op_gemm(layout, epi, A, B, m, n, k, bias=bias, want_c2=(epi == 1))
            e.record(); torch.cuda.synchronize()
            if r: ts.append(s.elapsed_time(e) / 8)
        ts.sort()
        row.append(f"{per_cu}/CU {2.0*m*n*k/ts[len(ts)//2]/1e9:6.1f}TF")
    lib.dgvit_set_gemm_lds_pad(0); lib.dgvit_set_gemm_tile(0)
    print(name, " | ".join(row), flush=True)

print("tile variants on the K = 256 shapes:")
for name, layout, epi, m, n, k in [("qkv fwd", 0, 0, 25600, 1536, 256), ("fc1 fwd gelu2", 0, 1, 25600, 2048, 256), ("dfc2 dgelu", 1, 2, 25600, 2048, 256),
                                   ("dout NN", 1, 0, 25600, 512, 256), ("out fwd K=512", 0, 0, 25600, 256, 512)]:
    A = torch.randn(m, k, device=dev)
    B = torch.randn(n, k, device=dev) if layout == 0 else torch.randn(k, n, device=dev)
    bias = torch.randn(n, device=dev) if layout == 0 else None
    aux = torch.randn(m, n, device=dev) if epi == 2 else None
    row = []
    for hint in (64128016, 64064016, 64064032):
        lib.dgvit_set_gemm_tile(hint)
        ts = []
        for r in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                F.op_gemm(layout, epi, A, B, m, n, k, bias=bias, aux=aux, want_c2=(epi == 1))
            e.record(); torch.cuda.synchronize()
            if r: ts.append(s.elapsed_time(e) / 8)
        ts.sort()
        row.append(f"{hint} {2.0*m*n*k/ts[len(ts)//2]/1e9:6.1f}TF")
    lib.dgvit_set_gemm_tile(0)
    print(f"{name:16s}", " | ".join(row), flush=True)
