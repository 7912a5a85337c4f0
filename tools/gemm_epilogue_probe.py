#!/usr/bin/env python3
"""What does the fused epilogue cost?  One shape (25600 x 2048 x 256, NT / NN, 64x128x16 tile), every epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.load_library()
dev = "cuda"
m, n, k = 25600, 2048, 256
A = torch.randn(m, k, device=dev)
Bt, Bn = torch.randn(n, k, device=dev), torch.randn(k, n, device=dev)
bias, aux, res = torch.randn(n, device=dev), torch.randn(m, n, device=dev), torch.randn(m, n, device=dev)


def t(fn):
    ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(8):
            fn()
        e.record(); torch.cuda.synchronize()
        if r: ts.append(s.elapsed_time(e) / 8)
    ts.sort(); return 2.0 * m * n * k / ts[len(ts) // 2] / 1e9


print("NT plain            ", round(t(lambda: F.op_gemm(0, 0, A, Bt, m, n, k)), 1))
print("NT bias             ", round(t(lambda: F.op_gemm(0, 0, A, Bt, m, n, k, bias=bias)), 1))
print("NT bias + residual  ", round(t(lambda: F.op_gemm(0, 0, A, Bt, m, n, k, bias=bias, res=res)), 1))
print("NT bias + relu      ", round(t(lambda: F.op_gemm(0, 3, A, Bt, m, n, k, bias=bias)), 1))
print("NT bias + gelu2     ", round(t(lambda: F.op_gemm(0, 1, A, Bt, m, n, k, bias=bias, want_c2=True)), 1))
print("NN plain            ", round(t(lambda: F.op_gemm(1, 0, A, Bn, m, n, k)), 1))
print("NN drelu (aux read) ", round(t(lambda: F.op_gemm(1, 4, A, Bn, m, n, k, aux=aux)), 1))
print("NN dgelu (aux read) ", round(t(lambda: F.op_gemm(1, 2, A, Bn, m, n, k, aux=aux)), 1))
