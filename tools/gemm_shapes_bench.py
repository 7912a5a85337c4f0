#!/usr/bin/env python3
"""Time every GEMM shape of one DGViT-small layer at B=512 (T = 25600 tokens) per tile choice.
GPU box only.  Prints TFLOP/s (algorithmic 2MNK / HIP-event time, median of 5 x 10 launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F

lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
T = int(os.environ.get("T", 25600))
D, I, M = 256, 512, 2048
dev = "cuda"
SHAPES = [  # name, layout, epi, M, N, K
    ("fwd qkv      NT", 0, 0, T, 3 * I, D), ("fwd out+res  NT", 0, 0, T, D, I), ("fwd fc1+gelu NT", 0, 1, T, M, D),
    ("fwd fc2+res  NT", 0, 0, T, D, M), ("dgrad fc2*g' NN", 1, 2, T, M, D), ("dgrad fc1    NN", 1, 0, T, D, M),
    ("dgrad out    NN", 1, 0, T, I, D), ("dgrad qkv    NN", 1, 0, T, D, 3 * I), ("wgrad fc2    TN", 2, 0, D, M, T),
    ("wgrad fc1    TN", 2, 0, M, D, T), ("wgrad out    TN", 2, 0, D, I, T), ("wgrad qkv    TN", 2, 0, 3 * I, D, T)]


def run(layout, epi, m, n, k, hint):
    if layout == 0:
        A, B = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    elif layout == 1:
        A, B = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)
    else:
        A, B = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
    bias = torch.randn(n, device=dev) if layout == 0 else None
    res = torch.randn(m, n, device=dev) if (layout == 0 and epi == 0 and n == D) else None
    aux = torch.randn(m, n, device=dev) if epi == 2 else None
    lib.dgvit_set_gemm_tile(hint)
    times = []
    try:
        for rep in range(6):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                F.op_gemm(layout, epi, A, B, m, n, k, bias=bias, res=res, aux=aux, want_c2=(epi == 1))
            e.record()
            torch.cuda.synchronize()
            if rep:
                times.append(s.elapsed_time(e) / 10)
    finally:
        lib.dgvit_set_gemm_tile(0)
    times.sort()
    return times[len(times) // 2]


HINTS = [128128032, 128128016, 64064032, 64064064, 128064032, 64128032, 64128016]
tot = {}
best_sum = 0.0
for name, layout, epi, m, n, k in SHAPES:
    row = []
    for hint in HINTS:
        ms = run(layout, epi, m, n, k, hint)
        tf = 2.0 * m * n * k / ms / 1e9
        row.append((ms, tf))
        tot.setdefault(hint, 0.0)
        tot[hint] += ms
    bi = min(range(len(row)), key=lambda i: row[i][0])
    best_sum += row[bi][0]
    print(f"{name} M={m:6d} N={n:5d} K={k:6d} | " + " | ".join(f"{h}:{r[0]*1e3:6.1f}us {r[1]:5.1f}TF" for h, r in zip(HINTS, row)) + f" | best {HINTS[bi]}", flush=True)
print("sum ms per tile choice:", {k: round(v, 3) for k, v in tot.items()}, "best-of sum", round(best_sum, 3))
