#!/usr/bin/env python3
"""Attention fwd+bwd at the C3 shape: a few launches for rocprofv3 --pmc passes, then event timings of the backward with the
single-pass kernel on and off (interleaved, same process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
B, N, H, dh = int(os.environ.get("B", 512)), int(os.environ.get("N", 50)), 8, 64
qkv = torch.randn(B, N, 3 * H * dh, device="cuda")
dout = torch.randn(B, N, H * dh, device="cuda")
for _ in range(4):
    out, lse = F.op_attention_fwd(qkv, H, dh)
    dq = F.op_attention_bwd(qkv, out, dout, lse, H, dh)
torch.cuda.synchronize()
if os.environ.get("TIME", "1") == "1":
    def t(on, reps=50):
        lib.dgvit_set_attention_bwd_single_pass(on)
        for _ in range(5):
            F.op_attention_bwd(qkv, out, dout, lse, H, dh)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            F.op_attention_bwd(qkv, out, dout, lse, H, dh)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    for rnd in range(3):
        print(f"round {rnd}: two-phase {t(0):.1f} us   single-pass {t(1):.1f} us", flush=True)
    lib.dgvit_set_attention_bwd_single_pass(1)
    def tf(reps=50):
        for _ in range(5):
            F.op_attention_fwd(qkv, H, dh)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            F.op_attention_fwd(qkv, H, dh)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    print(f"forward {tf():.1f} us  {tf():.1f} us", flush=True)
