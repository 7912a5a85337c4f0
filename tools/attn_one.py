#!/usr/bin/env python3
"""Run attention fwd+bwd at the C3 shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
B, N, H, dh = 512, int(os.environ.get("N", 50)), 8, 64
qkv = torch.randn(B, N, 3 * H * dh, device="cuda")
dout = torch.randn(B, N, H * dh, device="cuda")
for _ in range(4):
    out, lse = F.op_attention_fwd(qkv, H, dh)
    dq = F.op_attention_bwd(qkv, out, dout, lse, H, dh)
torch.cuda.synchronize()
