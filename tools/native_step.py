#!/usr/bin/env python3
"""fwd+bwd of DGViT-small at the reference's native 128x160 @ 16x20 (N = 65 tokens), B = 256 (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
import synthetic
dev = "cuda"
torch.manual_seed(0)
B = int(os.environ.get("B", 256))
m = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256).to(dev).train()
img, ps, _, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 0))
for _ in range(6):
    m.zero_grad(set_to_none=True)
    a, b = m([img, ps])
    ((a ** 2).mean() + (b ** 2).mean()).backward()
torch.cuda.synchronize()
