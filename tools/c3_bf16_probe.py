"""DGViT-small (config 3 shape: 84x84 @ 12, L6 H8 D256, B = 512) actor forward + backward + Adam with the encoder in the bf16
configuration (heads and optimiser fp32) next to the exact-fp32 path -- an OPTION for users, not the headline (whose parity
bar, 1e-3 on fp32 outputs, bf16 storage cannot meet)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
import synthetic  # noqa: E402
from dgvit_amd.optim import FlatAdam  # noqa: E402

B = 512
for dtype in (torch.float32, torch.bfloat16):
    torch.manual_seed(3407)
    m = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)).cuda().train()
    m.trans.set_compute_dtype(dtype)
    opt = FlatAdam([m], lr=1e-4)
    img, pstate, _, _ = (t.cuda() for t in synthetic.make_inputs((84, 84), B, 3407))
    tm, tl = torch.randn(B, 2).cuda(), torch.randn(B, 2).cuda()

    def step():
        opt.zero_grad()
        mean, log_std = m([img, pstate])
        loss = ((mean - tm) ** 2).mean() + ((log_std - tl) ** 2).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{str(dtype):16s} {dt * 1e3:7.3f} ms/step  {B / dt:9.1f} frames/s  loss {float(loss):.4f}", flush=True)
