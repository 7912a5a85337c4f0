"""C3 forward only (GoTPolicy DGViT-small, B=512, 84x84, eval/no_grad) for rocprofv3:  python tools/c3_forward.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(3407)
m = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)).to(dev).eval()
img, pstate, _, _ = (t.to(dev) for t in synthetic.make_inputs((84, 84), 512, 3407))
with torch.no_grad():
    for _ in range(3):
        m([img, pstate])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m([img, pstate])
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"forward {dt * 1e3:.3f} ms  {512 / dt:.0f} frames/s  {512 / dt * synthetic.fwd_flops_per_frame((84, 84), (12, 12), 256, 6, 8) / 1e12:.1f} TFLOP/s")
