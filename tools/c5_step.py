"""Config 5 (224x224 ViT-Base variant, bf16) steps for rocprofv3:  python tools/c5_step.py [fwd|train] [batch] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(5)
m = dgvit_amd.GoT(image_size=224, patch_size=16, num_classes=2, dim=768, depth=12, heads=12, mlp_dim=3072, channels=1)
m = m.cuda().set_compute_dtype(torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(0)
img, goal = torch.rand(B, 224, 224, device="cuda", generator=g), torch.randn(B, 768, device="cuda", generator=g)
tgt = torch.randn(B, 768, device="cuda", generator=g)
if mode == "fwd":
    m.eval()
    with torch.no_grad():
        for _ in range(steps + 2):
            f = m(img, goal)
else:
    m.train()
    for _ in range(steps + 1):
        for p in m.parameters():
            p.grad = None
        loss = ((m(img, goal) - tgt) ** 2).mean()
        loss.backward()
torch.cuda.synchronize()
print("done")
