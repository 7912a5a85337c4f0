#!/usr/bin/env python3
"""Phase timeline of the two small-batch block kernels (block.hip) from in-kernel wall-clock stamps (100 MHz: 10 ns steps) of workgroup 0,
diagnostic library: python tools/block_stamps.py [batch] [dim] [layer (-1: the last block)] [fuse bits (3)]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import dgvit_amd  # noqa: E402
import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = dgvit_amd.diagnostic_library().__enter__()
lib.dgvit_set_block_path(2, 4160)
layer = int(sys.argv[3]) if len(sys.argv) > 3 else -1
lib.dgvit_set_block_stamp_layer(layer)
lib.dgvit_set_block_fuse(int(sys.argv[4]) if len(sys.argv) > 4 else 3)
torch.manual_seed(0)
if D == 64:
    m, image = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64), (128, 160)
else:
    m, image = dgvit_amd.GoTPolicy(2, 2, 4, 4, D, image_size=(84, 84), patch_size=(12, 12)), (84, 84)
m = m.cuda().eval()
img, ps, _, _ = (t.cuda() for t in synthetic.make_inputs(image, B, 0))
goal = torch.zeros(B, D, device="cuda")
buf = torch.zeros(32, dtype=torch.int64, device="cuda")
names_a = ["start", "rows staged (+LN)", "q k v projected", "scores + max", "exp + PV", "output tile", "to_out share", "stored"]
names_m = ["start", "xmid + LN2 rows", "gelu(fc1 chunk)", "fc2 share", "published (drained)", "ticket", "combined + stored"]
acc_a, acc_m, n, comb = [0.0] * 8, [0.0] * 7, 0, 0.0
with torch.no_grad():
    for it in range(30):
        lib.dgvit_set_block_stamps(buf.data_ptr())
        m.trans(img, goal)
        torch.cuda.synchronize()
        v = buf.cpu().tolist()          # (the last block's kernels wrote last)
        if it >= 10:
            n += 1
            for i in range(8):
                acc_a[i] += (v[i] - v[0]) * 10.0
            for i in range(7):
                acc_m[i] += (v[16 + i] - v[16]) * 10.0
            comb = comb + (v[24] - v[23]) * 10.0
lib.dgvit_set_block_stamps(None)
print(f"attn_block_kernel, workgroup 0, B = {B}, D = {D}, block {layer if layer >= 0 else 'last (token-0 query tile only)'}, fuse bits {sys.argv[4] if len(sys.argv) > 4 else 3} (ns since its start, mean of {n})")
for nm, t in zip(names_a, acc_a):
    print(f"  {t / n:9.0f}  {nm}")
print("mlp_block_kernel, workgroup 0 (its 'combined' stamp is only meaningful when workgroup 0 arrived last)")
for nm, t in zip(names_m[:6], acc_m):
    print(f"  {t / n:9.0f}  {nm}")
print(f"  {comb / n:9.0f}  ns: the last arriver's combine (row tile 0: partial reads, bias, residual, LayerNorm, stores)")
