#!/usr/bin/env python3
"""One shipped-configuration SAC step (B=32) repeated, eager, for rocprofv3 kernel statistics."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
import synthetic
from dgvit_amd.optim import FlatAdam, flatten_parameters, soft_update
dev = "cuda"
torch.manual_seed(3407)
B = 32
pol = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).to(dev)
crt = dgvit_amd.QNetwork(2, 2).to(dev)
tgt = copy.deepcopy(crt)
flatten_parameters(crt), flatten_parameters(tgt)
op, oc = FlatAdam([pol], lr=1e-3), FlatAdam([crt], lr=1e-3)
img, ps, act, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 1))
nimg, nps, _, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 2))
rew = torch.randn(B, 1, device=dev)
for _ in range(10):
    with torch.no_grad():
        na, nlogp, _ = pol.sample([nimg, nps])
        q1n, q2n = tgt([nimg, nps, na])
        y = rew + 0.99 * (torch.min(q1n, q2n) - 0.2 * nlogp)
    q1, q2 = crt([img, ps, act])
    qf = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
    oc.zero_grad(); qf.backward(); oc.step()
    pi, logp, _ = pol.sample([img, ps])
    q1p, q2p = crt([img, ps, pi])
    pl = (0.2 * logp - torch.min(q1p, q2p)).mean()
    op.zero_grad(); oc.zero_grad(); pl.backward(); op.step()
    soft_update(tgt, crt, 0.005)
torch.cuda.synchronize()
