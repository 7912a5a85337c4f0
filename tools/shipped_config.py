#!/usr/bin/env python3
"""The reference's shipped training configuration (config.yaml: LATENT 64, block 4, head 4, BATCH_SIZE 32, actor
GaussianTransformer, critic CNN) as one SAC learn() step (DRL.py:373-437) on the HIP modules.  GPU box only."""
import copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
import synthetic
from dgvit_amd.optim import FlatAdam, flatten_parameters, soft_update

dev = "cuda"
torch.manual_seed(3407)
for B in (32, 256):
    pol = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).to(dev)
    crt = dgvit_amd.QNetwork(2, 2).to(dev)
    tgt = copy.deepcopy(crt)
    flatten_parameters(crt), flatten_parameters(tgt)
    op, oc = FlatAdam([pol], lr=1e-3, capturable=True), FlatAdam([crt], lr=1e-3, capturable=True)
    img, ps, act, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 1))
    nimg, nps, _, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 2))
    rew = torch.randn(B, 1, device=dev)
    alpha, gamma, tau = 0.2, 0.99, 0.005

    def step():
        with torch.no_grad():
            na, nlogp, _ = pol.sample([nimg, nps])
            q1n, q2n = tgt([nimg, nps, na])
            y = rew + gamma * (torch.min(q1n, q2n) - alpha * nlogp)
        q1, q2 = crt([img, ps, act])
        qf = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
        oc.zero_grad(); qf.backward(); oc.step()
        pi, logp, _ = pol.sample([img, ps])
        q1p, q2p = crt([img, ps, pi])
        pl = (alpha * logp - torch.min(q1p, q2p)).mean()
        op.zero_grad(); oc.zero_grad(); pl.backward(); op.step()
        soft_update(tgt, crt, tau)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    # GPU-side time of the same step: events around the loop are the same as wall when the host keeps up;
    # host-bound shows as wall >> sum of kernels, measured here by timing the host enqueue only
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    g = dgvit_amd.GraphedStep(step, warmup=2)
    for _ in range(5):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g()
    torch.cuda.synchronize()
    dtg = (time.perf_counter() - t0) / n
    print(json.dumps({"config": f"same step replayed as one HIP graph, B={B}", "ms_per_step": round(dtg * 1e3, 3),
                      "frames_per_s": round(B / dtg, 1), "speedup_vs_eager": round(dt / dtg, 2)}), flush=True)
    print(json.dumps({"config": f"shipped SAC learn() step: GoT actor L4/H4/D64 + CNN critic, 128x160, B={B}",
                      "ms_per_step": round(dt * 1e3, 3), "host_enqueue_ms": round(host * 1e3, 3), "frames_per_s": round(B / dt, 1)}), flush=True)
