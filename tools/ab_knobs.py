#!/usr/bin/env python3
"""Interleaved A/B of library knobs on the C3 training step in ONE process (boxes differ by several %, so A/B across gpurun
calls says nothing): python tools/ab_knobs.py grouped_reduce|wgrad_overlap|gemm_diagnostics [rounds] [value of the 1-arm]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd, synthetic
from dgvit_amd.optim import FlatAdam
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
knob = sys.argv[1] if len(sys.argv) > 1 else "grouped_reduce"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
on_value = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # value passed for the "1" arm (bit masks: gemm_diagnostics 8 = LDS-image epilogue)
if knob == "wgrad_overlap":      # a per-module schedule option now (dgvit_config.flags), not a library knob
    setter = lambda v: model.trans.set_schedule(wgrad_overlap=bool(v))
else:
    setter = getattr(lib, "dgvit_set_" + knob)
dev = torch.device("cuda")
torch.manual_seed(3407)
model = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)).to(dev).train()
opt = FlatAdam([model], lr=1e-4)
img, pstate, _, _ = (t.to(dev) for t in synthetic.make_inputs((84, 84), 512, 3407))
tm, tl = torch.randn(512, 2, device=dev), torch.randn(512, 2, device=dev)


def step():
    opt.zero_grad()
    mean, log_std = model([img, pstate])
    (torch.nn.functional.mse_loss(mean, tm) + torch.nn.functional.mse_loss(log_std, tl)).backward()
    opt.step()


res = {0: [], 1: []}
for r in range(rounds):
    for v in (0, 1):
        setter(on_value if v else 0)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 10 * 1e3)
for v in (0, 1):
    xs = sorted(res[v])
    print(f"{knob}={v}: median {xs[len(xs)//2]:.3f} ms  min {xs[0]:.3f}  all {[round(x, 3) for x in res[v]]}")
