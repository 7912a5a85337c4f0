#!/usr/bin/env python3
"""Latency of the fused SAC heads and the fused tanh-Gaussian sample at the reference's shipped sizes (HIP events, median)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
dgvit_amd.load_library()
dev = "cuda"


def lin(i, o):
    return torch.nn.Linear(i, o).to(dev)


def timeit(fn, n=30):
    ts = []
    for r in range(6):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        if r: ts.append(s.elapsed_time(e) / n * 1e3)
    ts.sort(); return ts[len(ts) // 2]


for name, B, ks, n2, towers, heads3 in [("policy head D=64", 32, (64,), 128, 1, 2), ("policy head D=64", 1, (64,), 128, 1, 2),
                                        ("policy head D=256", 512, (256,), 128, 1, 2), ("CNN twin-Q head", 32, (256, 32, 2), 32, 2, 1),
                                        ("GoT twin-Q head D=64", 32, (64, 2), 32, 2, 1)]:
    xs = [torch.randn(B, k, device=dev, requires_grad=True) for k in ks]
    mods = [(lin(sum(ks), 128), lin(128, n2), [lin(n2, 2) for _ in range(heads3)]) for _ in range(towers)]
    fwd = lambda: F.mlp_head(xs, mods)
    with torch.no_grad():
        tf = timeit(fwd)

    def fb():
        for x in xs: x.grad = None
        y = F.mlp_head(xs, mods)
        sum(v.sum() for r in y for v in r).backward()
    tb = timeit(fb, 10)
    print(f"{name:24s} B={B:4d} K0={sum(ks):4d}: forward {tf:7.1f} us   forward+backward (incl. autograd glue) {tb:7.1f} us", flush=True)
