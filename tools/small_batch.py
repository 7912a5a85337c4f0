#!/usr/bin/env python3
"""The reference's launch-bound regime (SURVEY 7 hard part 8): single-frame policy.sample() (SAC.choose_action, DRL.py:170-185) of
the shipped model (config.yaml: L4 / H4 / D64, 128x160), eager and replayed as one HIP graph.  `python tools/small_batch.py trace`
runs 20 eager calls only (for rocprofv3 --kernel-trace)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
import synthetic
dev = "cuda"
torch.manual_seed(0)
m = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).to(dev).eval()
out = {}
for B in (1, 2, 32):
    img, ps, _, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), B, 0))

    def call():
        with torch.no_grad():
            return m.sample([img, ps])
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        if B == 1:
            for _ in range(20):
                call()
            torch.cuda.synchronize()
        continue
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        call()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200
    g = dgvit_amd.GraphedStep(call, warmup=3)
    for _ in range(10):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 500
    out[f"B={B}"] = {"eager_ms": round(eager * 1e3, 4), "graph_ms": round(graphed * 1e3, 4)}
if out:
    print(json.dumps({"config": "policy.sample(), shipped GoT actor L4/H4/D64, 128x160 frames (no_grad, eval)", **out}))
