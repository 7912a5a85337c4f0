#!/usr/bin/env python3
"""C3 training step (B=512 actor fwd+bwd + Adam) eager vs replayed as one HIP graph, with and without the helper stream.
GPU box only.  Answers: how much of the step is launch gaps?"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import dgvit_amd
import synthetic
from dgvit_amd.optim import FlatAdam

dev = torch.device("cuda", 0)
lib = dgvit_amd.load_library()
B = 512
out = {}
for overlap in (0, 1):
    torch.manual_seed(3407)
    model = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)).to(dev).train()
    model.trans.set_schedule(wgrad_overlap=bool(overlap))
    opt = FlatAdam([model], lr=1e-4, capturable=True)
    img, pstate, _, _ = (t.to(dev) for t in synthetic.make_inputs((84, 84), B, 3407))
    tm, tl = torch.randn(B, 2, device=dev), torch.randn(B, 2, device=dev)

    def step():
        for p in model.parameters():
            p.grad = None
        mean, log_std = model([img, pstate])
        loss = ((mean - tm) ** 2).mean() + ((log_std - tl) ** 2).mean()
        loss.backward()
        opt.step()
        return loss

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    eager = timeit(step)
    g = dgvit_amd.GraphedStep(step, warmup=3)
    graph = timeit(g)
    out[f"overlap{overlap}"] = {"eager_ms": round(eager, 3), "graph_ms": round(graph, 3), "eager_fps": round(B / eager * 1e3),
                                "graph_fps": round(B / graph * 1e3)}
    print(out, flush=True)
print(json.dumps(out))
