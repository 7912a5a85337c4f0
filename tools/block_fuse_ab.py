#!/usr/bin/env python3
"""Single-frame / few-frame policy.sample() of the shipped actor as one HIP graph: the fused blocks with and without their two extra fusions
(bit 0: block 0 assembles its token rows; bit 1: the final RMSNorm in the last MLP kernel), interleaved in one process (diagnostic library)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd, synthetic
lib = dgvit_amd.diagnostic_library().__enter__()
lib.dgvit_set_block_path(2, 4160)
torch.manual_seed(0)
for mode in ("eval", "train"):
    m = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).to("cuda").train(mode == "train")
    for B in (1, 2, 8):
        img, ps, _, _ = (t.cuda() for t in synthetic.make_inputs((128, 160), B, 0))

        def call():
            with torch.no_grad():
                return m.sample([img, ps])
        graphs = {}
        for bits in (0, 1, 2, 3):
            lib.dgvit_set_block_fuse(bits)
            graphs[bits] = dgvit_amd.GraphedStep(call, warmup=3)
        res = {b: [] for b in graphs}
        for _ in range(5):
            for b, g in graphs.items():
                for _ in range(10):
                    g()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    g()
                torch.cuda.synchronize()
                res[b].append((time.perf_counter() - t0) / 100)
        print(json.dumps({"mode": mode, "batch": B, **{f"fuse={b}_ms": round(sorted(v)[2] * 1e3, 4) for b, v in res.items()}}), flush=True)
lib.dgvit_set_block_fuse(2)
