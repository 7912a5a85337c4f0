#!/usr/bin/env python3
"""The reference's launch-bound regime, fused two-launch blocks (block.hip, the product's default) against the seven-launch GEMM schedule,
interleaved in one process on the diagnostic library: policy.sample() of the shipped actor (L4 / H4 / D64, 128x160) at B = 1, 2, 32, 64
and of DGViT-small (84x84 @ 12, L6 / H8 / D256) at B = 1, 32, each replayed as one HIP graph.

    python tools/small_batch_ab.py            timing table (JSON lines)
    python tools/small_batch_ab.py trace [B]  20 eager calls at batch B (default 1) on the product library (for rocprofv3 --kernel-trace)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import dgvit_amd  # noqa: E402
import synthetic  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "trace":
    m = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).to(dev).eval()
    img, ps, _, _ = (t.to(dev) for t in synthetic.make_inputs((128, 160), int(sys.argv[2]) if len(sys.argv) > 2 else 1, 0))
    with torch.no_grad():
        for _ in range(20):
            m.sample([img, ps])
    torch.cuda.synchronize()
    sys.exit(0)

lib = dgvit_amd.diagnostic_library().__enter__()


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for label, model, image, batches in (("shipped L4/H4/D64 128x160", dgvit_amd.GoTPolicy(2, 2, 4, 4, 64), (128, 160), (1, 2, 4, 8, 16, 32, 64)),
                                     ("DGViT-small L6/H8/D256 84x84@12", dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)), (84, 84), (1, 2, 4, 32, 256)),
                                     ("L4/H4/D128 84x84@12", dgvit_amd.GoTPolicy(2, 2, 4, 4, 128, image_size=(84, 84), patch_size=(12, 12)), (84, 84), (1, 2, 4, 8, 32))):
    m = model.to(dev).eval()
    for B in batches:
        img, ps, _, _ = (t.to(dev) for t in synthetic.make_inputs(image, B, 0))

        def call():
            with torch.no_grad():
                return m.sample([img, ps])
        res = {}
        graphs = {}
        for on in (1, 0):
            lib.dgvit_set_block_path(2 * on, 1 << 20)   # 2: fused blocks wherever supported, any batch (the product uses them where this table says they win)
            graphs[on] = dgvit_amd.GraphedStep(call, warmup=3)
        rounds = {1: [], 0: []}
        for _ in range(5):
            for on in (1, 0):
                rounds[on].append(timed(graphs[on], 100 if B <= 64 else 10))
        lib.dgvit_set_block_path(1, 4160)
        med = {on: sorted(v)[len(v) // 2] for on, v in rounds.items()}
        print(json.dumps({"model": label, "batch": B, "fused_two_launch_blocks_ms": round(med[1] * 1e3, 4), "gemm_schedule_ms": round(med[0] * 1e3, 4),
                          "ratio": round(med[1] / med[0], 3)}), flush=True)
