"""bf16 configuration timings on one GPU: the four GEMM shapes of config 5 per tile, attention, and the whole forward.

    python tools/bf16_bench.py [--batch 256] [--tiles 256256,256128,128128]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tiles", default="256256,256257")
    ap.add_argument("--no-forward", action="store_true")
    ap.add_argument("--mfma16", type=int, default=-1, help="force the ring GEMM's MFMA shape: 1 = 16x16x32, 0 = 32x32x16 (default: library default)")
    a = ap.parse_args()
    lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
    if a.mfma16 >= 0:
        lib.dgvit_set_gemm_bf16_mfma16(a.mfma16)
    B, N, D, I, M = a.batch, 197, 768, 768, 3072
    T = B * N
    out = {"batch": B}
    shapes = {"qkv": (T, 3 * I, D, 0), "out": (T, D, I, 0), "fc1": (T, M, D, 1), "fc2": (T, D, M, 0)}
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, (m, n, k, epi) in shapes.items():
        x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
        bias = torch.randn(n, device="cuda", generator=g)
        res = torch.randn(m, n, device="cuda", generator=g) if epi == 2 else None
        for tile in [int(t) for t in a.tiles.split(",")]:
            lib.dgvit_set_gemm_bf16_tile(tile)
            ms = timeit(lambda: F.op_gemm_bf16(epi, x, w, bias=bias, res=res))
            lib.dgvit_set_gemm_bf16_tile(0)
            out[f"{name}_{tile}"] = {"ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}
            print(name, (m, n, k), tile, out[f"{name}_{tile}"], flush=True)
    qkv = torch.randn(B, N, 3 * I, device="cuda", generator=g).to(torch.bfloat16)
    ms = timeit(lambda: F.op_attention_bf16(qkv, 12, 64))
    out["attention"] = {"ms": round(ms, 4), "tflops": round(4.0 * N * N * 64 * 12 * B / ms / 1e9, 1)}
    print("attention", out["attention"], flush=True)
    x = torch.randn(T, D, device="cuda", generator=g)
    gam, bet = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    ms = timeit(lambda: F.op_layernorm_bf16(x, gam, bet))
    out["layernorm"] = {"ms": round(ms, 4), "GBps": round(T * D * 6 / ms / 1e6, 1)}
    print("layernorm", out["layernorm"], flush=True)
    if not a.no_forward:
        m = dgvit_amd.GoT(image_size=224, patch_size=16, num_classes=2, dim=768, depth=12, heads=12, mlp_dim=3072, channels=1)
        m = m.cuda().eval().set_compute_dtype(torch.bfloat16)
        img, goal = torch.rand(B, 224, 224, device="cuda", generator=g), torch.randn(B, 768, device="cuda", generator=g)
        with torch.no_grad():
            ms = timeit(lambda: m(img, goal), iters=10)
        flops = 34.972e9 * B
        out["forward_c5"] = {"ms": round(ms, 3), "frames_per_s": round(B / ms * 1e3, 1), "tflops_dense": round(flops / ms / 1e9, 1)}
        print("forward_c5", out["forward_c5"], flush=True)
    if not a.no_forward:
        # forward + backward (training mode: dense last block, activations kept), no optimizer step
        m.train()
        tgt = torch.randn(B, 768, device="cuda", generator=g)

        def fb():
            for p_ in m.parameters():
                p_.grad = None
            loss = ((m(img, goal) - tgt) ** 2).mean()
            loss.backward()

        import ctypes
        from dgvit_amd import _lib
        ms = timeit(fb, iters=5, warm=2)
        lib.dgvit_profile_start(8192)
        fb()
        torch.cuda.synchronize()
        kinds = _lib.PROFILE_KINDS
        pm, pw, pc = (ctypes.c_double * kinds)(), (ctypes.c_double * kinds)(), (ctypes.c_longlong * kinds)()
        lib.dgvit_profile_stop(pm, pw, pc)
        out["fwd_bwd_c5"] = {"ms": round(ms, 3), "frames_per_s": round(B / ms * 1e3, 1), "tflops_dense": round(3 * 34.972e9 * B / ms / 1e9, 1),
                             "gemm_ms": round(pm[0], 3), "gemm_tflops": round(pw[0] / max(pm[0], 1e-9) / 1e9, 1), "attn_fwd_ms": round(pm[1], 3),
                             "attn_bwd_ms": round(pm[2], 3), "other_profiled_ms": round(pm[3], 3),
                             "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
        print("fwd_bwd_c5", out["fwd_bwd_c5"], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
