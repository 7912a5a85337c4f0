"""Timing decomposition of the bf16 stream GEMM (diagnostic library): which of DMA delivery / fragment reads / barrier / MFMA / epilogue
sets the pace.  python tools/bf16_stream_diag.py [batch]  -- results of the diagnostic arms are garbage by construction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 440
lib = dgvit_amd.diagnostic_library().__enter__()
T = B * 197
g = torch.Generator(device="cuda").manual_seed(0)
# timing variants of gemm_bf16_stream_kernel (csrc/gemm_bf16_stream.hip, SDIAG bits): 1 cache-hot source, 2 no DMA, 4 no fragment reads,
# 32768 / 65536 who issues the LDS-DMAs, 131072.. priority window of waves 4-7, 8 no epilogue, 16 no barrier, 32 no MFMA, 64 accumulators kept alive without an epilogue, 128 DMA amid the MFMAs, 256 SIMD partners
# lead / trail, 512 per-wave stamps (tools/bf16_stream_stamps.py), 1024 waves 4-7 at priority 1, 2048 ORDINARY output stores (the
# shipped kernel's are non-temporal)
ARMS = [("stream (warm-up)", 256257, 0), ("stream", 256257, 0), ("fragment reads 2 blocks ahead", 256257, 1048576), ("fragment reads 4 blocks ahead", 256257, 2097152),
        ("stream again", 256257, 0), ("2 ahead again", 256257, 1048576), ("4 ahead again", 256257, 2097152)]
EXACT = (32768, 65536, 1024, 131072, 262144, 1048576, 2097152)     # variants that compute the real result: checked bit for bit against the shipped schedule


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


for name, (m, n, k, epi) in {"qkv": (T, 2304, 768, 0), "fc1": (T, 3072, 768, 1), "fc2": (T, 768, 3072, 0)}.items():
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda", generator=g)
    for arm, tile, diag in ARMS:
        if epi == 4 and diag not in (0, 2, 32, 40):
            continue
        lib.dgvit_set_gemm_bf16_tile(tile)
        lib.dgvit_set_gemm_diagnostics(diag)
        if diag in EXACT:
            got = F.op_gemm_bf16(epi, x, w, bias=bias)
            lib.dgvit_set_gemm_diagnostics(0)
            assert torch.equal(got, F.op_gemm_bf16(epi, x, w, bias=bias)), (name, arm)
            lib.dgvit_set_gemm_diagnostics(diag)
        ms = timeit(lambda: F.op_gemm_bf16(epi, x, w, bias=bias))
        print(f"{name} ({m}, {n}, {k}) {arm:24s} {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s-equivalent", flush=True)
    lib.dgvit_set_gemm_diagnostics(0)
    for phases in ():
        lib.dgvit_set_gemm_bf16_group_m(8 + 1000 * phases)
        lib.dgvit_set_gemm_bf16_tile(256257)
        ms = timeit(lambda: F.op_gemm_bf16(epi, x, w, bias=bias))
        print(f"{name} ({m}, {n}, {k}) stagger {phases:2d} phases        {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
    lib.dgvit_set_gemm_bf16_group_m(8)
    lib.dgvit_set_gemm_bf16_tile(0)
