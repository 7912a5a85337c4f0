"""Where a workgroup of the bf16 stream GEMM spends its cycles (diagnostic build, timing variant 512: s_memtime stamps per wave).

    python tools/bf16_stream_stamps.py M N K [epi]
Per tile (first 8 of every workgroup), over workgroups and waves: cycles of the main loop split into first halves / barrier waits /
second halves, and of the epilogue.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd import functional as F  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0     # further timing-variant bits (256: SIMD partner roles, 1024: priority)
lib = dgvit_amd.diagnostic_library().__enter__()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
lib.dgvit_set_gemm_bf16_tile(256257)
for _ in range(20):      # warm clocks
    F.op_gemm_bf16(epi, x, w, bias=bias)
ntiles = ((M + 255) // 256) * ((N + 255) // 256)
nwg = min(ntiles, 256)
stamps = torch.zeros(nwg * 8 * 8 * 8, dtype=torch.int64, device="cuda")
lib.dgvit_set_gemm_bf16_stamps(ctypes.c_void_p(stamps.data_ptr()))
lib.dgvit_set_gemm_diagnostics(512 | variant)
F.op_gemm_bf16(epi, x, w, bias=bias)
torch.cuda.synchronize()
lib.dgvit_set_gemm_diagnostics(0)
lib.dgvit_set_gemm_bf16_stamps(ctypes.c_void_p(0))
s = stamps.cpu().numpy().reshape(nwg, 8, 8, 8)
nkt = (K + 63) // 64
f = lambda v: f"{np.median(v):7.0f} [{np.percentile(v, 10):6.0f} {np.percentile(v, 90):6.0f}]"
print(f"{M}x{N}x{K} epilogue {epi}: {ntiles} tiles on {nwg} workgroups, {nkt} k-tiles per tile (MFMA-bound: 2048 cycles per k-tile and SIMD)")
print("tile | whole | main loop = first halves + waits/barriers + second halves | epilogue   (median [p10 p90] cycles over workgroups and waves)")
for ti in range(min(8, (ntiles + nwg - 1) // nwg)):
    a = s[:, :, ti].reshape(-1, 8)
    a = a[a[:, 2] != 0]
    if len(a) == 0:
        continue
    whole, main, epil = a[:, 2] - a[:, 0], a[:, 1] - a[:, 0], a[:, 2] - a[:, 1]
    print(f"{ti} | {f(whole)} | {f(main)} = {f(a[:, 4])} + {f(a[:, 3])} + {f(a[:, 5])} | {f(epil)}   per k-tile: {np.median(main) / nkt:.0f}")
for wv in range(8):
    a = s[:, wv, 1:4].reshape(-1, 8)
    a = a[a[:, 2] != 0]
    if len(a):
        print(f"wave {wv}: first halves {f(a[:, 4])}  waits {f(a[:, 3])}  second halves {f(a[:, 5])}  epilogue {f(a[:, 2] - a[:, 1])}")
a = s[:, 0]
ok = a[:, :, 2] != 0
first = a[:, 0, 0]
print(f"workgroup start spread: {(first.max() - first.min())} cycles; end-of-tile realtime span {(a[:, :, 6][ok].max() - a[:, :, 6][ok].min()) / 100.0:.1f} us")
