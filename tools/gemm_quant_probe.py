#!/usr/bin/env python3
"""How much of the narrow-output GEMMs' time is tile quantisation?  Same N, K, tile; M swept so that the tile count moves
through whole and fractional multiples of the CU count (T = 25600 gives 1600 64x64 tiles = 6.25 per CU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
if "PERSIST" in os.environ:
    lib.dgvit_set_gemm_persistent(int(os.environ["PERSIST"]), int(os.environ.get("PGRID", 0)))
if "GEMMDIAG" in os.environ:
    lib.dgvit_set_gemm_diagnostics(int(os.environ["GEMMDIAG"]))
MS = [int(x) for x in os.environ["MS"].split(",")] if "MS" in os.environ else (8192, 12288, 16384, 20480, 24576, 25600, 28672, 32768, 65536)
dev = "cuda"


def timeit(layout, epi, m, n, k, hint, reps=5, inner=8):
    A = torch.randn(m, k, device=dev)
    B = torch.randn(n, k, device=dev) if layout == 0 else torch.randn(k, n, device=dev)
    bias = torch.randn(n, device=dev) if layout == 0 else None
    res = torch.randn(m, n, device=dev) if (layout == 0 and epi == 0 and n <= 512) else None     # (QKV has no residual)
    aux = torch.randn(m, n, device=dev) if epi == 2 else None
    lib.dgvit_set_gemm_tile(hint)
    ts = []
    for r in range(reps + 1):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            F.op_gemm(layout, epi, A, B, m, n, k, bias=bias, res=res, aux=aux, want_c2=(epi == 1))
        e.record(); torch.cuda.synchronize()
        if r: ts.append(s.elapsed_time(e) / inner)
    lib.dgvit_set_gemm_tile(0)
    ts.sort(); return ts[len(ts) // 2]


CASES = [("fc2 fwd  NT N=256 K=2048 64x64x32", 0, 0, 256, 2048, 64064032, 64, 64),
         ("out fwd  NT N=256 K=512  64x64x32", 0, 0, 256, 512, 64064032, 64, 64),
         ("dfc1     NN N=256 K=2048 64x64x32", 1, 0, 256, 2048, 64064032, 64, 64),
         ("qkv fwd  NT N=1536 K=256 64x128x16", 0, 0, 1536, 256, 64128016, 64, 128),
         ("fc1 fwd  NT N=2048 K=256 64x128x16 gelu2", 0, 1, 2048, 256, 64128016, 64, 128),
         ("dfc2     NN N=2048 K=256 64x128x16 dgelu", 1, 2, 2048, 256, 64128016, 64, 128)]
for name, layout, epi, n, k, hint, bm, bn in CASES:
    for m in MS:
        ms = timeit(layout, epi, m, n, k, hint)
        tiles = (m // bm) * (n // bn)
        tf = 2.0 * m * n * k / ms / 1e9
        print(f"{name:42s} M={m:6d} tiles={tiles:6d} ({tiles/256:6.2f}/CU) {ms*1e3:8.1f} us {tf:6.1f} TF", flush=True)
