#!/usr/bin/env python3
"""Persistent vs per-tile fp32 GEMM: max |difference| over a grid of layouts / epilogues / tiles (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
for layout, epi, M, N, K, tile, wgs in [(1, 0, 900, 512, 256, 64128016, 24), (1, 2, 900, 512, 256, 64128016, 24), (1, 2, 900, 512, 256, 64064032, 24),
                                        (1, 4, 900, 512, 256, 64128016, 24), (1, 0, 128, 256, 256, 64128016, 0), (1, 2, 64, 128, 256, 64128016, 0), (0, 0, 1000, 384, 256, 64128016, 8), (0, 1, 777, 512, 256, 64128016, 16), (1, 0, 1000, 264, 512, 64064032, 8),
                                        (1, 2, 900, 512, 256, 64128016, 0)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) if layout == 0 else torch.randn(K, N, generator=g)).cuda()
    aux = torch.randn(M, N, generator=g).cuda() if epi in (2, 4) else None
    bias = torch.randn(N, generator=g).cuda() if layout == 0 else None
    res = torch.randn(M, N, generator=g).cuda() if (layout == 0 and epi == 0) else None
    outs = []
    for mode in (0, 2):
        lib.dgvit_set_gemm_tile(tile); lib.dgvit_set_gemm_persistent(mode, wgs); lib.dgvit_set_gemm_split(0)
        o = F.op_gemm(layout, epi, A, B, M, N, K, aux=aux, bias=bias, res=res, want_c2=(epi == 1))
        outs.append(torch.cat([x.reshape(-1) for x in o]).reshape(-1, N).clone() if isinstance(o, tuple) else o.clone())
        torch.cuda.synchronize()
    lib.dgvit_set_gemm_tile(0); lib.dgvit_set_gemm_persistent(0, 0); lib.dgvit_set_gemm_split(1)
    d = (outs[0] - outs[1]).abs()
    bad = (d > 0).nonzero()
    print(layout, epi, M, N, K, tile, wgs, "max diff", d.max().item(), "n bad", bad.shape[0], "first bad", bad[:3].tolist(), "rows bad", sorted(set(bad[:, 0].tolist()))[:12],
          "cols bad", sorted(set(bad[:, 1].tolist()))[:12], flush=True)
