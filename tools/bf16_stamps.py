"""Where a persistent bf16 ring-GEMM workgroup spends its cycles (diagnostic build with in-kernel s_memtime stamps).

    python tools/bf16_stamps.py M N K [tile]
Prints, over workgroups and their first 8 tiles, the cycles of a tile's main loop and of its epilogue.
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
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 256256
bm, bn = tile // 1000, tile % 1000
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
lib.dgvit_set_gemm_bf16_tile(tile)
for _ in range(20):      # warm clocks
    F.op_gemm_bf16(0, x, w)
ntiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
nwg = min(ntiles, 256)
stamps = torch.zeros(nwg * 2 * 8 * 4, dtype=torch.int64, device="cuda")
lib.dgvit_set_gemm_bf16_stamps(ctypes.c_void_p(stamps.data_ptr()))
F.op_gemm_bf16(0, x, w)
torch.cuda.synchronize()
lib.dgvit_set_gemm_bf16_stamps(ctypes.c_void_p(0))
s = stamps.cpu().numpy().reshape(nwg, 2, 8, 4)
nkt = (K + 31) // 32
mfma = 2 * (bm // 64) * (bn // 128) * 32 // 2 * 2
f = lambda v: f"median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}"
for grp in (0, 1):
    for ti in range(min(8, (ntiles + nwg - 1) // nwg)):
        a = s[:, grp, ti]
        a = a[a[:, 2] != 0]
        if len(a) == 0:
            continue
        main, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1]
        print(f"group {grp} tile {ti}: mainloop {f(main)} ({np.median(main) / nkt:.0f} cycles per 32-deep k-tile, MFMA-bound {mfma})"
              f" | epilogue {f(epi)}")
a = s[:, 0]
valid = a[:, :, 2] != 0
rt = a[:, :, 3][valid]
print(f"span of the stamped epilogue ends: {(rt.max() - rt.min()) / 100.0:.1f} us")
