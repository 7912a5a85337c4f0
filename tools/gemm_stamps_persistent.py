#!/usr/bin/env python3
"""Stamps of the persistent fp32 GEMM (dgvit_set_gemm_persistent(2)): first two tiles of every workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
dev = "cuda"
M = 25600
for name, layout, epi, n, k, hint in [("qkv fwd NT N=1536 K=256", 0, 0, 1536, 256, 64128016), ("fc1 fwd NT N=2048 K=256 gelu2", 0, 1, 2048, 256, 64128016),
                                      ("dfc2 NN N=2048 K=256 dgelu", 1, 2, 2048, 256, 64128016)]:
    A = torch.randn(M, k, device=dev)
    B = torch.randn(n, k, device=dev) if layout == 0 else torch.randn(k, n, device=dev)
    bias = torch.randn(n, device=dev) if layout == 0 else None
    aux = torch.randn(M, n, device=dev) if epi == 2 else None
    lib.dgvit_set_gemm_tile(hint)
    lib.dgvit_set_gemm_persistent(2, 0)
    call = lambda: F.op_gemm(layout, epi, A, B, M, n, k, bias=bias, aux=aux, want_c2=(epi == 1))
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    wgs = 1280
    st = torch.zeros(wgs, 16, dtype=torch.int64, device=dev)
    lib.dgvit_set_gemm_stamps(st.data_ptr(), wgs)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.dgvit_set_gemm_stamps(None, 0)
    lib.dgvit_set_gemm_tile(0)
    lib.dgvit_set_gemm_persistent(0, 0)
    t = st.cpu().numpy()
    med = lambda a: float(np.median(a))
    print(f"{name}: cycles (median): start -> tile0 loop start {med(t[:,8]-t[:,0]):.0f}; tile0 loop {med(t[:,9]-t[:,8]):.0f}; tile0 tile-end work (side, fetch issue, stores) {med(t[:,10]-t[:,9]):.0f}; "
          f"tile0 stores issued -> tile1 loop start {med(t[:,11]-t[:,10]):.0f}; tile1 loop {med(t[:,12]-t[:,11]):.0f}; tile1 tile-end {med(t[:,13]-t[:,12]):.0f}; whole workgroup {med(t[:,7]-t[:,0]):.0f}; "
          f"clock {med((t[:,7]-t[:,0])/np.maximum(t[:,6]-t[:,4],1)*100):.0f} MHz", flush=True)
