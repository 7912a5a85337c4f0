#!/usr/bin/env python3
"""Where does a workgroup of the fp32 GEMM spend its cycles?  Per-workgroup shader-clock stamps (dgvit_set_gemm_stamps) of one
launch per shape: start -> first k-tile in LDS (prologue) -> main loop done -> stores drained (epilogue), plus the CU it ran on.

For every shape: medians of the three phases, the loop's slowdown against a wave that has its SIMD to itself
(k-tiles x MFMAs per k-tile x 64 cycles), the workgroups resident per CU on average, and per CU the MFMA cycles the resident
workgroups issued over the CU's busy span (= MFMA pipe use seen from the stamps).
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dgvit_amd
from dgvit_amd import functional as F
lib = dgvit_amd.diagnostic_library().__enter__()   # libdgvit_hip_diag.so: the A/B knobs live there (include/dgvit_hip_diag.h)
dev = "cuda"

CASES = [("qkv fwd  NT N=1536 K=256 ", 0, 0, 1536, 256, 64, 128, 16),
         ("fc1 fwd  NT N=2048 K=256 gelu2", 0, 1, 2048, 256, 64, 128, 16),
         ("dfc2     NN N=2048 K=256 dgelu", 1, 2, 2048, 256, 64, 128, 16),
         ("fc2 fwd  NT N=256 K=2048", 0, 0, 256, 2048, 64, 64, 32),
         ("dfc1     NN N=256 K=2048", 1, 0, 256, 2048, 64, 64, 32)]
M = int(os.environ.get("M", 25600))


def run(name, layout, epi, n, k, bm, bn, bk):
    A = torch.randn(M, k, device=dev)
    B = torch.randn(n, k, device=dev) if layout == 0 else torch.randn(k, n, device=dev)
    bias = torch.randn(n, device=dev) if layout == 0 else None
    res = torch.randn(M, n, device=dev) if (layout == 0 and epi == 0) else None
    aux = torch.randn(M, n, device=dev) if epi == 2 else None
    hint = bm * 1000000 + bn * 1000 + bk
    lib.dgvit_set_gemm_tile(hint)
    lib.dgvit_set_gemm_split(0)
    call = lambda: F.op_gemm(layout, epi, A, B, M, n, k, bias=bias, res=res, aux=aux, want_c2=(epi == 1))
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        call()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    wgs = (M // bm) * (n // bn)
    st = torch.zeros(wgs, 16, dtype=torch.int64, device=dev)
    lib.dgvit_set_gemm_stamps(st.data_ptr(), wgs)
    for _ in range(3):      # the last launch's stamps stay (earlier ones keep the chip in its loaded state)
        call()
    torch.cuda.synchronize()
    lib.dgvit_set_gemm_stamps(None, 0)
    lib.dgvit_set_gemm_tile(0)
    lib.dgvit_set_gemm_split(1)
    t = st.cpu().numpy()
    pro, loop, epi_c = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    drain = t[:, 7] - t[:, 3]
    t[:, 3] = t[:, 7]
    tot = t[:, 3] - t[:, 0]
    mhz = tot / np.maximum(t[:, 6] - t[:, 4], 1) * 100.0
    hw = t[:, 5] & 0xFFFFFFFF
    xcc = (t[:, 5] >> 32) & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    nk = k // bk
    mfma_per_ktile = (bm * bn * bk * 2) // 4 // 4096          # per wave (4 waves)
    ideal = nk * mfma_per_ktile * 64
    med = lambda a: float(np.median(a))
    print(f"{name:34s} M={M} tile {bm}x{bn}x{bk}: {us:7.1f} us/launch, {2.0*M*n*k/us/1e6:6.1f} TF;  {wgs} workgroups on {len(np.unique(cu))} CUs, clock {med(mhz):.0f} MHz")
    print(f"    cycles per workgroup (median / p10 / p90): prologue {med(pro):7.0f} {np.percentile(pro,10):7.0f} {np.percentile(pro,90):7.0f}   "
          f"loop {med(loop):7.0f} {np.percentile(loop,10):7.0f} {np.percentile(loop,90):7.0f}   epilogue {med(epi_c):7.0f} {np.percentile(epi_c,10):7.0f} {np.percentile(epi_c,90):7.0f}   store drain {med(drain):7.0f} {np.percentile(drain,10):7.0f} {np.percentile(drain,90):7.0f}")
    print(f"    loop alone on its SIMD would be {ideal} cycles ({nk} k-tiles x {mfma_per_ktile} MFMAs x 64): loop slowdown {med(loop)/ideal:.2f}x; "
          f"phase shares of a workgroup's life: prologue {pro.sum()/tot.sum():.3f} loop {loop.sum()/tot.sum():.3f} epilogue {epi_c.sum()/tot.sum():.3f} drain {drain.sum()/tot.sum():.3f}")
    ch = []
    prev = t[:, 2]
    for c in range(4):
        if t[:, 8 + 2 * c].max() == 0:
            break
        ch.append(f"chunk {c}: image in LDS +{med(t[:, 8 + 2 * c] - prev):.0f}, stores issued +{med(t[:, 9 + 2 * c] - t[:, 8 + 2 * c]):.0f}")
        prev = t[:, 9 + 2 * c]
    print("    epilogue steps (median cycles): " + "; ".join(ch))
    use, conc, span_all = [], [], []
    for c in np.unique(cu):
        m = cu == c
        span = t[m, 3].max() - t[m, 0].min()
        use.append(m.sum() * ideal / span)          # per SIMD: every workgroup puts `ideal` MFMA cycles on each of the 4 SIMDs
        conc.append(tot[m].sum() / span)
        span_all.append(span)
    print(f"    per CU: workgroups resident on average {np.mean(conc):.2f}; MFMA pipe use over the CU's busy span: mean {np.mean(use):.3f} min {np.min(use):.3f} max {np.max(use):.3f}; "
          f"busy span mean {np.mean(span_all):.0f} cycles = {np.mean(span_all)/med(mhz):.1f} us, max {np.max(span_all)/med(mhz):.1f} us")
    # what an average resident workgroup is doing: sum of phase cycles / sum of CU spans
    tspan = float(np.sum(span_all))
    print(f"    resident workgroups by phase (time average per CU): prologue {pro.sum()/tspan:.2f}  loop {loop.sum()/tspan:.2f}  epilogue {epi_c.sum()/tspan:.2f}  drain {drain.sum()/tspan:.2f}", flush=True)


for c in CASES:
    run(*c)
