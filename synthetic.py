"""Synthetic workload helpers for bench.py and tools/ (NOT part of the oracle, NOT part of the product package).

Input distributions follow SURVEY.md section 8(d) "Synthetic inputs": frames U[0,1) fp32 (real frames are
uint8-origin + noise + blur, /255: env_lab.py:295-299,432-433), goal distance U[0,1], heading U[-1,1]
(env_lab.py:296-297), actions U[-1,1], targets N(0,1); all from numpy.random.RandomState(seed).
"""
import glob
import hashlib
import json
import os

import numpy as np
import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dgvit-depth-goal-guided-vision-transformer-_amd", "csrc")


def kernel_source_digest() -> str:
    """SHA-256 over the GEMM kernel sources and the headers they include (csrc/gemm*.hip, common.h, bf16.h, knobs.h, small_mma.h).
    tools/pmc_traffic.py stores it with the HBM-traffic figures it extracts from rocprofv3 counter passes; bench.py quotes such a
    figure only when the digest still matches, i.e. the bytes were measured on the kernels this run is timing."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(_CSRC, "gemm*.hip"))) + [os.path.join(_CSRC, n) for n in ("common.h", "bf16.h", "knobs.h", "small_mma.h")]:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()


def profile_traffic(pattern: str, kernel: str):
    """(bytes per launch or None, note) from the newest profiles/<pattern> whose kernel-source digest matches the tree."""
    root = os.path.dirname(os.path.abspath(__file__))
    paths = sorted(glob.glob(os.path.join(root, "profiles", pattern)))
    if not paths:
        return None, "not collected"
    path = paths[-1]
    rel = os.path.relpath(path, root)
    try:
        with open(path) as f:
            d = json.load(f)
        val = d["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None, f"{rel}: unreadable"
    commit = d.get("collected_at_commit", "unrecorded")
    if d.get("kernel_sources_sha256") != kernel_source_digest():
        return None, (f"{rel} (collected at commit {commit}) was measured on other GEMM kernel sources than this run's: not quoted "
                      "(re-collect with tools/collect_profiles.sh)")
    return val, f"from profile {rel}, collected at commit {commit} on the same GEMM kernel sources (sha256 {d['kernel_sources_sha256'][:12]}); not measured by this run"


def make_inputs(image, batch, seed):
    """(img (B,H,W), pstate (B,2), act (B,2), tgt (B,1)) as fp32 CPU tensors."""
    rs = np.random.RandomState(seed + 100003)
    img = rs.random_sample((batch,) + tuple(image))
    pstate = np.stack([rs.uniform(0.0, 1.0, batch), rs.uniform(-1.0, 1.0, batch)], 1)
    act = rs.uniform(-1.0, 1.0, (batch, 2))
    tgt = rs.standard_normal((batch, 1))
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    return f(img), f(pstate), f(act), f(tgt)


def fwd_flops_per_frame(image, patch, dim, depth, heads, dim_head=64, mlp_dim=2048):
    """GEMM-only forward FLOPs of one frame through the encoder:
    2*P*pd*D + L*(2*N*D*3I + 4*N^2*I + 2*N*I*D + 4*N*D*M); backward = 2x, so fwd+bwd = 3x."""
    P = (image[0] // patch[0]) * (image[1] // patch[1])
    pd, N, I = patch[0] * patch[1], P + 1, heads * dim_head
    return 2.0 * P * pd * dim + depth * (2.0 * N * dim * 3 * I + 4.0 * N * N * I + 2.0 * N * I * dim + 4.0 * N * dim * mlp_dim)


def fwd_flops_per_frame_executed(image, patch, dim, depth, heads, dim_head=64, mlp_dim=2048, prune_last=True):
    """The FLOPs the schedule really executes: with the last block pruned to token 0 (K and V for every token,
    Q / attention / to_out / feed-forward for one row per frame) the last layer costs
    2*N*D*2I (K, V) + 2*D*I (Q) + 4*N*I (one query row) + 2*I*D + 4*D*M instead of the dense layer."""
    P = (image[0] // patch[0]) * (image[1] // patch[1])
    pd, N, I = patch[0] * patch[1], P + 1, heads * dim_head
    layer = 2.0 * N * dim * 3 * I + 4.0 * N * N * I + 2.0 * N * I * dim + 4.0 * N * dim * mlp_dim
    last = 2.0 * N * dim * 2 * I + 2.0 * dim * I + 4.0 * N * I + 2.0 * I * dim + 4.0 * dim * mlp_dim
    return 2.0 * P * pd * dim + (depth - 1) * layer + (last if prune_last else layer)
