"""Synthetic workload helpers for bench.py and tools/ (NOT part of the oracle, NOT part of the product package).

Input distributions follow SURVEY.md section 8(d) "Synthetic inputs": frames U[0,1) fp32 (real frames are
uint8-origin + noise + blur, /255: env_lab.py:295-299,432-433), goal distance U[0,1], heading U[-1,1]
(env_lab.py:296-297), actions U[-1,1], targets N(0,1); all from numpy.random.RandomState(seed).
"""
import numpy as np
import torch


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
