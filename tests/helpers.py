"""Shared test helpers: fixture loading, input regeneration, gradient digests."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import dgvit_oracle as O  # noqa: E402  (tests are allowed to import the oracle)

GOLDEN = os.path.join(ROOT, "tests", "golden")

import contextlib  # noqa: E402

# A/B knobs of libdgvit_hip_diag.so (include/dgvit_hip_diag.h): name -> (setter, default arguments)
_KNOBS = {
    "gemm_tile": ("dgvit_set_gemm_tile", (0,)), "gemm_split": ("dgvit_set_gemm_split", (1,)), "ln_fusion": ("dgvit_set_ln_fusion", (1,)),
    "conv_gather": ("dgvit_set_conv_gather", (1,)), "grouped_reduce": ("dgvit_set_grouped_reduce", (1,)),
    "gemm_diagnostics": ("dgvit_set_gemm_diagnostics", (0,)), "gemm_persistent": ("dgvit_set_gemm_persistent", (0, 0)),
    "small_batch_path": ("dgvit_set_small_batch_path", (0, 0)), "block_path": ("dgvit_set_block_path", (1, 4160)), "gelu_grad_store": ("dgvit_set_gelu_grad_store", (1,)), "block_fuse": ("dgvit_set_block_fuse", (2,)), "gemm_bf16_tile": ("dgvit_set_gemm_bf16_tile", (0,)),
    "gemm_bf16_mfma16": ("dgvit_set_gemm_bf16_mfma16", (1,)), "gemm_bf16_group_m": ("dgvit_set_gemm_bf16_group_m", (8,)),
    "gemm_bf16_l2_budget_kb": ("dgvit_set_gemm_bf16_l2_budget_kb", (2048,)),
    "attention_bwd_single_pass": ("dgvit_set_attention_bwd_single_pass", (1,)), "attention_single_query": ("dgvit_set_attention_single_query", (1,)), "gemm_wgrad_slice_major": ("dgvit_set_gemm_wgrad_slice_major", (1,)), "gemm_lds_pad": ("dgvit_set_gemm_lds_pad", (0,)),
}


@contextlib.contextmanager
def knobs(force_diag=False, **kw):
    """Run a block with A/B knobs set.  The product library has no knobs: when every requested value is the shipped default the
    block runs on libdgvit_hip.so itself; otherwise the package is routed through libdgvit_hip_diag.so (same sources, -DDGVIT_DIAG)
    for the duration, the knobs are set there and put back afterwards.  Yields the active library."""
    import dgvit_amd
    want = {k: (v if isinstance(v, tuple) else (v,)) for k, v in kw.items()}
    for k in want:
        assert k in _KNOBS, k
    changed = {k: v for k, v in want.items() if tuple(int(x) for x in v) != _KNOBS[k][1]}
    if not changed and not force_diag:
        yield dgvit_amd.load_library()
        return
    with dgvit_amd.diagnostic_library() as lib:
        try:
            for k, v in changed.items():
                getattr(lib, _KNOBS[k][0])(*v)
            yield lib
        finally:
            for k in changed:
                getattr(lib, _KNOBS[k][0])(*_KNOBS[k][1])


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def fixture_cfg(fx):
    d = fx["meta/dims"]
    return O.GoTConfig(image=tuple(int(v) for v in fx["meta/image"]), patch=tuple(int(v) for v in fx["meta/patch"]),
                       dim=int(d[0]), depth=int(d[1]), heads=int(d[2]), dim_head=int(d[3]), mlp_dim=int(d[4]))


def got_case_inputs(fx, cfg, with_mask):
    """Regenerate exactly what make_golden.case_got fed the reference."""
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    img, _, _, _ = O.make_inputs(cfg, batch, seed)
    rs = np.random.RandomState(seed + 7)
    goal = torch.from_numpy(rs.standard_normal((batch, cfg.dim))).float()
    wout = torch.from_numpy(rs.standard_normal((batch, cfg.dim))).float()
    mask = None
    if with_mask:
        mask = torch.from_numpy((rs.random_sample((batch, cfg.tokens, cfg.dim)) < 0.9).astype(np.float32))
    return img, goal, wout, mask


def grad_digest(named_grads):
    """Same digest as make_golden.grad_summary, from a dict name -> grad tensor (or None)."""
    out = {}
    for k, g in named_grads.items():
        if g is None:
            out[("none", k)] = None
            continue
        g = g.detach().double().flatten().cpu()
        out[("norm", k)] = g.norm().item()
        out[("sum", k)] = g.sum().item()
        out[("head", k)] = g[:16].float().numpy()
    return out


def check_grad_digest(fx, tag, named_grads, rtol, atol, strip=""):
    """Compare gradients with the reference digest stored under ``tag`` in fixture ``fx``."""
    dig = grad_digest(named_grads)
    n_checked = 0
    for key in fx:
        if not key.startswith(tag + "/"):
            continue
        _, kind, pname = key.split("/", 2)
        ours = dig.get((kind, pname[len(strip):] if strip and pname.startswith(strip) else pname), "missing")
        assert not isinstance(ours, str), f"no gradient entry for {pname}"
        if kind == "none":
            assert ours is None or float(np.abs(named_grads[pname]).max()) == 0.0, f"{pname} should have no grad"
            continue
        assert ours is not None, f"{pname}: reference has a gradient, we have none"
        ref = fx[key]
        scale = float(fx[f"{tag}/norm/{pname}"])
        if kind == "head":
            np.testing.assert_allclose(ours, ref, rtol=rtol, atol=atol * max(1.0, scale), err_msg=key)
        elif kind == "norm":
            np.testing.assert_allclose(ours, ref, rtol=rtol, atol=atol, err_msg=key)
        else:  # sum: cancellation-prone, scale tolerance by the norm and sqrt(numel)
            np.testing.assert_allclose(ours, ref, rtol=rtol, atol=atol * max(1.0, scale) * 64, err_msg=key)
        n_checked += 1
    assert n_checked > 0
    return n_checked
