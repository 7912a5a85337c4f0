#!/usr/bin/env python3
"""Generate golden vectors for the DGViT hot path FROM THE REFERENCE ITSELF.

Run only in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

It imports the reference's ``GoalFormer.py`` / ``got_sac_network.py`` unchanged,
fills the modules' ``state_dict`` from ``oracle.dgvit_oracle.make_params`` (a
``numpy.random.RandomState`` stream, so only the seed has to be stored), runs
them on CPU and stores outputs, loss values and gradient summaries under
``tests/golden/*.npz``.  Fixtures are data only: seeds, shapes, outputs.

For image/patch shapes the reference classes cannot take (they hard-wire
``Rearrange(p1=16, p2=20)`` + ``Linear(320, dim)``, GoalFormer.py:137-139) the
generator swaps that one ``nn.Sequential`` for the same two ops with the
requested patch size; everything else -- Transformer,
Attention, FeedForward, PreNorm, RMSNorm, GoT.forward -- is the reference's code.
Train-mode dropout is reproduced by swapping ``trans.dropout`` for a module
that applies a stored Bernoulli mask with nn.Dropout's 1/(1-p) scaling.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src/vis_nav/vis_nav")

import numpy as np
import torch
from torch import nn
from einops.layers.torch import Rearrange

import GoalFormer as ref_gf            # noqa: E402  (the reference)
import got_sac_network as ref_net      # noqa: E402  (the reference)
from oracle import dgvit_oracle as O   # noqa: E402

torch.set_num_threads(8)


class MaskDrop(nn.Module):
    """nn.Dropout(p) with the Bernoulli draw supplied from outside."""

    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, x):
        return x * self.mask / (1.0 - self.p)


def resize_patch_embed(trans, cfg):
    ph, pw = cfg.patch
    trans.to_patch_embedding = nn.Sequential(
        Rearrange('b (h p1) (w p2) -> b (h w) (p1 p2)', p1=ph, p2=pw),
        nn.Linear(ph * pw, cfg.dim))


def build_got(cfg, pool='cls'):
    m = ref_gf.GoT(image_size=cfg.image, patch_size=cfg.patch, num_classes=cfg.num_classes, dim=cfg.dim,
                   depth=cfg.depth, heads=cfg.heads, mlp_dim=cfg.mlp_dim, channels=1, dim_head=cfg.dim_head, pool=pool)
    if tuple(cfg.patch) != (16, 20):
        resize_patch_embed(m, cfg)
    return m


def build_net(kind, cfg):
    ctor = {"policy": ref_net.GoTPolicy, "qnet": ref_net.GoTQNetwork, "detpolicy": ref_net.DeterministicGoTPolicy}[kind]
    m = ctor(2, 2, cfg.depth, cfg.heads, cfg.dim)
    assert cfg.mlp_dim == 2048 and cfg.dim_head == 64
    if tuple(cfg.image) != (128, 160) or tuple(cfg.patch) != (16, 20):
        resize_patch_embed(m.trans, cfg)
        m.trans.pos_embedding = nn.Parameter(torch.zeros(1, cfg.tokens, cfg.dim))
    return m


def load(m, spec, seed):
    params = O.make_params(spec, seed)
    keys_ref = list(m.state_dict().keys())
    assert keys_ref == [k for k, _, _ in spec], "state_dict key order differs from oracle spec"
    m.load_state_dict(params, strict=True)
    return params


def grad_summary(m, out, tag):
    """Per-parameter gradient digest: norm, sum, first 16 values (None-grad params flagged)."""
    for k, p in m.named_parameters():
        if p.grad is None:
            out[f"{tag}/none/{k}"] = np.zeros(0, np.float32)
            continue
        g = p.grad.detach().double().flatten()
        out[f"{tag}/norm/{k}"] = np.array(g.norm().item())
        out[f"{tag}/sum/{k}"] = np.array(g.sum().item())
        out[f"{tag}/head/{k}"] = g[:16].float().numpy()


def save(name, out):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def cfg_meta(cfg, batch, seed):
    return {"meta/image": np.array(cfg.image), "meta/patch": np.array(cfg.patch),
            "meta/dims": np.array([cfg.dim, cfg.depth, cfg.heads, cfg.dim_head, cfg.mlp_dim]),
            "meta/batch": np.array(batch), "meta/seed": np.array(seed)}


def case_got(name, cfg, batch, seed, with_mask=False, full_grads=False, pool='cls'):
    """Bare GoT encoder: features, per-layer token-0 rows, gradients of sum(feat * w)."""
    m = build_got(cfg, pool)
    load(m, O.got_param_spec(cfg, prefix=""), seed)
    img, _, _, _ = O.make_inputs(cfg, batch, seed)
    rs = np.random.RandomState(seed + 7)
    goal = torch.from_numpy(rs.standard_normal((batch, cfg.dim))).float().requires_grad_(True)
    wout = torch.from_numpy(rs.standard_normal((batch, cfg.dim))).float()
    out = cfg_meta(cfg, batch, seed)
    if with_mask:
        mask = torch.from_numpy((rs.random_sample((batch, cfg.tokens, cfg.dim)) < 0.9).astype(np.float32))
        m.dropout = MaskDrop(mask, 0.1)
        m.train()
    else:
        m.eval()
    feat = m(img, goal)
    (feat * wout).sum().backward()
    out["feat"] = feat.detach().numpy()
    out["dgoal"] = goal.grad.numpy()
    grad_summary(m, out, "g")
    if full_grads:
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"gfull/{k}"] = p.grad.numpy()
    save(name, out)


def case_policy(name, cfg, batch, seed, grads=True):
    m = build_net("policy", cfg)
    load(m, O.policy_param_spec(cfg), seed)
    m.eval()
    img, pstate, act, tgt = O.make_inputs(cfg, batch, seed)
    out = cfg_meta(cfg, batch, seed)
    mean, log_std = m([img, pstate])
    out["mean"], out["log_std"] = mean.detach().numpy(), log_std.detach().numpy()
    # sample(): the N(0,1) draw of Normal.rsample (got_sac_network.py:242) under a fixed torch seed
    torch.manual_seed(seed)
    action, log_prob, tmean = m.sample([img, pstate])
    torch.manual_seed(seed)
    eps = torch.randn(batch, 2)
    std = log_std.exp()
    assert torch.allclose(action, torch.tanh(mean + std * eps), atol=1e-6), "noise stream mismatch"
    out["noise"], out["action"] = eps.numpy(), action.detach().numpy()
    out["log_prob"], out["tanh_mean"] = log_prob.detach().numpy(), tmean.detach().numpy()
    if grads:
        m.zero_grad()
        loss = (mean ** 2).mean() + (log_std ** 2).mean()
        loss.backward()
        out["loss"] = np.array(loss.item())
        grad_summary(m, out, "g")
    save(name, out)


def case_sac(name, cfg, batch, seed):
    """Actor + transformer critic, losses exactly as DRL.py:396-410 (alpha fixed 0.2)."""
    pol = build_net("policy", cfg)
    load(pol, O.policy_param_spec(cfg), seed)
    crt = build_net("qnet", cfg)
    load(crt, O.qnet_param_spec(cfg), seed + 1)
    pol.eval(); crt.eval()
    img, pstate, act, tgt = O.make_inputs(cfg, batch, seed)
    out = cfg_meta(cfg, batch, seed)
    alpha = 0.2
    # critic update (DRL.py:396-402)
    q1, q2 = crt([img, pstate, act])
    qf_loss = torch.nn.functional.mse_loss(q1, tgt.expand_as(q1)) + torch.nn.functional.mse_loss(q2, tgt.expand_as(q2))
    crt.zero_grad(); qf_loss.backward()
    out["q1"], out["q2"], out["qf_loss"] = q1.detach().numpy(), q2.detach().numpy(), np.array(qf_loss.item())
    grad_summary(crt, out, "gc")
    # actor update (DRL.py:405-413)
    torch.manual_seed(seed)
    pi, log_pi, _ = pol.sample([img, pstate])
    torch.manual_seed(seed)
    out["noise"] = torch.randn(batch, 2).numpy()
    q1p, q2p = crt([img, pstate, pi])
    policy_loss = ((alpha * log_pi) - torch.min(q1p, q2p)).mean()
    pol.zero_grad(); crt.zero_grad(); policy_loss.backward()
    out["pi"], out["log_pi"] = pi.detach().numpy(), log_pi.detach().numpy()
    out["policy_loss"] = np.array(policy_loss.item())
    grad_summary(pol, out, "ga")
    save(name, out)


def case_det(name, cfg, batch, seed):
    m = build_net("detpolicy", cfg)
    load(m, O.detpolicy_param_spec(cfg), seed)
    m.eval()
    img, pstate, _, _ = O.make_inputs(cfg, batch, seed)
    out = cfg_meta(cfg, batch, seed)
    mean = m([img, pstate])
    out["mean"] = mean.detach().numpy()
    (mean ** 2).mean().backward()
    grad_summary(m, out, "g")
    save(name, out)


def case_cnn(name, kind, batch, seed, image=(128, 160)):
    """Reference QNetwork / GaussianPolicy (CNN, got_sac_network.py:125-170, 258-327), unmodified classes."""
    cfg = O.GoTConfig(image=image)
    if kind == "qnet":
        m, spec = ref_net.QNetwork(2, 2), O.cnn_qnet_param_spec()
    else:
        m, spec = ref_net.GaussianPolicy(2, 2), O.cnn_policy_param_spec()
    load(m, spec, seed)
    img, pstate, act, tgt = O.make_inputs(cfg, batch, seed)
    out = {"meta/image": np.array(image), "meta/batch": np.array(batch), "meta/seed": np.array(seed)}
    if kind == "qnet":
        q1, q2 = m([img, pstate, act])
        loss = torch.nn.functional.mse_loss(q1, tgt.expand_as(q1)) + torch.nn.functional.mse_loss(q2, tgt.expand_as(q2))
        out["q1"], out["q2"] = q1.detach().numpy(), q2.detach().numpy()
    else:
        mean, log_std = m([img, pstate])
        loss = (mean ** 2).mean() + (log_std ** 2).mean()
        out["mean"], out["log_std"] = mean.detach().numpy(), log_std.detach().numpy()
        # sample() (got_sac_network.py:303-316) under a fixed torch seed, with the N(0,1) draw it consumed
        torch.manual_seed(seed)
        action, log_prob, tmean = m.sample([img, pstate])
        torch.manual_seed(seed)
        eps = torch.randn(batch, 2)
        assert torch.allclose(action, torch.tanh(mean + log_std.exp() * eps), atol=1e-6), "noise stream mismatch"
        out["noise"], out["action"] = eps.numpy(), action.detach().numpy()
        out["log_prob"], out["tanh_mean"] = log_prob.detach().numpy(), tmean.detach().numpy()
    loss.backward()
    out["loss"] = np.array(loss.item())
    grad_summary(m, out, "g")
    save(name, out)


def main():
    C = O.GoTConfig
    cases = []

    def add(fn, name, *a, **k):
        cases.append((name, lambda: fn(name, *a, **k)))

    # tiny, arbitrary dims (bare GoT; dim_head 32, the smallest the fused attention kernel takes), full gradients
    tiny = C(image=(16, 24), patch=(8, 8), dim=32, depth=2, heads=2, dim_head=32, mlp_dim=64)
    add(case_got, "got_tiny_eval", tiny, 3, 0, full_grads=True)
    add(case_got, "got_tiny_mask", tiny, 3, 1, with_mask=True, full_grads=True)
    add(case_got, "got_tiny_meanpool", tiny, 3, 2, full_grads=True, pool='mean')
    # patch sizes BASELINE leaves open for 84x84 (N = 50 / 37 / 145 / 197), small width
    for ps in (12, 14, 7, 6):
        add(case_got, f"got_84p{ps}", C(image=(84, 84), patch=(ps, ps), dim=64, depth=1, heads=2), 2, 10 + ps)
    # native 128x160, unmodified reference classes (C0a shipped, C0b small)
    add(case_policy, "policy_native_shipped", C(dim=64, depth=4, heads=4), 2, 3407)
    add(case_policy, "policy_native_small", C(dim=256, depth=6, heads=8), 2, 3408)
    add(case_det, "detpolicy_native_shipped", C(dim=64, depth=4, heads=4), 2, 3409)
    # heads == 1 and dim_head == dim: Attention.to_out is nn.Identity() (GoalFormer.py:56,66-69) -- GoTPolicy(2, 2, 2, 1, 64), unmodified
    # reference class; and the same corner on the bare encoder with a train-mode mask and full gradients (dim_head 32)
    add(case_policy, "policy_native_h1", C(dim=64, depth=2, heads=1), 2, 3410)
    add(case_got, "got_tiny_h1_mask", C(image=(16, 24), patch=(8, 8), dim=32, depth=2, heads=1, dim_head=32, mlp_dim=64), 3, 4,
        with_mask=True, full_grads=True)
    # C2/C3 shape: 84x84 @ 12, L6 H8 D256
    c2 = C(image=(84, 84), patch=(12, 12), dim=256, depth=6, heads=8)
    add(case_policy, "policy_c2", c2, 4, 0)
    add(case_sac, "sac_c2", c2, 4, 0)
    add(case_got, "got_c2_mask", c2, 2, 5, with_mask=True)
    # the shipped CNN critic and the CNN actor, native 128x160 and 84x84
    add(case_cnn, "cnn_qnet_native", "qnet", 3, 21)
    add(case_cnn, "cnn_qnet_84", "qnet", 2, 22, image=(84, 84))
    add(case_cnn, "cnn_policy_native", "policy", 2, 23)
    # C5 shape cut to depth 2: 224x224 @ 16, H12 D768 M3072
    add(case_got, "got_c5_l2", C(image=(224, 224), patch=(16, 16), dim=768, depth=2, heads=12, mlp_dim=3072), 1, 6)

    want = sys.argv[1:]            # `python make_golden.py name ...` regenerates only those cases
    unknown = [w for w in want if w not in [n for n, _ in cases]]
    assert not unknown, f"unknown cases {unknown}"
    for name, run in cases:
        if not want or name in want:
            run()


if __name__ == "__main__":
    main()
