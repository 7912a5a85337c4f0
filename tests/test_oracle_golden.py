"""CPU: the oracle restatement is pinned against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import numpy as np
import pytest
import torch

from helpers import O, load_fixture, fixture_cfg, got_case_inputs, check_grad_digest

TOL = 2e-6  # fp32 abs, outputs are O(1) (RMSNorm'd features / small head outputs)


def _leaf_params(spec, seed):
    p = O.make_params(spec, seed)
    for v in p.values():
        v.requires_grad_(True)
    return p


@pytest.mark.parametrize("name,mask", [("got_tiny_eval", False), ("got_tiny_mask", True), ("got_84p12", False),
                                       ("got_84p14", False), ("got_84p7", False), ("got_84p6", False),
                                       ("got_c2_mask", True), ("got_c5_l2", False), ("got_tiny_h1_mask", True)])
def test_got_cases(name, mask):
    fx = load_fixture(name)
    cfg = fixture_cfg(fx)
    p = _leaf_params(O.got_param_spec(cfg, prefix=""), int(fx["meta/seed"]))
    img, goal, wout, m = got_case_inputs(fx, cfg, mask)
    goal.requires_grad_(True)
    feat = O.got_forward(p, img, goal, cfg, drop_mask=m, prefix="")
    np.testing.assert_allclose(feat.detach().numpy(), fx["feat"], rtol=0, atol=TOL)
    (feat * wout).sum().backward()
    np.testing.assert_allclose(goal.grad.numpy(), fx["dgoal"], rtol=1e-4, atol=1e-5)
    check_grad_digest(fx, "g", {k: v.grad for k, v in p.items()}, rtol=2e-4, atol=2e-5)
    if "gfull/pos_embedding" in fx:
        for k, v in p.items():
            if f"gfull/{k}" in fx:
                np.testing.assert_allclose(v.grad.numpy(), fx[f"gfull/{k}"], rtol=1e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("name", ["policy_native_shipped", "policy_native_small", "policy_c2", "policy_native_h1"])
def test_policy_cases(name):
    fx = load_fixture(name)
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    p = _leaf_params(O.policy_param_spec(cfg), seed)
    img, pstate, _, _ = O.make_inputs(cfg, batch, seed)
    mean, log_std = O.policy_forward(p, img, pstate, cfg)
    np.testing.assert_allclose(mean.detach().numpy(), fx["mean"], rtol=0, atol=TOL)
    np.testing.assert_allclose(log_std.detach().numpy(), fx["log_std"], rtol=0, atol=TOL)
    act, logp, tmean = O.policy_sample(p, img, pstate, cfg, torch.from_numpy(fx["noise"]))
    np.testing.assert_allclose(act.detach().numpy(), fx["action"], rtol=0, atol=TOL)
    np.testing.assert_allclose(logp.detach().numpy(), fx["log_prob"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tmean.detach().numpy(), fx["tanh_mean"], rtol=0, atol=TOL)
    loss = (mean ** 2).mean() + (log_std ** 2).mean()
    np.testing.assert_allclose(loss.item(), float(fx["loss"]), rtol=1e-5)
    loss.backward()
    check_grad_digest(fx, "g", {k: v.grad for k, v in p.items()}, rtol=5e-4, atol=2e-6)


def test_detpolicy_case():
    fx = load_fixture("detpolicy_native_shipped")
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    p = _leaf_params(O.detpolicy_param_spec(cfg), seed)
    img, pstate, _, _ = O.make_inputs(cfg, batch, seed)
    mean = O.detpolicy_forward(p, img, pstate, cfg)
    np.testing.assert_allclose(mean.detach().numpy(), fx["mean"], rtol=0, atol=TOL)
    (mean ** 2).mean().backward()
    check_grad_digest(fx, "g", {k: v.grad for k, v in p.items()}, rtol=5e-4, atol=2e-6)


def test_sac_losses_case():
    """Critic and actor losses restated from DRL.py:396-410 against the reference run."""
    fx = load_fixture("sac_c2")
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    pa = _leaf_params(O.policy_param_spec(cfg), seed)
    pc = _leaf_params(O.qnet_param_spec(cfg), seed + 1)
    img, pstate, act, tgt = O.make_inputs(cfg, batch, seed)
    q1, q2 = O.qnet_forward(pc, img, pstate, act, cfg)
    np.testing.assert_allclose(q1.detach().numpy(), fx["q1"], rtol=0, atol=TOL)
    np.testing.assert_allclose(q2.detach().numpy(), fx["q2"], rtol=0, atol=TOL)
    qf = O.sac_critic_loss(q1, q2, tgt)
    np.testing.assert_allclose(qf.item(), float(fx["qf_loss"]), rtol=1e-5)
    qf.backward()
    check_grad_digest(fx, "gc", {k: v.grad for k, v in pc.items()}, rtol=5e-4, atol=2e-6)
    for v in pc.values():
        v.grad = None
    pi, log_pi, _ = O.policy_sample(pa, img, pstate, cfg, torch.from_numpy(fx["noise"]))
    np.testing.assert_allclose(pi.detach().numpy(), fx["pi"], rtol=0, atol=TOL)
    q1p, q2p = O.qnet_forward(pc, img, pstate, pi, cfg)
    loss = O.sac_actor_loss(0.2, log_pi, q1p, q2p)
    np.testing.assert_allclose(loss.item(), float(fx["policy_loss"]), rtol=1e-5, atol=1e-6)
    loss.backward()
    check_grad_digest(fx, "ga", {k: v.grad for k, v in pa.items()}, rtol=5e-4, atol=2e-6)


def test_identity_to_out_has_no_parameters():
    """heads == 1 and dim_head == dim: the reference's Attention.to_out is nn.Identity() (GoalFormer.py:56,66-69); the fixture was
    produced by strict-loading the oracle's parameter spec into the reference's GoTPolicy(2, 2, 2, 1, 64), so the key order is pinned."""
    cfg = fixture_cfg(load_fixture("policy_native_h1"))
    assert not cfg.project_out and cfg.heads == 1 and cfg.dim == cfg.dim_head == 64
    keys = [k for k, _, _ in O.policy_param_spec(cfg)]
    assert not any("to_out" in k for k in keys) and sum("to_qkv" in k for k in keys) == cfg.depth


def test_flop_model_matches_survey():
    """0.9781 GFLOP/frame at C2/C3, 34.972 at C5, 0.1903 shipped."""
    C = O.GoTConfig
    assert abs(C(image=(84, 84), patch=(12, 12), dim=256, depth=6, heads=8).fwd_flops_per_frame() / 1e9 - 0.9781) < 1e-3
    assert abs(C(image=(224, 224), patch=(16, 16), dim=768, depth=12, heads=12, mlp_dim=3072).fwd_flops_per_frame() / 1e9 - 34.972) < 1e-2
    assert abs(C().fwd_flops_per_frame() / 1e9 - 0.1903) < 1e-3


def test_oracle_gradcheck_fp64():
    """Finite-difference check of the restatement on a tiny shape in fp64."""
    cfg = O.GoTConfig(image=(8, 8), patch=(4, 4), dim=8, depth=1, heads=2, dim_head=4, mlp_dim=16)
    p = O.make_params(O.got_param_spec(cfg, prefix=""), 3, dtype=torch.float64)
    img, _, _, _ = O.make_inputs(cfg, 2, 3, dtype=torch.float64)
    goal = torch.randn(2, cfg.dim, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    keys = ["to_patch_embedding.1.weight", "transformer.layers.0.0.fn.to_qkv.weight",
            "transformer.layers.0.1.fn.net.0.weight", "transformer.layers.0.0.norm.weight", "layer_norm.g"]
    leaves = [p[k].requires_grad_(True) for k in keys]

    def f(*ws):
        q = dict(p)
        q.update(dict(zip(keys, ws)))
        return O.got_forward(q, img, goal, cfg, prefix="")

    assert torch.autograd.gradcheck(f, leaves, eps=1e-6, atol=1e-5)


@pytest.mark.parametrize("name,kind", [("cnn_qnet_native", "qnet"), ("cnn_qnet_84", "qnet"), ("cnn_policy_native", "policy")])
def test_cnn_cases(name, kind):
    """CNN critic / actor restatement against the reference's unmodified classes."""
    fx = load_fixture(name)
    image = tuple(int(v) for v in fx["meta/image"])
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    spec = O.cnn_qnet_param_spec() if kind == "qnet" else O.cnn_policy_param_spec()
    p = _leaf_params(spec, seed)
    img, pstate, act, tgt = O.make_inputs(O.GoTConfig(image=image), batch, seed)
    if kind == "qnet":
        q1, q2 = O.cnn_qnet_forward(p, img, pstate, act)
        np.testing.assert_allclose(q1.detach().numpy(), fx["q1"], rtol=0, atol=TOL)
        np.testing.assert_allclose(q2.detach().numpy(), fx["q2"], rtol=0, atol=TOL)
        loss = O.sac_critic_loss(q1, q2, tgt)
    else:
        mean, log_std = O.cnn_policy_forward(p, img, pstate)
        np.testing.assert_allclose(mean.detach().numpy(), fx["mean"], rtol=0, atol=TOL)
        np.testing.assert_allclose(log_std.detach().numpy(), fx["log_std"], rtol=0, atol=TOL)
        loss = (mean ** 2).mean() + (log_std ** 2).mean()
    np.testing.assert_allclose(loss.item(), float(fx["loss"]), rtol=1e-5)
    loss.backward()
    check_grad_digest(fx, "g", {k: v.grad for k, v in p.items()}, rtol=5e-4, atol=2e-6)


def test_got_meanpool_case():
    """pool='mean' (GoalFormer.py:167) against the reference."""
    fx = load_fixture("got_tiny_meanpool")
    cfg = fixture_cfg(fx)
    p = _leaf_params(O.got_param_spec(cfg, prefix=""), int(fx["meta/seed"]))
    img, goal, wout, _ = got_case_inputs(fx, cfg, False)
    goal.requires_grad_(True)
    feat = O.got_forward(p, img, goal, cfg, prefix="", pool="mean")
    np.testing.assert_allclose(feat.detach().numpy(), fx["feat"], rtol=0, atol=TOL)
    (feat * wout).sum().backward()
    np.testing.assert_allclose(goal.grad.numpy(), fx["dgoal"], rtol=1e-4, atol=1e-5)
    for k, v in p.items():
        if f"gfull/{k}" in fx:
            np.testing.assert_allclose(v.grad.numpy(), fx[f"gfull/{k}"], rtol=1e-4, atol=2e-5, err_msg=k)


# ---------------------------------------------------------------- bf16 configuration (BASELINE config 5)
@pytest.mark.parametrize("name", ["got_c5_l2_bf16", "got_c5_l12_bf16", "got_84p12_bf16"])
def test_oracle_bf16_model_vs_reference(name):
    """The oracle's bf16-storage model sits as close to the reference's fp32 output as the reference's own autocast(bf16)
    run does, and within the same distance of that autocast run (fixtures: tests/golden/make_golden_bf16.py)."""
    fx = load_fixture(name)
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    params = O.make_params(O.got_param_spec(cfg, prefix=""), seed)
    img, _, _, _ = O.make_inputs(cfg, batch, seed)
    goal = torch.from_numpy(np.random.RandomState(seed + 7).standard_normal((batch, cfg.dim))).float()
    ref32, refbf = torch.from_numpy(fx["feat_fp32"]), torch.from_numpy(fx["feat_autocast_bf16"])
    np.testing.assert_allclose(O.got_forward(params, img, goal, cfg, prefix="").numpy(), ref32.numpy(), atol=5e-6)
    emu = O.got_forward_bf16(params, img, goal, cfg, prefix="")
    ref_gap = float((refbf - ref32).abs().max())
    assert float((emu - ref32).abs().max()) < 1.5 * ref_gap + 2e-3
    assert float((emu - refbf).abs().max()) < 1.5 * ref_gap + 2e-3
    assert float((emu - ref32).abs().mean()) < 5e-3


# ---------------------------------------------------------------- SURVEY 8(f4) preprocessing restatement (parity unpinned)
def test_f4_oracle_self_consistency():
    """The f4 functions restate OpenCV's published formulas (cv2 is not installed: PARITY UNPINNED); checked here: invariants and the
    bilinear resize against torch's independent implementation of the same half-pixel-centre formula."""
    rs = np.random.RandomState(0)
    d = rs.uniform(0.3, 9.0, (2, 44, 64)).astype(np.float32)
    u = O.f4_depth_to_uint8(d)
    assert u.min() == 0 and u.max() in (254, 255) and (u == np.trunc(u)).all()
    c = np.full((1, 40, 60), 137, np.float32)
    assert np.allclose(O.f4_gaussian_blur(c, 5), 137) and np.allclose(O.f4_gaussian_blur(c, 11), 137, atol=1e-4)
    assert abs(float(O.f4_gaussian_kernel(11).sum()) - 1) < 1e-6 and np.array_equal(O.f4_gaussian_kernel(5) * 16, [1, 4, 6, 4, 1])
    noise = (rs.standard_normal(d.shape) * 50).astype(np.float32)
    b = O.f4_blurring(O.f4_add_nose(u, noise))
    s = O.f4_resize_to_state(b)
    t = torch.nn.functional.interpolate(torch.from_numpy(b)[:, None], size=(128, 160), mode="bilinear", align_corners=False)[:, 0] / 255
    np.testing.assert_allclose(s, t.numpy(), atol=1e-6)
    assert O.f4_pipeline(d, noise).shape == (2, 128, 160)
