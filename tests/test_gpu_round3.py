"""GPU tests added in round 3:
  * train mode (emb-dropout live) at the HEADLINE model shape (84x84 @ 12, D256 / L6 / H8) against the oracle with the HIP Philox mask
    replayed into it -- small batch, and inside a full B = 512 batch through a loss that selects 8 scattered frames;
  * optimiser state that must follow parameters into a new home (advisor finding, round 2), frozen target networks, plain leaf
    tensors (DRL.py's log_alpha), the capturable step counter in state_dict().
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import O  # noqa: E402

OUT_TOL = 1e-4
GRAD_RTOL = 2e-3


@pytest.fixture(scope="module")
def amd():
    import dgvit_amd
    dgvit_amd.load_library()
    assert torch.cuda.is_available()
    return dgvit_amd


def _build_got(amd, cfg):
    return amd.GoT(image_size=cfg.image, patch_size=cfg.patch, num_classes=cfg.num_classes, dim=cfg.dim, depth=cfg.depth,
                   heads=cfg.heads, mlp_dim=cfg.mlp_dim, channels=1, dim_head=cfg.dim_head)


HEADLINE = dict(image=(84, 84), patch=(12, 12), dim=256, depth=6, heads=8)   # BASELINE configs C2 / C3


def _philox_mask(amd, seed, batch, cfg):
    ones = torch.ones(batch * cfg.tokens * cfg.dim, device="cuda")
    amd.functional.op_dropout_(ones, seed, 0.9)
    return (ones != 0).float().reshape(batch, cfg.tokens, cfg.dim).cpu()


def test_train_mode_at_the_headline_shape_matches_oracle(amd):
    """GoalFormer.py:160-163 in train mode at the shape bench.py runs (84x84 @ 12x12, D256 / L6 / H8 / M2048, N = 50): outputs 1e-4,
    every gradient 2e-3 against the oracle fed the same Bernoulli mask (the HIP kernel's Philox stream, extracted with the seed)."""
    cfg = O.GoTConfig(**HEADLINE)
    B, seed = 4, 23
    params = O.make_params(O.got_param_spec(cfg, prefix=""), seed)
    m = _build_got(amd, cfg)
    m.load_state_dict(params, strict=True)
    m = m.cuda().train()
    img, _, _, _ = O.make_inputs(cfg, B, seed)
    rs = np.random.RandomState(seed)
    goal = torch.from_numpy(rs.standard_normal((B, cfg.dim))).float()
    wout = torch.from_numpy(rs.standard_normal((B, cfg.dim))).float()
    torch.manual_seed(seed)
    dseed = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.manual_seed(seed)                     # the module draws the same seed from the CPU generator
    gg = goal.cuda().requires_grad_(True)
    feat = m(img.cuda(), gg)
    (feat * wout.cuda()).sum().backward()
    mask = _philox_mask(amd, dseed, B, cfg)
    assert 0.88 < mask.mean().item() < 0.92
    p = {k: v.clone().double().requires_grad_(True) for k, v in params.items()}
    go = goal.clone().double().requires_grad_(True)
    ref = O.got_forward(p, img.double(), go, cfg, drop_mask=mask.double(), prefix="")
    (ref * wout.double()).sum().backward()
    np.testing.assert_allclose(feat.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(gg.grad.cpu().numpy(), go.grad.numpy(), rtol=GRAD_RTOL, atol=2e-4 * max(1.0, go.grad.abs().max().item()))
    checked = 0
    for k, prm in m.named_parameters():
        if prm.grad is None:
            assert p[k].grad is None, k
            continue
        r = p[k].grad.numpy()
        np.testing.assert_allclose(prm.grad.cpu().numpy(), r, rtol=GRAD_RTOL, atol=2e-4 * max(1.0, np.abs(r).max()), err_msg=k)
        checked += 1
    assert checked == 4 + 11 * cfg.depth


@pytest.mark.parametrize("train", [False, True])
def test_full_batch_gradients_through_selected_frames_match_oracle(amd, train):
    """BASELINE full size (B = 512, DGViT-small): the loss weights are zero except on 8 frames scattered over the batch, so every
    parameter gradient of the full-batch HIP backward (512-frame GEMMs, split-K weight gradients over 25600 token rows) must equal
    the oracle's gradient on those 8 frames alone -- in eval mode and in train mode (the 8 frames' rows of the Philox mask)."""
    cfg = O.GoTConfig(**HEADLINE)
    B, seed = 512, 3407
    params = O.make_params(O.got_param_spec(cfg, prefix=""), seed)
    m = _build_got(amd, cfg)
    m.load_state_dict(params, strict=True)
    m = m.cuda()
    m.train(train)
    img, _, _, _ = O.make_inputs(cfg, B, seed)
    rs = np.random.RandomState(seed + 1)
    goal = torch.from_numpy(rs.standard_normal((B, cfg.dim))).float()
    sel = [0, 63, 64, 201, 255, 256, 402, 511]
    wout = torch.zeros(B, cfg.dim)
    wout[sel] = torch.from_numpy(rs.standard_normal((len(sel), cfg.dim))).float()
    torch.manual_seed(seed)
    dseed = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.manual_seed(seed)
    gg = goal.cuda().requires_grad_(True)
    feat = m(img.cuda(), gg)
    (feat * wout.cuda()).sum().backward()
    mask = _philox_mask(amd, dseed, B, cfg)[sel].double() if train else None
    p = {k: v.clone().double().requires_grad_(True) for k, v in params.items()}
    go = goal[sel].clone().double().requires_grad_(True)
    ref = O.got_forward(p, img[sel].double(), go, cfg, drop_mask=mask, prefix="")
    (ref * wout[sel].double()).sum().backward()
    np.testing.assert_allclose(feat[sel].detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    dg = gg.grad.cpu()
    np.testing.assert_allclose(dg[sel].numpy(), go.grad.numpy(), rtol=GRAD_RTOL, atol=2e-4 * max(1.0, go.grad.abs().max().item()))
    rest = torch.ones(B, dtype=torch.bool)
    rest[sel] = False
    assert float(dg[rest].abs().max()) == 0.0          # frames are independent: no gradient leaks into the other 504 goals
    for k, prm in m.named_parameters():
        if prm.grad is None:
            assert p[k].grad is None, k
            continue
        r = p[k].grad.numpy()
        np.testing.assert_allclose(prm.grad.cpu().numpy(), r, rtol=GRAD_RTOL, atol=2e-4 * max(1.0, np.abs(r).max()), err_msg=k)


# ------------------------------------------------------------------------------------------------ optimiser host logic
def _policy(amd, seed=8):
    cfg = O.GoTConfig(image=(48, 48), patch=(12, 12), dim=64, depth=1, heads=2)
    m = amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, image_size=cfg.image, patch_size=cfg.patch)
    m.load_state_dict(O.make_params(O.policy_param_spec(cfg), seed), strict=True)
    return cfg, m.eval().to("cuda")


def test_flat_adam_state_follows_parameters_into_a_new_home(amd):
    """A loose home (FlatAdam over a parameter sub-set, DRL.py:145-148) takes two steps, THEN the first soft_update re-homes the whole
    network: moments and step counts must move with the parameters (round 2 silently zeroed them and restarted the bias correction).
    Compared with torch.optim.Adam + the reference's per-parameter soft_update over four steps."""
    from dgvit_amd.optim import FlatAdam, soft_update, home_of
    cfg, a = _policy(amd)
    b, tgt, ref_t = copy.deepcopy(a), copy.deepcopy(a), copy.deepcopy(a)

    def subset(m):
        return list(m.fc1.parameters()) + list(m.fc2.parameters()) + list(m.mean_linear.parameters()) + list(m.trans.parameters())
    oa, ob = FlatAdam(subset(a), lr=3e-3), torch.optim.Adam(subset(b), lr=3e-3)
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 4, 8))

    def step():
        for m, o in ((a, oa), (b, ob)):
            for q in m.parameters():
                q.grad = None
            mean, log_std = m([img, pstate])
            ((mean ** 2).mean() + ((log_std + 1) ** 2).mean()).backward()
            o.step()
    step()
    step()
    loose = {id(h) for h in oa._homes()}
    soft_update(tgt, a, 0.05)                  # first home_of(a): every parameter moves into the module's home
    for tp, sp in zip(ref_t.parameters(), b.parameters()):
        tp.data.copy_(tp.data * 0.95 + sp.data * 0.05)
    assert {id(h) for h in oa._homes()} == {id(home_of(a))} and not (loose & {id(home_of(a))})
    step()
    step()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    for (k, pa), (_, pb) in zip(tgt.named_parameters(), ref_t.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    steps = [None if s is None else s["step"] for s in oa.state_dict()["state"]]
    assert all(s == 4 for s in steps if s is not None) and steps.count(None) == 5     # cls_token and the four mlp_head tensors never get gradients


def test_child_home_swallowed_by_parent_keeps_adam_state(amd):
    from dgvit_amd.optim import FlatAdam, flatten_parameters, home_of
    cfg, a = _policy(amd, seed=9)
    b = copy.deepcopy(a)
    oa, ob = FlatAdam([a.trans], lr=2e-3), torch.optim.Adam(b.trans.parameters(), lr=2e-3)
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 3, 9))

    def step():
        for m, o in ((a, oa), (b, ob)):
            for q in m.parameters():
                q.grad = None
            mean, _ = m([img, pstate])
            (mean ** 2).mean().backward()
            o.step()
    step()
    flatten_parameters(a)                      # the policy's home takes over the encoder's parameters
    assert home_of(a).intact() and home_of(a).exp_avg is not None
    step()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)


def test_soft_update_into_a_frozen_target(amd):
    """DRL.py:169 deep-copies the policy; a target frozen with requires_grad_(False) must keep its source's layout."""
    from dgvit_amd.optim import soft_update, hard_update
    cfg, a = _policy(amd, seed=10)
    tgt = copy.deepcopy(a).requires_grad_(False)
    with torch.no_grad():
        for q in a.parameters():
            q.add_(0.01)
    before = {k: v.detach().clone() for k, v in tgt.named_parameters()}
    soft_update(tgt, a, 0.25)
    for (k, tp), (_, sp) in zip(tgt.named_parameters(), a.named_parameters()):
        np.testing.assert_allclose(tp.detach().cpu().numpy(), (before[k] * 0.75 + sp.detach() * 0.25).cpu().numpy(), rtol=1e-6, atol=1e-7, err_msg=k)
    hard_update(tgt, a)
    for (k, tp), (_, sp) in zip(tgt.named_parameters(), a.named_parameters()):
        assert torch.equal(tp, sp), k


def test_flat_adam_takes_a_plain_leaf_tensor(amd):
    """DRL.py:120-124: log_alpha = torch.zeros(1, requires_grad=True); Adam([log_alpha])."""
    from dgvit_amd.optim import FlatAdam
    la = torch.zeros(1, device="cuda", requires_grad=True)
    lb = torch.zeros(1, device="cuda", requires_grad=True)
    oa, ob = FlatAdam([la], lr=1e-2), torch.optim.Adam([lb], lr=1e-2)
    for it in range(3):
        for t, o in ((la, oa), (lb, ob)):
            t.grad = None
            (-(t.exp() * (0.3 * it - 1.0))).sum().backward()
            o.step()
    np.testing.assert_allclose(la.detach().cpu().numpy(), lb.detach().cpu().numpy(), rtol=1e-6, atol=1e-8)


def test_capturable_state_dict_reads_the_device_step(amd):
    from dgvit_amd.optim import FlatAdam
    cfg, a = _policy(amd, seed=11)
    opt = FlatAdam([a], lr=1e-3, capturable=True)
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 2, 11))
    for _ in range(2):
        opt.zero_grad()
        mean, log_std = a([img, pstate])
        ((mean ** 2).mean() + (log_std ** 2).mean()).backward()
        opt.step()
    opt._step_dev += 5                          # what five graph replays do: only the device counter advances
    sd = opt.state_dict()
    assert {s["step"] for s in sd["state"] if s is not None} == {7} and sd["step"] == 7
    opt2 = FlatAdam([a], lr=1e-3, capturable=True)
    opt2.load_state_dict(sd)
    assert {s["step"] for s in opt2.state_dict()["state"] if s is not None} == {7}


@pytest.mark.parametrize("N,H,dh", [(50, 8, 64), (37, 8, 32), (64, 4, 64), (33, 16, 64)])
def test_pipelined_attention_forward_is_bit_identical(N, H, dh):
    """fp32 attention forward for 32 < N <= 64 and >= 2048 (frame, head) items runs the pipelined kernel (a workgroup walks several items
    and holds the next one's K / V / Q in registers while it computes); smaller launches run the one-item-per-workgroup kernel.  Same
    arithmetic in the same order: the first frames of a large batch equal the same frames run as a small batch, bit for bit."""
    import dgvit_amd
    F = dgvit_amd.functional
    big = (2048 + H - 1) // H + 3
    g = torch.Generator().manual_seed(N * 100 + H)
    qkv = torch.randn(big, N, 3 * H * dh, generator=g).cuda()
    small = 2048 // H - 1                                   # below the threshold
    o_big, l_big = F.op_attention_fwd(qkv, H, dh)
    o_small, l_small = F.op_attention_fwd(qkv[:small].contiguous(), H, dh)
    assert torch.equal(o_big[:small], o_small) and torch.equal(l_big[:small], l_small)
    # and against fp64 arithmetic
    I = H * dh
    q, k, v = (qkv[-2:].double().cpu()[..., j * I:(j + 1) * I].reshape(2, N, H, dh).permute(0, 2, 1, 3) for j in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(2, N, I)
    assert float((o_big[-2:].double().cpu() - ref).abs().max()) < 2e-5
