"""GPU: what round 4 added, through the C ABI.

* Attention without an output projection (heads == 1 and dim_head == dim: to_out = nn.Identity(), GoalFormer.py:56,66-69) against
  fixtures from the reference's own classes and against the oracle, forward and backward, eval and train mode.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import O, load_fixture, fixture_cfg, got_case_inputs, check_grad_digest  # noqa: E402

OUT_TOL = 1e-4
GRAD_RTOL = 2e-3
GRAD_ATOL = 2e-5


@pytest.fixture(scope="module")
def amd():
    import dgvit_amd
    dgvit_amd.load_library()
    assert torch.cuda.is_available()
    return dgvit_amd


def _build_got(amd, cfg):
    return amd.GoT(image_size=cfg.image, patch_size=cfg.patch, num_classes=cfg.num_classes, dim=cfg.dim, depth=cfg.depth,
                   heads=cfg.heads, mlp_dim=cfg.mlp_dim, channels=1, dim_head=cfg.dim_head)


# ------------------------------------------------------------------------------------------------ project_out == False
def test_policy_without_output_projection_golden(amd):
    """GoTPolicy(2, 2, 2, 1, 64) -- a legal reference constructor call (config.yaml:5 already sets 64) whose Attention has
    to_out = nn.Identity(): outputs and every gradient digest against the reference's unmodified class."""
    fx = load_fixture("policy_native_h1")
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    m = amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim)
    params = O.make_params(O.policy_param_spec(cfg), seed)
    assert list(m.state_dict().keys()) == list(params.keys()) and not any("to_out" in k for k in params)
    m.load_state_dict(params, strict=True)
    m = m.to("cuda").eval()
    img, pstate, _, _ = O.make_inputs(cfg, batch, seed)
    mean, log_std = m([img.cuda(), pstate.cuda()])
    np.testing.assert_allclose(mean.detach().cpu().numpy(), fx["mean"], rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(log_std.detach().cpu().numpy(), fx["log_std"], rtol=0, atol=OUT_TOL)
    loss = (mean ** 2).mean() + (log_std ** 2).mean()
    np.testing.assert_allclose(loss.item(), float(fx["loss"]), rtol=1e-4)
    loss.backward()
    check_grad_digest(fx, "g", {k: p.grad for k, p in m.named_parameters()}, rtol=GRAD_RTOL, atol=GRAD_ATOL)
    with torch.no_grad():       # the inference schedule (shared layer buffers, patch gather in the GEMM loader)
        mean2, log_std2 = m([img.cuda(), pstate.cuda()])
    np.testing.assert_allclose(mean2.cpu().numpy(), fx["mean"], rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(log_std2.cpu().numpy(), fx["log_std"], rtol=0, atol=OUT_TOL)
    assert float((mean2 - mean.detach()).abs().max()) < 1e-5


def test_got_without_output_projection_train_mode_golden(amd):
    """Bare encoder, heads 1 / dim_head 32 = dim, train mode with the reference's injected dropout mask: the HIP path cannot take a
    mask from outside, so it is checked through the oracle (pinned to this fixture at 2e-6 on the CPU): same Philox mask on both sides,
    outputs 1e-4, every gradient in full."""
    fx = load_fixture("got_tiny_h1_mask")
    cfg = fixture_cfg(fx)
    assert not cfg.project_out
    seed = int(fx["meta/seed"])
    params = O.make_params(O.got_param_spec(cfg, prefix=""), seed)
    m = _build_got(amd, cfg)
    m.load_state_dict(params, strict=True)
    m = m.cuda().train()
    img, goal, wout, _ = got_case_inputs(fx, cfg, False)
    B = img.shape[0]
    torch.manual_seed(123)
    dseed = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.manual_seed(123)                      # the module draws the same seed from the CPU generator
    gg = goal.cuda().requires_grad_(True)
    feat = m(img.cuda(), gg)
    (feat * wout.cuda()).sum().backward()
    ones = torch.ones(B * cfg.tokens * cfg.dim, device="cuda")
    from dgvit_amd import functional as F
    mask = (F.op_dropout_(ones, dseed, 0.9) > 0).float().reshape(B, cfg.tokens, cfg.dim).cpu()
    assert 0.85 < mask.mean().item() < 0.95
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    g2 = goal.clone().requires_grad_(True)
    ref = O.got_forward(p, img, g2, cfg, drop_mask=mask, prefix="")
    (ref * wout).sum().backward()
    np.testing.assert_allclose(feat.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(gg.grad.cpu().numpy(), g2.grad.numpy(), rtol=GRAD_RTOL, atol=1e-4)
    for k, q in m.named_parameters():
        if p[k].grad is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, k
            continue
        r = p[k].grad.numpy()
        np.testing.assert_allclose(q.grad.cpu().numpy(), r, rtol=GRAD_RTOL, atol=2e-5 * max(1.0, float(np.abs(r).max())), err_msg=k)


@pytest.mark.parametrize("dense_last", [False, True])
def test_without_output_projection_schedules_and_larger_batch(amd, dense_last):
    """B = 40 frames of 84x84 @ 12 (N = 50), depth 3, heads 1, dim 64: the token-0-only last block and the dense schedule both match the
    oracle (outputs 1e-4, gradients 2e-3); D = 64 also exercises the LayerNorm-in-the-GEMM-epilogue path around the missing to_out GEMM."""
    cfg = O.GoTConfig(image=(84, 84), patch=(12, 12), dim=64, depth=3, heads=1)
    params = O.make_params(O.got_param_spec(cfg, prefix=""), 9)
    m = _build_got(amd, cfg)
    m.load_state_dict(params, strict=True)
    m = m.cuda().eval().set_schedule(dense_last_block=dense_last)
    B = 40
    img, _, _, _ = O.make_inputs(cfg, B, 9)
    goal = torch.randn(B, cfg.dim, generator=torch.Generator().manual_seed(2))
    wout = torch.randn(B, cfg.dim, generator=torch.Generator().manual_seed(3))
    gg = goal.cuda().requires_grad_(True)
    feat = m(img.cuda(), gg)
    (feat * wout.cuda()).sum().backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    g2 = goal.clone().requires_grad_(True)
    ref = O.got_forward(p, img, g2, cfg, prefix="")
    (ref * wout).sum().backward()
    np.testing.assert_allclose(feat.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(gg.grad.cpu().numpy(), g2.grad.numpy(), rtol=GRAD_RTOL, atol=1e-4)
    for k, q in m.named_parameters():
        if p[k].grad is None:
            continue
        r = p[k].grad
        err = float((q.grad.cpu() - r).norm() / (r.norm() + 1e-12))
        assert err < GRAD_RTOL, (k, err)


def test_bf16_configuration_refuses_a_projectionless_attention(amd):
    cfg = O.GoTConfig(image=(84, 84), patch=(12, 12), dim=64, depth=1, heads=1)
    m = _build_got(amd, cfg).cuda().eval().set_compute_dtype(torch.bfloat16)
    with pytest.raises(NotImplementedError):
        m(torch.rand(2, 84, 84, device="cuda"), torch.randn(2, 64, device="cuda"))


# ------------------------------------------------------------------------------------------------ two optimisers, one set of parameters
def test_two_live_optimisers_over_the_same_parameters_keep_separate_state(amd):
    """Imitation_learning.py:379 builds `policy_optim` and :812 a second optim.Adam over the SAME ego.policy.parameters(); both stay
    alive.  torch keeps Adam state per optimiser, so the second one starts from empty moments and the first keeps its own: FlatAdam
    must do the same (round 3 kept the state with the parameters' home: building the second optimiser wiped the first's)."""
    import copy
    from dgvit_amd.optim import FlatAdam
    cfg = O.GoTConfig(image=(48, 48), patch=(12, 12), dim=64, depth=1, heads=2)
    a = amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, image_size=cfg.image, patch_size=cfg.patch)
    a.load_state_dict(O.make_params(O.policy_param_spec(cfg), 12), strict=True)
    a = a.eval().to("cuda")
    b = copy.deepcopy(a)
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 4, 12))
    il_a, il_b = FlatAdam(a, lr=3e-3), torch.optim.Adam(b.parameters(), lr=3e-3)

    def step(pairs):
        for m, o in pairs:
            for q in m.parameters():
                q.grad = None
            mean, log_std = m([img, pstate])
            ((mean ** 2).mean() + ((log_std + 1) ** 2).mean()).backward()
            o.step()
    step([(a, il_a), (b, il_b)])
    step([(a, il_a), (b, il_b)])
    rl_a, rl_b = FlatAdam(a.parameters(), lr=1e-3), torch.optim.Adam(b.parameters(), lr=1e-3)     # the second, while the first lives
    assert rl_a._private and not il_a._private
    step([(a, rl_a), (b, rl_b)])          # bias correction restarts at step 1 for the new optimiser ...
    step([(a, il_a), (b, il_b)])          # ... and the first continues with ITS moments at step 3
    step([(a, rl_a), (b, rl_b)])
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        # (a wiped or shared state moves a parameter by O(lr) = 1e-3 .. 3e-3; Adam's m / sqrt(v) amplifies fp32 rounding of near-zero
        #  gradients to a few 1e-5 on a handful of elements, hence the absolute tolerance)
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=1e-4, atol=1e-4, err_msg=k)
    s1 = [s["step"] for s in il_a.state_dict()["state"] if s is not None]
    s2 = [s["step"] for s in rl_a.state_dict()["state"] if s is not None]
    assert set(s1) == {3} and set(s2) == {2}
    # an optimiser whose predecessor is gone takes the home's slots over (and starts empty), as before
    del il_a, rl_a
    import gc
    gc.collect()
    c = FlatAdam(a, lr=1e-3)
    assert not c._private and all(s is None for s in c.state_dict()["state"])


# ------------------------------------------------------------------------------------------------ two launches per block (block.hip)
from helpers import knobs  # noqa: E402


def _block_path(on):
    """mode 1 (the product): the fused blocks where they win (few frames); 0: the seven-launch GEMM schedule for every size; 2: the fused
    blocks for every shape they support.  0 and 2 exist in the diagnostic library only."""
    return knobs(block_path=(2 if on else 0, 4160))


@pytest.mark.parametrize("name,cls", [("policy_native_shipped", "policy"), ("policy_native_small", "policy"), ("policy_c2", "policy"),
                                      ("detpolicy_native_shipped", "det"), ("policy_native_h1", "policy")])
def test_block_path_matches_reference_goldens(amd, name, cls):
    """no_grad forward (what SAC.choose_action runs, DRL.py:170-185) through the product library: fixtures from the reference's own
    classes within 1e-4.  policy_native_small (N = 65 at D = 256) exceeds the LDS budget of the fused kernels and policy_native_h1 has
    no output projection: both must fall through to the GEMM schedule without a trace."""
    fx = load_fixture(name)
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    kw = dict(image_size=cfg.image, patch_size=cfg.patch)
    if cls == "policy":
        m = amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw)
        m.load_state_dict(O.make_params(O.policy_param_spec(cfg), seed), strict=True)
    else:
        m = amd.DeterministicGoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw)
        m.load_state_dict(O.make_params(O.detpolicy_param_spec(cfg), seed), strict=True)
    m = m.eval().to("cuda")
    img, pstate, _, _ = O.make_inputs(cfg, batch, seed)
    with torch.no_grad():
        out = m([img.cuda(), pstate.cuda()])
    if cls == "policy":
        np.testing.assert_allclose(out[0].cpu().numpy(), fx["mean"], rtol=0, atol=OUT_TOL)
        np.testing.assert_allclose(out[1].cpu().numpy(), fx["log_std"], rtol=0, atol=OUT_TOL)
    else:
        np.testing.assert_allclose(out.cpu().numpy(), fx["mean"], rtol=0, atol=OUT_TOL)


@pytest.mark.parametrize("image,patch,dim,depth,heads,dim_head,B,pool", [
    ((128, 160), (16, 20), 64, 4, 4, 64, 1, "cls"), ((128, 160), (16, 20), 64, 4, 4, 64, 2, "cls"), ((128, 160), (16, 20), 64, 4, 4, 64, 32, "cls"),
    ((128, 160), (16, 20), 64, 2, 4, 64, 64, "cls"),  # 4160 rows: the largest batch the path takes
    ((128, 160), (16, 20), 64, 2, 4, 64, 65, "cls"),  # one frame more: the GEMM schedule
    ((84, 84), (12, 12), 256, 3, 8, 64, 3, "cls"),    # DGViT-small width, N = 50 (two key tiles)
    ((84, 84), (12, 12), 256, 2, 8, 64, 33, "cls"),   # ... 1650 rows: 52 row tiles, the last one ragged
    ((84, 84), (14, 14), 128, 2, 3, 32, 5, "cls"),    # N = 37, dim_head 32, three heads
    ((84, 84), (14, 14), 96, 2, 3, 32, 5, "cls"),     # D = 96: not a power of two -> GEMM schedule
    ((84, 84), (7, 7), 64, 1, 2, 64, 2, "cls"),       # N = 145 > 128: not eligible
    ((96, 96), (12, 12), 64, 2, 2, 64, 3, "cls"),     # N = 65 exactly three key tiles with ONE real key in the last
    ((60, 96), (12, 12), 32, 2, 1, 64, 4, "cls"),     # N = 41, D = 32 (four k-slices in the output stages), one head of 64
    ((24, 24), (24, 24), 32, 2, 2, 32, 4, "cls"),     # N = 2
    ((84, 84), (12, 12), 64, 3, 4, 64, 6, "mean"),    # pool = 'mean': every row of the last block is needed (dense)
])
def test_block_path_equals_gemm_schedule(amd, image, patch, dim, depth, heads, dim_head, B, pool):
    """Same module, same inputs, eval and train mode (same dropout seed): the two-launch blocks and the seven-launch GEMM schedule agree
    to fp32 summation-order rounding, both with the oracle (1e-4), and the fused path is bit-reproducible."""
    cfg = O.GoTConfig(image=image, patch=patch, dim=dim, depth=depth, heads=heads, dim_head=dim_head, mlp_dim=256 if dim < 64 else 2048)
    params = O.make_params(O.got_param_spec(cfg, prefix=""), 17)
    m = amd.GoT(image_size=image, patch_size=patch, num_classes=2, dim=dim, depth=depth, heads=heads, mlp_dim=cfg.mlp_dim, channels=1,
                dim_head=dim_head, pool=pool)
    m.load_state_dict(params, strict=True)
    m = m.cuda()
    img, _, _, _ = O.make_inputs(cfg, B, 17)
    goal = torch.randn(B, dim, generator=torch.Generator().manual_seed(3))
    ref = O.got_forward(params, img, goal, cfg, prefix="", pool=pool)
    outs = {}
    for mode in ("eval", "train"):
        m.train(mode == "train")
        for on in (True, False):
            with _block_path(on):
                torch.manual_seed(5)      # same dropout seed draw for both paths
                with torch.no_grad():
                    outs[(mode, on)] = m(img.cuda(), goal.cuda()).cpu()
        np.testing.assert_allclose(outs[(mode, True)].numpy(), outs[(mode, False)].numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(outs[("eval", True)].numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    assert (outs[("train", True)] - outs[("eval", True)]).abs().max().item() > 1e-3      # dropout was live in train mode
    m.eval()
    with _block_path(True), torch.no_grad():
        again = m(img.cuda(), goal.cuda()).cpu()
        assert torch.equal(again, outs[("eval", True)]), "the in-launch combines must sum in a fixed order"
        # the dense last block (A/B flag) through the fused kernels too
        m.set_schedule(dense_last_block=True)
        dense = m(img.cuda(), goal.cuda()).cpu()
    np.testing.assert_allclose(dense.numpy(), outs[("eval", True)].numpy(), rtol=0, atol=2e-5)
    # block 0 assembling its own token rows (goal row, emb-dropout with the same Philox mask, counters zeroed in the kernel): opt-in, same results
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with knobs(block_path=(2, 4160), block_fuse=3):
            torch.manual_seed(5)
            with torch.no_grad():
                fused_first = m(img.cuda(), goal.cuda()).cpu()
        np.testing.assert_allclose(fused_first.numpy(), outs[(mode, True)].numpy(), rtol=0, atol=2e-5, err_msg=mode)
    m.eval()
    # ... and the product library's own choice (fused for a few frames, the GEMM schedule otherwise) gives the same features
    m.set_schedule(dense_last_block=False)
    with torch.no_grad():
        prod = m(img.cuda(), goal.cuda()).cpu()
    np.testing.assert_allclose(prod.numpy(), outs[("eval", True)].numpy(), rtol=0, atol=2e-5)


def test_block_path_runs_two_launches_per_block(amd):
    """The product library takes the fused kernels for a single frame (profile counters of kind 'other': 1 LayerNorm + 2 per block
    + patch/goal/final kernels) and the GEMM kind sees only the patch embedding."""
    import ctypes
    from dgvit_amd import _lib
    lib = amd.load_library()
    m = amd.GoTPolicy(2, 2, 4, 4, 64).to("cuda").eval()
    img, ps = torch.rand(1, 128, 160, device="cuda"), torch.rand(1, 2, device="cuda")
    with torch.no_grad():
        m([img, ps])
        torch.cuda.synchronize()
        lib.dgvit_profile_sampling(1)
        lib.dgvit_profile_start(256)
        m.trans(img, torch.zeros(1, 64, device="cuda"))
        torch.cuda.synchronize()
    kinds = _lib.PROFILE_KINDS
    ms, work, cnt = (ctypes.c_double * kinds)(), (ctypes.c_double * kinds)(), (ctypes.c_longlong * kinds)()
    lib.dgvit_profile_stop(ms, work, cnt)
    assert cnt[0] == 1, f"{cnt[0]} GEMM launches in a single-frame encoder forward (expected the patch embedding only)"
    assert cnt[1] == 0 and cnt[2] == 0


# ------------------------------------------------------------------------------------------------ stored gelu' factor
def test_stored_gelu_derivative_is_bit_identical_to_the_backward_epilogue_form(amd):
    """The training forward's fc1 epilogue stores gelu'(t) where round 3 stored t, and fc2's data gradient multiplies by it instead of
    evaluating erf / exp per element: the same operations on the same values, so every gradient must come out bit for bit the same
    (headline shape, train mode, same dropout seed)."""
    cfg = O.GoTConfig(image=(84, 84), patch=(12, 12), dim=256, depth=2, heads=8)
    params = O.make_params(O.got_param_spec(cfg, prefix=""), 21)
    m = _build_got(amd, cfg)
    m.load_state_dict(params, strict=True)
    m = m.cuda().train()
    B = 6
    img, _, _, _ = O.make_inputs(cfg, B, 21)
    goal = torch.randn(B, cfg.dim, generator=torch.Generator().manual_seed(4)).cuda()
    wout = torch.randn(B, cfg.dim, generator=torch.Generator().manual_seed(5)).cuda()
    got = {}
    for on in (1, 0):
        with knobs(gelu_grad_store=on):
            torch.manual_seed(9)
            for q in m.parameters():
                q.grad = None
            gg = goal.clone().requires_grad_(True)
            feat = m(img.cuda(), gg)
            (feat * wout).sum().backward()
            got[on] = (feat.detach().clone(), gg.grad.clone(), {k: q.grad.clone() for k, q in m.named_parameters() if q.grad is not None})
    assert torch.equal(got[1][0], got[0][0]) and torch.equal(got[1][1], got[0][1])
    for k in got[1][2]:
        assert torch.equal(got[1][2][k], got[0][2][k]), k


# ------------------------------------------------------------------------------------------------ one-query attention (the last block's token 0)
@pytest.mark.parametrize("B,N,H,dh", [(37, 50, 8, 64), (5, 37, 4, 64), (3, 64, 2, 64), (9, 17, 4, 32), (2, 1, 1, 64), (130, 50, 8, 32), (4, 33, 1, 64),
                                      (3, 2, 3, 32), (2, 3, 2, 64), (7, 5, 1, 32), (1, 8, 5, 64), (2, 31, 2, 32), (3, 32, 3, 64), (2, 48, 2, 32), (1, 63, 3, 64),
                                      (1025, 50, 4, 64)])
def test_single_query_attention_matches_torch_and_the_tile_kernels(amd, B, N, H, dh):
    """GoalFormer.py:167 reads x[:, 0]: the last block's attention has one query row per (frame, head).  attn_q1_fwd / attn_q1_bwd (plain
    fp32 FMAs, a wave per (frame, head)) against torch autograd on the same row, and against the MFMA tile kernels they replace."""
    import ctypes
    from dgvit_amd import functional as F
    g = torch.Generator(device="cuda").manual_seed(1000 * N + B)
    I = H * dh
    qkv = torch.randn(B, N, 3 * I, device="cuda", generator=g)
    dout0 = torch.randn(B, I, device="cuda", generator=g)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())

    def run(lib):
        out = torch.full((B, N, I), 7.0, device="cuda")
        lse = torch.full((B, H, N), 7.0, device="cuda")
        dqkv = torch.full_like(qkv, 7.0)
        dout = torch.zeros(B, N, I, device="cuda")
        dout[:, 0] = dout0
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.dgvit_attention_forward_queries(ptr(qkv), ptr(out), ptr(lse), B, N, H, dh, 1, st) == 0
        assert lib.dgvit_attention_backward_queries(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), B, N, H, dh, 1, st) == 0
        torch.cuda.synchronize()
        return out, lse, dqkv

    with amd.diagnostic_library() as lib:
        out, lse, dqkv = run(lib)
        with knobs(attention_single_query=0):
            out_t, lse_t, dqkv_t = run(lib)
    # rows the one-query form must not touch
    assert torch.all(out[:, 1:] == 7.0) and torch.all(lse[:, :, 1:] == 7.0) and torch.all(dqkv[:, 1:, :I] == 7.0)
    x = qkv.detach().clone().requires_grad_(True)
    q, k, v = (t.view(B, N, H, dh).transpose(1, 2) for t in x.split(I, dim=-1))
    s = (q[:, :, :1] @ k.transpose(-1, -2)) * dh ** -0.5
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, I)
    ref.backward(dout0)
    ref_lse = torch.logsumexp(s[:, :, 0], dim=-1) * 1.4426950408889634
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(lse[:, :, 0].cpu().numpy(), ref_lse.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    want = x.grad
    got = dqkv.clone()
    got[:, 1:, :I] = 0.0                                   # (not written: no gradient reaches the other queries)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-5 * scale + 1e-7
    # ... and the tile kernels agree to the same level (different summation order)
    np.testing.assert_allclose(out[:, 0].cpu().numpy(), out_t[:, 0].cpu().numpy(), rtol=1e-5, atol=2e-6)
    got_t = dqkv_t.clone()
    got_t[:, 1:, :I] = 0.0
    assert float((got - got_t).abs().max()) <= 1e-5 * scale + 1e-7


# ------------------------------------------------------------------------------------------------ stream GEMM: who issues the LDS-DMAs
@pytest.mark.parametrize("M,N,K", [(86680 // 8, 2304, 768), (1000, 776, 200), (257, 264, 72), (5000, 3072, 768), (4099, 768, 3080)])
@pytest.mark.parametrize("epi", [0, 1, 4, 5])
def test_stream_gemm_dma_sharing_is_bit_identical_to_the_round3_schedule(amd, M, N, K, epi):
    """gemm_bf16_stream.hip: waves 0-3 issue the LDS-DMAs of their SIMD partners too (round 4).  Only WHO issues a load changes: every
    epilogue must equal the schedule in which every wave issues its own (timing variant 32768 of the diagnostic build) bit for bit,
    including ragged M / N / K (out-of-range rows of the partner's DMAs) and the extra barrier behind the epilogue."""
    from dgvit_amd import functional as F
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)

    def run():
        if epi == 5:
            return F.op_gemm_bf16(1, x, w, bias=bias, want_c2=True)
        return (F.op_gemm_bf16(epi, x, w, bias=None if epi == 4 else bias),)

    with knobs(force_diag=True, gemm_bf16_tile=256257) as lib:
        new = run()
        if epi in (0, 1):
            lib.dgvit_set_gemm_diagnostics(32768)
            try:
                old = run()
            finally:
                lib.dgvit_set_gemm_diagnostics(0)
            for a, b in zip(new, old):
                assert torch.equal(a, b)
    ref = x.float() @ w.float().t() + (0 if epi == 4 else bias)
    got = new[-1].float() if epi == 5 else new[0].float()     # (epilogue 5: c2 is the pre-activation copy)
    if epi in (1,):
        ref = torch.nn.functional.gelu(ref)
    tol = 2e-2 if epi != 4 else 1e-3
    assert float((got - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------------------------------------ split-K for the gathered (implicit-GEMM) forms
@pytest.mark.parametrize("B,H,W", [(32, 128, 160), (2, 128, 160), (7, 84, 84), (1, 29, 33)])
def test_implicit_gemm_convolutions_take_k_slices_at_small_batches(amd, B, H, W):
    """conv2 / conv3 of the CNN critic run as implicit GEMMs (window gather in the A loader); at the shipped batch 32 conv3 is 72 tiles of a
    1600-deep GEMM, now cut into k-slices inside the launch like the dense forms.  Same sums in another order: equal to the unsplit
    launch (diagnostic knob) at 1e-5, and to torch's conv2d."""
    from dgvit_amd import functional as F
    g = torch.Generator().manual_seed(B * 1000 + H)
    shapes = [(16, 1, 5, 5), (16,), (64, 16, 5, 5), (64,), (256, 64, 5, 5), (256,)]
    params = [(torch.randn(*s, generator=g) * (0.2 if len(s) > 1 else 0.05)).cuda() for s in shapes]
    img = torch.rand(B, H, W, generator=g).cuda()
    with torch.no_grad():
        feat = F.cnn_features(img, params)
        with knobs(gemm_split=0):
            feat0 = F.cnn_features(img, params)
        x = img[:, None]
        for l in range(3):
            x = torch.relu(torch.nn.functional.conv2d(x.double(), params[2 * l].double(), params[2 * l + 1].double(), stride=2))
        ref = x.mean(dim=(2, 3)).float()
    scale = max(1.0, float(ref.abs().max()))
    assert float((feat - feat0).abs().max()) <= 1e-5 * scale
    assert float((feat - ref).abs().max()) <= 1e-4 * scale


# ------------------------------------------------------------------------------------------------ weight gradients: k-slice-major block order
def test_weight_gradient_slice_major_block_order_is_bit_identical(amd):
    """gemm.hip, EPI_SPLITK: (tile, k-slice) pairs are dealt to the XCDs k-slice major (an XCD reads its slices of dY and X once) instead of
    grid (tiles, 1, slices).  Only WHICH workgroup computes a slab changes: every gradient of a training step is bit-identical."""
    cfg = O.GoTConfig(image=(84, 84), patch=(12, 12), dim=256, depth=2, heads=8)
    m = amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, image_size=cfg.image, patch_size=cfg.patch)
    m.load_state_dict(O.make_params(O.policy_param_spec(cfg), 77), strict=True)
    m = m.cuda().train()
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 96, 77))

    def run():
        for q in m.parameters():
            q.grad = None
        torch.manual_seed(5)
        mean, log_std = m([img, pstate])
        ((mean ** 2).mean() + (log_std ** 2).mean()).backward()
        torch.cuda.synchronize()
        return {k: q.grad.clone() for k, q in m.named_parameters() if q.grad is not None}

    with knobs(force_diag=True):
        a = run()
    with knobs(gemm_wgrad_slice_major=0):
        b = run()
    assert a.keys() == b.keys() and len(a) > 20
    for k in a:
        assert torch.equal(a[k], b[k]), k


# ------------------------------------------------------------------------------------------------ more than 224 tokens
@pytest.mark.parametrize("image,patch,heads,dense_last", [((224, 224), (14, 14), 2, False), ((224, 224), (14, 14), 1, True), ((136, 168), (8, 8), 4, False)])
def test_encoder_with_up_to_288_tokens_matches_the_oracle(amd, image, patch, heads, dense_last):
    """GoT on 224x224 frames with 14x14 patches has 257 tokens: the fp32 attention kernels keep K and V of one (frame, head) in LDS up to
    288 tokens (round 3 refused N > 224).  Outputs 1e-4, gradients 2e-3 against the CPU oracle, token-0 and dense last block; the third
    case (136x168 @ 8: 17 * 21 + 1 = 358 tokens) must still be refused, with the limit in the message."""
    cfg = O.GoTConfig(image=image, patch=patch, dim=64, depth=2, heads=heads, dim_head=64, mlp_dim=128)
    if cfg.tokens > 288:
        m = _build_got(amd, cfg).cuda().eval()
        with pytest.raises(Exception, match="288"):
            m(torch.rand(1, *image, device="cuda"), torch.randn(1, cfg.dim, device="cuda"))
        return
    assert cfg.tokens == 257
    params = O.make_params(O.got_param_spec(cfg, prefix=""), 21)
    m = _build_got(amd, cfg)
    m.load_state_dict(params, strict=True)
    m = m.cuda().eval().set_schedule(dense_last_block=dense_last)
    B = 3
    img, _, _, _ = O.make_inputs(cfg, B, 21)
    goal = torch.randn(B, cfg.dim, generator=torch.Generator().manual_seed(4))
    wout = torch.randn(B, cfg.dim, generator=torch.Generator().manual_seed(5))
    gg = goal.cuda().requires_grad_(True)
    feat = m(img.cuda(), gg)
    (feat * wout.cuda()).sum().backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    g2 = goal.clone().requires_grad_(True)
    ref = O.got_forward(p, img, g2, cfg, prefix="")
    (ref * wout).sum().backward()
    np.testing.assert_allclose(feat.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(gg.grad.cpu().numpy(), g2.grad.numpy(), rtol=GRAD_RTOL, atol=1e-4)
    for k, q in m.named_parameters():
        if p[k].grad is None:
            continue
        r = p[k].grad
        err = float((q.grad.cpu() - r).norm() / (r.norm() + 1e-12))
        assert err < GRAD_RTOL, (k, err)
