"""GPU tests added in round 2 for the holes the round-1 review named:
  * bf16 configuration + writers that bypass autograd's version counters (FlatAdam, soft_update, captured graphs);
  * GoTPolicy.sample / GaussianPolicy.sample value checks against the reference's own (action, log_prob) with its noise injected;
  * out-of-bounds canaries around every caller-sized buffer of the C ABI (GPU ASAN does not exist on this pool);
  * the RCCL ("nccl") branch of GradSync on a one-rank group;
  * frozen parameters (needs_input_grad), a second backward, FlatAdam on parameter sub-sets, FlatAdam + soft_update on one network.
"""
import copy
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import O, load_fixture, fixture_cfg, knobs  # noqa: E402

OUT_TOL = 1e-4


@pytest.fixture(scope="module")
def amd():
    import dgvit_amd
    dgvit_amd.load_library()
    assert torch.cuda.is_available()
    return dgvit_amd


def _load_state(module, params):
    module.load_state_dict(params, strict=True)
    return module.cuda()


def _small_got(amd, seed=3, depth=2, dim=64, heads=2, mlp=256):
    torch.manual_seed(seed)
    return amd.GoT(image_size=(48, 48), patch_size=(12, 12), num_classes=2, dim=dim, depth=depth, heads=heads, mlp_dim=mlp,
                   channels=1).cuda()


# ------------------------------------------------------------------------------------------------ bf16 + raw parameter writers
def _fresh_bf16_features(amd, like, img, goal):
    """Features of a freshly constructed bf16 module that strict-loads ``like``'s current state_dict (its pack happens now)."""
    fresh = type(like)(image_size=(48, 48), patch_size=(12, 12), num_classes=2, dim=like._cfg[4], depth=like._cfg[5], heads=like._cfg[6],
                       mlp_dim=like._cfg[8], channels=1).cuda()
    fresh.load_state_dict(like.state_dict(), strict=True)
    fresh.eval().set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        return fresh(img, goal)


def test_bf16_flat_adam_steps_reach_the_gemm_weights(amd):
    """Round-1 bug: FlatAdam writes parameters from a HIP kernel through raw pointers (no autograd version bump), and the bf16
    weight copies were keyed on version counters, so every encoder GEMM kept using the step-0 weights.  After 3 steps the
    eval-mode features must be bit-equal to those of a fresh bf16 module loaded with the updated state_dict, and must have
    moved away from the step-0 features."""
    from dgvit_amd.optim import FlatAdam
    m = _small_got(amd).eval().set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    img, goal = torch.rand(6, 48, 48, generator=g).cuda(), torch.randn(6, 64, generator=g).cuda()
    tgt = torch.randn(6, 64, generator=g).cuda()
    with torch.no_grad():
        f0 = m(img, goal).clone()
    opt = FlatAdam([m], lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        ((m(img, goal) - tgt) ** 2).mean().backward()
        opt.step()
    with torch.no_grad():
        f3 = m(img, goal)
    assert torch.equal(f3, _fresh_bf16_features(amd, m, img, goal)), "bf16 GEMM weights are stale after FlatAdam.step()"
    assert (f3 - f0).abs().max().item() > 1e-2, "features did not move: the optimiser's writes never reached the bf16 copies"
    # frozen (serving) mode: packed once, reused, and still invalidated by FlatAdam
    m.freeze_bf16_weights()
    with torch.no_grad():
        m(img, goal); n0 = m._bf16_weights.packs
        m(img, goal); m(img, goal)
        assert m._bf16_weights.packs == n0, "frozen weights must not be re-packed per forward"
    opt.zero_grad()
    ((m(img, goal) - tgt) ** 2).mean().backward()
    opt.step()
    with torch.no_grad():
        f4 = m(img, goal)
    assert m._bf16_weights.packs > n0
    assert torch.equal(f4, _fresh_bf16_features(amd, m, img, goal))


def test_bf16_target_tracks_soft_and_hard_update(amd):
    """A bf16 target network must follow flat soft_update / hard_update AND the reference's own ``param.data.copy_`` loop
    (utils.py:31-37), none of which moves a version counter."""
    from dgvit_amd.optim import soft_update, hard_update
    src = _small_got(amd, seed=1).eval()
    tgt = _small_got(amd, seed=2).eval().set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    img, goal = torch.rand(4, 48, 48, generator=g).cuda(), torch.randn(4, 64, generator=g).cuda()
    with torch.no_grad():
        f0 = tgt(img, goal).clone()
    soft_update(tgt, src, 0.5)
    with torch.no_grad():
        f1 = tgt(img, goal).clone()
    assert torch.equal(f1, _fresh_bf16_features(amd, tgt, img, goal))
    assert (f1 - f0).abs().max().item() > 1e-2
    for tp, sp in zip(tgt.parameters(), src.parameters()):       # the reference's loop, verbatim semantics
        tp.data.copy_(tp.data * 0.5 + sp.data * 0.5)
    with torch.no_grad():
        f2 = tgt(img, goal).clone()
    assert torch.equal(f2, _fresh_bf16_features(amd, tgt, img, goal))
    assert (f2 - f1).abs().max().item() > 1e-3
    hard_update(tgt, src)
    with torch.no_grad():
        f3 = tgt(img, goal)
    for tp, sp in zip(tgt.parameters(), src.parameters()):
        assert torch.equal(tp, sp)
    assert torch.equal(f3, _fresh_bf16_features(amd, src, img, goal))


def test_bf16_captured_training_step_repacks_inside_the_graph(amd):
    """A bf16 training step captured into a HIP graph: the weight pack is part of the graph, so replays train on fresh weights."""
    from dgvit_amd.optim import FlatAdam
    base = _small_got(amd, seed=5).eval().set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(2)
    img, goal = torch.rand(4, 48, 48, generator=g).cuda(), torch.randn(4, 64, generator=g).cuda()
    tgt = torch.randn(4, 64, generator=g).cuda()

    def make(model):
        opt = FlatAdam([model], lr=1e-2, capturable=True)

        def step():
            opt.zero_grad()
            loss = ((model(img, goal) - tgt) ** 2).mean()
            loss.backward()
            opt.step()
            return loss.detach()
        return step
    ma, mb = copy.deepcopy(base), copy.deepcopy(base)
    sa, sb = make(ma), make(mb)
    gs = amd.GraphedStep(sb, warmup=2)
    for _ in range(2):
        sa()
    for _ in range(3):
        la, lb = sa(), gs()
    torch.cuda.synchronize()
    assert abs(la.item() - lb.item()) <= 1e-5 * max(1.0, abs(la.item()))
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


# ------------------------------------------------------------------------------------------------ sample(): values, not shapes
class _InjectedNoise:
    """Normal.rsample replaced by mean + std * eps with the reference run's own eps (its CPU generator stream cannot be
    reproduced on the device), exactly what got_sac_network.py:242 computes."""

    def __init__(self, eps):
        self.eps, self.orig = eps, None

    def __enter__(self):
        from torch.distributions import Normal
        from dgvit_amd import sac_networks
        self.orig, self.orig_draw = Normal.rsample, sac_networks._standard_normal
        eps = self.eps
        Normal.rsample = lambda self_, sample_shape=torch.Size(): self_.loc + self_.scale * eps.to(self_.loc.device)
        sac_networks._standard_normal = lambda like: eps.to(like.device)     # the fused sample kernel's draw
        return self

    def __exit__(self, *a):
        from torch.distributions import Normal
        from dgvit_amd import sac_networks
        Normal.rsample, sac_networks._standard_normal = self.orig, self.orig_draw


@pytest.mark.parametrize("name", ["policy_native_shipped", "policy_native_small", "policy_c2"])
def test_policy_sample_values_match_reference(amd, name):
    """GoTPolicy.sample (got_sac_network.py:238-251): action, log_prob and tanh(mean) equal the reference's with its noise
    injected; independently, log_prob is re-derived from the returned action through the oracle's formula."""
    fx = load_fixture(name)
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    m = amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, image_size=cfg.image, patch_size=cfg.patch)
    m = _load_state(m, O.make_params(O.policy_param_spec(cfg), seed)).eval().to("cuda")
    img, pstate, _, _ = O.make_inputs(cfg, batch, seed)
    with _InjectedNoise(torch.from_numpy(fx["noise"])):
        action, log_prob, tmean = m.sample([img.cuda(), pstate.cuda()])
    np.testing.assert_allclose(action.detach().cpu().numpy(), fx["action"], rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(log_prob.detach().cpu().numpy(), fx["log_prob"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(tmean.detach().cpu().numpy(), fx["tanh_mean"], rtol=0, atol=OUT_TOL)
    # un-patched sample(): recover x_t = atanh(action) and evaluate the tanh-Gaussian density on the oracle's mean / log_std
    torch.manual_seed(seed)
    a2, lp2, _ = m.sample([img.cuda(), pstate.cuda()])
    rm, rl = O.policy_forward(O.make_params(O.policy_param_spec(cfg), seed), img, pstate, cfg)
    y = a2.detach().cpu().double().clamp(-1 + 1e-7, 1 - 1e-7)
    x_t = torch.atanh(y)
    std = rl.double().exp()
    ref = (-((x_t - rm.double()) ** 2) / (2 * std ** 2) - rl.double() - 0.5 * np.log(2 * np.pi) - torch.log(1 - y ** 2 + 1e-6)).sum(1, keepdim=True)
    ok = (y.abs() < 0.999).all(dim=1)          # atanh is ill-conditioned next to +-1
    assert ok.any()
    np.testing.assert_allclose(lp2.detach().cpu().numpy()[ok.numpy()], ref.numpy()[ok.numpy()], rtol=0, atol=5e-3)


def test_cnn_gaussian_policy_sample_values_match_reference(amd):
    """GaussianPolicy.sample (got_sac_network.py:310-321) against the reference's own action / log_prob."""
    fx = load_fixture("cnn_policy_native")
    image = tuple(int(v) for v in fx["meta/image"])
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    m = _load_state(amd.GaussianPolicy(2, 2), O.make_params(O.cnn_policy_param_spec(), seed)).to("cuda")
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(O.GoTConfig(image=image), batch, seed))
    with _InjectedNoise(torch.from_numpy(fx["noise"])):
        action, log_prob, tmean = m.sample([img, pstate])
    np.testing.assert_allclose(action.detach().cpu().numpy(), fx["action"], rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(log_prob.detach().cpu().numpy(), fx["log_prob"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(tmean.detach().cpu().numpy(), fx["tanh_mean"], rtol=0, atol=OUT_TOL)


def test_sac_actor_loss_through_product_sample(amd):
    """The actor loss of DRL.py:405-410 with log_pi taken from the product's sample() (round 1 re-derived it in the test)."""
    fx = load_fixture("sac_c2")
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    kw = dict(image_size=cfg.image, patch_size=cfg.patch)
    pol = _load_state(amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw), O.make_params(O.policy_param_spec(cfg), seed)).eval().to("cuda")
    crt = _load_state(amd.GoTQNetwork(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw), O.make_params(O.qnet_param_spec(cfg), seed + 1)).eval()
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, batch, seed))
    with _InjectedNoise(torch.from_numpy(fx["noise"])):
        pi, log_pi, _ = pol.sample([img, pstate])
    np.testing.assert_allclose(pi.detach().cpu().numpy(), fx["pi"], rtol=0, atol=OUT_TOL)
    np.testing.assert_allclose(log_pi.detach().cpu().numpy(), fx["log_pi"], rtol=0, atol=5e-4)
    q1p, q2p = crt([img, pstate, pi])
    loss = ((0.2 * log_pi) - torch.min(q1p, q2p)).mean()
    np.testing.assert_allclose(loss.item(), float(fx["policy_loss"]), rtol=2e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------ out-of-bounds canaries
GUARD = 4096          # guard elements in front of and behind every buffer
PATTERN = 0x7FC0DEAD  # a NaN with a recognisable payload: any write, even of a NaN, changes the bits


class Guarded:
    """payload of `n` elements of `dtype` between two guard zones filled with PATTERN; `t` is the payload view."""

    def __init__(self, n, dtype=torch.float32, fill=None):
        isz = torch.empty(0, dtype=dtype).element_size()
        gbytes = GUARD * 4
        nbytes = (n * isz + 255) // 256 * 256
        self.raw = torch.empty(gbytes + nbytes + gbytes, dtype=torch.uint8, device="cuda")
        assert self.raw.data_ptr() % 256 == 0
        self.raw.view(torch.int32).fill_(PATTERN)
        self.t = self.raw[gbytes:gbytes + n * isz].view(dtype)
        self.lo, self.hi = self.raw[:gbytes].view(torch.int32), self.raw[gbytes + n * isz:].view(torch.uint8)
        self.hi_ref = self.hi.clone()
        if fill is not None:
            self.t.copy_(fill)

    def check(self, what):
        assert bool((self.lo == PATTERN).all()), f"{what}: wrote in FRONT of the buffer"
        assert torch.equal(self.hi, self.hi_ref), f"{what}: wrote BEHIND the buffer"


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


CANARY_SHAPES = [
    # (image, patch, dim, depth, heads, dim_head, mlp, batch): ragged token counts, widths that are not tile multiples
    ((84, 84), (12, 12), 64, 2, 2, 64, 256, 3),
    ((36, 60), (12, 12), 36, 1, 3, 32, 100, 5),
    ((84, 84), (6, 6), 128, 1, 2, 64, 388, 2),     # N = 197
    ((24, 24), (24, 24), 8, 2, 1, 64, 12, 7),      # N = 2, tiny everything
]


@pytest.mark.parametrize("shape", CANARY_SHAPES)
def test_canaries_encoder_fp32(amd, shape):
    """dgvit_got_forward / dgvit_got_backward with every caller-sized buffer (workspace, scratch, feat, dgoal, gradients)
    fenced by guard zones: the size queries must cover everything the kernels write."""
    from dgvit_amd import _lib
    lib = _lib.load()
    image, patch, dim, depth, heads, dh, mlp, B = shape
    torch.manual_seed(0)
    m = amd.GoT(image_size=image, patch_size=patch, num_classes=2, dim=dim, depth=depth, heads=heads, mlp_dim=mlp, channels=1,
                dim_head=dh).cuda()
    cfg = _lib.dgvit_config(*m._cfg)
    params = [p.detach().contiguous() for p in m.param_table()]
    table = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
    img, goal = torch.rand(B, *image, device="cuda"), torch.randn(B, dim, device="cuda")
    for save in (0, 1):
        nws = lib.dgvit_got_workspace_floats(ctypes.byref(cfg), B, save)
        ws, feat = Guarded(nws), Guarded(B * dim)
        rc = lib.dgvit_got_forward(ctypes.byref(cfg), table, _p(img), _p(goal), _p(feat.t), _p(ws.t), nws, B, save, 0.9, 1234, None, _st())
        _lib.check(rc, "dgvit_got_forward")
        torch.cuda.synchronize()
        ws.check(f"forward workspace (save={save})"); feat.check("feat")
        assert torch.isfinite(feat.t).all()
    # one float short must be refused, not overrun
    rc = lib.dgvit_got_forward(ctypes.byref(cfg), table, _p(img), _p(goal), _p(feat.t), _p(ws.t), nws - 1, B, 1, 0.9, 1234, None, _st())
    assert rc == -4
    nsc = lib.dgvit_got_backward_scratch_floats(ctypes.byref(cfg), B)
    sc, dgoal = Guarded(nsc), Guarded(B * dim)
    gr = [Guarded(p.numel()) for p in params]
    gtable = (ctypes.c_void_p * len(params))(*[g.t.data_ptr() for g in gr])
    dfeat = torch.randn(B, dim, device="cuda")
    rc = lib.dgvit_got_backward(ctypes.byref(cfg), table, gtable, _p(dfeat), _p(dgoal.t), _p(ws.t), nws, _p(sc.t), nsc, B, 0.9, 1234, None, _st())
    _lib.check(rc, "dgvit_got_backward")
    torch.cuda.synchronize()
    sc.check("backward scratch"); dgoal.check("dgoal"); ws.check("workspace during backward")
    for i, g in enumerate(gr):
        g.check(f"gradient {i}")
        assert torch.isfinite(g.t).all(), f"gradient {i} has non-finite entries (partly unwritten?)"
    # the raw call did the real thing: in eval mode (keep = 1) it reproduces the product module bit for bit
    nws0 = lib.dgvit_got_workspace_floats(ctypes.byref(cfg), B, 0)
    ws0, feat0 = Guarded(nws0), Guarded(B * dim)
    _lib.check(lib.dgvit_got_forward(ctypes.byref(cfg), table, _p(img), _p(goal), _p(feat0.t), _p(ws0.t), nws0, B, 0, 1.0, 0, None, _st()),
               "dgvit_got_forward")
    with torch.no_grad():
        assert torch.equal(m.eval()(img, goal).reshape(-1), feat0.t)


@pytest.mark.parametrize("shape", [((48, 48), (12, 12), 64, 2, 2, 64, 256, 3), ((80, 112), (16, 16), 72, 1, 3, 64, 200, 2)])
def test_canaries_encoder_bf16(amd, shape):
    from dgvit_amd import _lib
    lib = _lib.load()
    image, patch, dim, depth, heads, dh, mlp, B = shape
    torch.manual_seed(0)
    m = amd.GoT(image_size=image, patch_size=patch, num_classes=2, dim=dim, depth=depth, heads=heads, mlp_dim=mlp, channels=1).cuda()
    cfg = _lib.dgvit_config(*m._cfg)
    params = [p.detach().contiguous() for p in m.param_table()]
    table = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
    img, goal = torch.rand(B, *image, device="cuda"), torch.randn(B, dim, device="cuda")
    nwp = lib.dgvit_got_bf16_weight_elems(ctypes.byref(cfg))
    wp = Guarded(nwp, torch.bfloat16)
    _lib.check(lib.dgvit_got_pack_weights_bf16(ctypes.byref(cfg), table, _p(wp.t), nwp, 1, _st()), "pack")
    torch.cuda.synchronize()
    wp.check("bf16 weight arena")
    for save in (0, 1):
        nws = lib.dgvit_got_bf16_workspace_bytes(ctypes.byref(cfg), B, save)
        ws, feat = Guarded(nws, torch.uint8), Guarded(B * dim)
        rc = lib.dgvit_got_forward_bf16(ctypes.byref(cfg), table, _p(wp.t), _p(img), _p(goal), _p(feat.t), _p(ws.t), nws, B, save, 0.9, 77,
                                        None, _st())
        _lib.check(rc, "dgvit_got_forward_bf16")
        torch.cuda.synchronize()
        ws.check(f"bf16 workspace (save={save})"); feat.check("feat"); wp.check("weight arena during forward")
        assert torch.isfinite(feat.t).all()
    nsc = lib.dgvit_got_bf16_backward_scratch_bytes(ctypes.byref(cfg), B)
    sc, dgoal = Guarded(nsc, torch.uint8), Guarded(B * dim)
    gr = [Guarded(p.numel()) for p in params]
    gtable = (ctypes.c_void_p * len(params))(*[g.t.data_ptr() for g in gr])
    dfeat = torch.randn(B, dim, device="cuda")
    rc = lib.dgvit_got_backward_bf16(ctypes.byref(cfg), table, _p(wp.t), gtable, _p(dfeat), _p(dgoal.t), _p(img), _p(ws.t), nws, _p(sc.t), nsc, B,
                                     0.9, 77, None, _st())
    _lib.check(rc, "dgvit_got_backward_bf16")
    torch.cuda.synchronize()
    sc.check("bf16 backward scratch"); dgoal.check("dgoal"); ws.check("bf16 workspace during backward")
    for i, g in enumerate(gr):
        g.check(f"gradient {i}")
        assert torch.isfinite(g.t).all()


@pytest.mark.parametrize("B,H,W", [(3, 128, 160), (2, 84, 84), (5, 29, 33)])
def test_canaries_cnn_stack(amd, B, H, W):
    from dgvit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(0)
    shapes = [(16, 1, 5, 5), (16,), (64, 16, 5, 5), (64,), (256, 64, 5, 5), (256,)]
    params = [torch.randn(*s, device="cuda") * 0.1 for s in shapes]
    table = (ctypes.c_void_p * 6)(*[p.data_ptr() for p in params])
    img = torch.rand(B, H, W, device="cuda")
    nws, nsc = lib.dgvit_cnn_workspace_floats(B, H, W), lib.dgvit_cnn_forward_scratch_floats(B, H, W)
    ws, sc, feat = Guarded(nws), Guarded(nsc), Guarded(B * 256)
    _lib.check(lib.dgvit_cnn_forward(_p(img), table, _p(feat.t), _p(ws.t), nws, _p(sc.t), nsc, B, H, W, _st()), "dgvit_cnn_forward")
    torch.cuda.synchronize()
    ws.check("cnn workspace"); sc.check("cnn forward scratch"); feat.check("cnn feat")
    nsb = lib.dgvit_cnn_backward_scratch_floats(B, H, W)
    sb = Guarded(nsb)
    gr = [Guarded(p.numel()) for p in params]
    gtable = (ctypes.c_void_p * 6)(*[g.t.data_ptr() for g in gr])
    dfeat = torch.randn(B, 256, device="cuda")
    _lib.check(lib.dgvit_cnn_backward(_p(img), table, gtable, _p(dfeat), _p(ws.t), nws, _p(sb.t), nsb, B, H, W, _st()), "dgvit_cnn_backward")
    torch.cuda.synchronize()
    sb.check("cnn backward scratch"); ws.check("cnn workspace during backward")
    for i, g in enumerate(gr):
        g.check(f"cnn gradient {i}")
        assert torch.isfinite(g.t).all()


@pytest.mark.parametrize("M,N,K,act", [(1, 2, 128, 0), (33, 130, 66, 1), (512, 128, 258, 1), (7, 32, 3, 0), (2050, 2, 128, 0)])
def test_canaries_linear(amd, M, N, K, act):
    from dgvit_amd import _lib
    lib = _lib.load()
    torch.manual_seed(0)
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    y = Guarded(M * N)
    _lib.check(lib.dgvit_linear_forward(_p(x), _p(w), _p(b), _p(y.t), M, N, K, act, _st()), "dgvit_linear_forward")
    torch.cuda.synchronize()
    y.check("linear y")
    ref = torch.nn.functional.linear(x.double().cpu(), w.double().cpu(), b.double().cpu())
    ref = ref.clamp_min(0) if act else ref
    np.testing.assert_allclose(y.t.view(M, N).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    nsc = lib.dgvit_linear_backward_scratch_floats(M, N, K)
    sc, dx, dw, db = Guarded(nsc), Guarded(M * K), Guarded(N * K), Guarded(N)
    dy = torch.randn(M, N, device="cuda")
    _lib.check(lib.dgvit_linear_backward(_p(dy), _p(x), _p(w), _p(y.t), _p(dx.t), _p(dw.t), _p(db.t), _p(sc.t), nsc, M, N, K, act, _st()),
               "dgvit_linear_backward")
    torch.cuda.synchronize()
    for g, what in ((sc, "scratch"), (dx, "dx"), (dw, "dw"), (db, "db")):
        g.check(f"linear backward {what}")
    assert torch.isfinite(dx.t).all() and torch.isfinite(dw.t).all() and torch.isfinite(db.t).all()


# ------------------------------------------------------------------------------------------------ RCCL on one rank
@pytest.mark.parametrize("overlap", [False, True], ids=["after-backward", "overlapped"])
def test_gradsync_collective_branch_on_a_one_rank_rccl_group(amd, overlap):
    """backend "nccl" is RCCL on ROCm.  A one-rank group makes all_reduce the identity, so the collective branch of
    GradSync.sync() (buckets over the fused backward's flat gradient buffer, then /world) can run on this one-GPU box:
    gradients must come back unchanged, and FlatAdam must still pick the buffer up in place."""
    import torch.distributed as dist
    from dgvit_amd.parallel import GradSync
    from dgvit_amd.optim import FlatAdam, home_of
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29000 + os.getpid() % 2000)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        cfg = O.GoTConfig(image=(84, 84), patch=(12, 12), dim=64, depth=2, heads=2)
        kw = dict(image_size=cfg.image, patch_size=cfg.patch)
        m = _load_state(amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw), O.make_params(O.policy_param_spec(cfg), 9)).eval().to("cuda")
        img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 8, 9))
        opt = FlatAdam([m], lr=1e-3)
        # small buckets: several collectives per buffer; overlapped: ReduceOp.AVG all-reduces queued from inside the backward
        sync = GradSync([m], force_collective=True, bucket_bytes=64 << 10, overlap=overlap)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        sync.broadcast_parameters(0)
        for k, p in m.named_parameters():
            assert torch.equal(p, before[k]), k
        sync.zero_grad()
        mean, log_std = m([img, pstate])
        ((mean ** 2).mean() + (log_std ** 2).mean()).backward()
        assert sync.early_launches >= 2 * 18 if overlap else sync.early_launches == 0
        torch.cuda.synchronize()      # (the early all-reduces of a one-rank group are identities: the copy below sees final values)
        ref = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        sync.sync()
        torch.cuda.synchronize()
        assert sync.grad_numel() == sum(g.numel() for g in ref.values())
        for k, p in m.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, ref[k]), f"{k}: gradient changed by a one-rank all-reduce"
        opt.step()
        assert home_of(m).zero_copy_elems > 0
        torch.cuda.synchronize()
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ frozen parameters, re-entry
@pytest.mark.parametrize("bf16", [False, True])
def test_frozen_encoder_skips_weight_gradients(amd, bf16):
    """The reference's heads-only optimiser (DRL.py:145-148) with the encoder frozen (requires_grad off): encoder gradients are
    None, head gradients and the gradient w.r.t. the goal embedding equal the unfrozen run's, and the C ABI is handed NULL
    gradient slots (no weight-gradient GEMMs)."""
    cfg = O.GoTConfig(image=(48, 48), patch=(12, 12), dim=64, depth=2, heads=2)
    kw = dict(image_size=cfg.image, patch_size=cfg.patch)
    a = _load_state(amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw), O.make_params(O.policy_param_spec(cfg), 4)).eval().to("cuda")
    b = copy.deepcopy(a)
    if bf16:
        a.trans.set_compute_dtype(torch.bfloat16); b.trans.set_compute_dtype(torch.bfloat16)
    for p in b.trans.parameters():
        p.requires_grad_(False)
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 6, 4))
    for m in (a, b):
        mean, log_std = m([img, pstate])
        ((mean ** 2).mean() + (log_std ** 2).mean()).backward()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if k.startswith("trans."):
            assert pb.grad is None, k
        elif pa.grad is not None:
            assert torch.equal(pa.grad, pb.grad), k      # same kernels on the same data: bit identical
    assert b.fc_embed.weight.grad is not None
    # a partly frozen encoder: only the LayerNorm parameters train
    c = copy.deepcopy(a)
    for k, p in c.trans.named_parameters():
        p.requires_grad_(".norm." in k)
    mean, log_std = c([img, pstate])
    ((mean ** 2).mean() + (log_std ** 2).mean()).backward()
    for (k, pa), (_, pc) in zip(a.named_parameters(), c.named_parameters()):
        if k.startswith("trans.") and ".norm." not in k:
            assert pc.grad is None, k
        elif pa.grad is not None:
            assert torch.equal(pa.grad, pc.grad), k


def test_second_backward_raises_a_clear_error(amd):
    m = _small_got(amd).eval()
    img, goal = torch.rand(2, 48, 48, device="cuda"), torch.randn(2, 64, device="cuda", requires_grad=True)
    f = m(img, goal).sum()
    f.backward(retain_graph=True)
    with pytest.raises(amd.DgvitError, match="second time"):
        f.backward()


# ------------------------------------------------------------------------------------------------ optimiser host logic
def test_flat_adam_and_soft_update_share_one_home(amd):
    """bench.py's pattern: flatten_parameters(critic), FlatAdam([critic]), soft_update(target, critic).  Round 1 re-homed the
    parameters twice and silently fell back to a Python loop; now there is one flat buffer and one Polyak kernel."""
    from dgvit_amd.optim import FlatAdam, flatten_parameters, soft_update, home_of
    torch.manual_seed(0)
    crt = amd.GoTQNetwork(2, 2, 1, 2, 64, image_size=(48, 48), patch_size=(12, 12)).to("cuda").eval()   # eval: no dropout draw
    tgt = copy.deepcopy(crt)
    ref_c, ref_t = copy.deepcopy(crt), copy.deepcopy(crt)
    flat = flatten_parameters(crt)
    flatten_parameters(tgt)
    opt = FlatAdam([crt], lr=1e-3)
    assert home_of(crt).flat.data_ptr() == flat.data_ptr() and home_of(crt).intact() and home_of(tgt).intact()
    ropt = torch.optim.Adam(ref_c.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(0)
    img, ps, act = torch.rand(4, 48, 48, generator=g).cuda(), torch.rand(4, 2, generator=g).cuda(), torch.rand(4, 2, generator=g).cuda()
    for _ in range(2):
        for net, o in ((crt, opt), (ref_c, ropt)):
            o.zero_grad()
            q1, q2 = net([img, ps, act])
            ((q1 ** 2).mean() + (q2 ** 2).mean()).backward()
            o.step()
        soft_update(tgt, crt, 0.05)
        for tp, sp in zip(ref_t.parameters(), ref_c.parameters()):
            tp.data.copy_(tp.data * 0.95 + sp.data * 0.05)
    assert home_of(crt).intact() and home_of(tgt).intact()
    for (k, a), (_, b) in zip(tgt.named_parameters(), ref_t.named_parameters()):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    assert bool(torch.isfinite(home_of(crt).exp_avg).all()) and bool(torch.isfinite(home_of(crt).exp_avg_sq).all())


def test_flat_adam_parameter_subsets_and_late_gradients(amd):
    """torch.optim.Adam semantics the reference relies on: an optimiser over a sub-set of a network's parameters
    (DRL.py:145-148), parameters without a gradient are skipped, a parameter that first gets a gradient at a later step
    starts its own bias correction then."""
    from dgvit_amd.optim import FlatAdam
    cfg = O.GoTConfig(image=(48, 48), patch=(12, 12), dim=64, depth=1, heads=2)
    kw = dict(image_size=cfg.image, patch_size=cfg.patch)
    a = _load_state(amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw), O.make_params(O.policy_param_spec(cfg), 8)).eval().to("cuda")
    b = copy.deepcopy(a)

    def subset(m):
        return list(m.fc1.parameters()) + list(m.fc2.parameters()) + list(m.mean_linear.parameters()) + list(m.log_std_linear.parameters())
    oa, ob = FlatAdam(subset(a), lr=3e-3), torch.optim.Adam(subset(b), lr=3e-3)
    img, pstate, _, _ = (t.cuda() for t in O.make_inputs(cfg, 4, 8))
    for it in range(4):
        for m, o in ((a, oa), (b, ob)):
            for p in m.parameters():
                p.grad = None
            mean, log_std = m([img, pstate])
            # steps 0-1: the loss ignores log_std (its Linear gets no gradient); it joins from step 2 on
            loss = (mean ** 2).mean() if it < 2 else (mean ** 2).mean() + ((log_std + 1) ** 2).mean()
            loss.backward()
            o.step()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    sd = oa.state_dict()
    steps = [None if s is None else s["step"] for s in sd["state"]]
    assert steps == [4, 4, 4, 4, 4, 4, 2, 2]
    oa2 = FlatAdam(subset(a), lr=3e-3)
    oa2.load_state_dict(sd)
    assert [None if s is None else s["step"] for s in oa2.state_dict()["state"]] == steps


# ------------------------------------------------------------------------------------------------ in-launch split-K GEMM
@pytest.mark.parametrize("layout,epi,M,N,K", [
    (0, 0, 2080, 64, 2048),     # shipped model, B = 32: 33 tiles of 64 x 64 against 1024 workgroup slots
    (0, 0, 65, 64, 2048),       # single frame
    (0, 0, 33, 64, 300),        # ragged M and K, few k-tiles per slice
    (1, 0, 2080, 64, 2048),     # data gradient of fc1
    (0, 3, 130, 128, 1000),     # ReLU epilogue on the split path
    (1, 2, 200, 64, 1024),      # GELU' epilogue (aux read by the last arriver)
    (0, 0, 25600, 256, 2048),   # C3: 1600 tiles = 6.25 per CU, only the last 64 tiles are split
    (1, 0, 25600, 256, 1536),
])
def test_gemm_in_launch_split_k(amd, layout, epi, M, N, K):
    """Partial tiles + last-arriver epilogue must give the unsplit kernel's result (to fp32 summation-order rounding), the same
    bits on every run, and leave the arrival counters zero (a second call on the same scratch works)."""
    from dgvit_amd import functional as F, _lib
    lib = _lib.load()
    assert lib.dgvit_gemm_scratch_floats(layout, M, N, K) > 0, "this shape is supposed to take the split path"
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) if layout == 0 else torch.randn(K, N, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda() if layout == 0 else None
    res = torch.randn(M, N, generator=g).cuda() if (layout == 0 and epi == 0) else None
    aux = torch.randn(M, N, generator=g).cuda() if epi == 2 else None
    run = lambda: F.op_gemm(layout, epi, A, B, M, N, K, bias=bias, res=res, aux=aux)
    y1, y2 = run(), run()
    assert torch.equal(y1, y2), "split-K result is not reproducible"
    with knobs(gemm_split=0):       # the unsplit kernel: A/B knob of the diagnostic library
        y0 = run()
    ref = A.double().cpu() @ (B.double().cpu().T if layout == 0 else B.double().cpu())
    if bias is not None:
        ref = ref + bias.double().cpu()
    if res is not None:
        ref = ref + res.double().cpu()
    if epi == 3:
        ref = ref.clamp_min(0)
    if epi == 2:
        a = aux.double().cpu()
        ref = ref * (0.5 * (1 + torch.erf(a / 2 ** 0.5)) + a * torch.exp(-0.5 * a * a) / (2 * np.pi) ** 0.5)
    tol = 2e-4 * K ** 0.5
    assert (y1.double().cpu() - ref).abs().max().item() <= tol
    assert (y0.double().cpu() - ref).abs().max().item() <= tol
    assert (y1 - y0).abs().max().item() <= 1e-5 * K ** 0.5 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------------------------------------ fused MLP heads
def _ref_head(xs, towers):
    x = torch.cat(xs, dim=1)
    out = []
    for (w1, b1, w2, b2, l3) in towers:
        h = torch.relu(torch.relu(x @ w1.T + b1) @ w2.T + b2)
        out.append([h @ w3.T + b3 for (w3, b3) in l3])
    return out


@pytest.mark.parametrize("B,ks,n1,n2,n3,towers,heads3", [
    (1, (64,), 128, 128, 2, 1, 2),          # shipped policy head, single frame
    (32, (64,), 128, 128, 2, 1, 2),         # shipped policy head, training batch
    (33, (64, 2), 128, 32, 2, 2, 1),        # GoTQNetwork(l_f_size 64): cat(feat, a) -> twin towers; 2 row blocks, K0 = 66 (unaligned)
    (512, (256,), 128, 128, 2, 1, 2),       # C3 actor head
    (70, (256, 32, 2), 128, 32, 2, 2, 1),   # CNN QNetwork: cat(conv feat, goal embedding, action), K0 = 290
    (5, (256, 32), 128, 32, 2, 1, 2),       # CNN GaussianPolicy
    (40, (64,), 128, 32, 2, 1, 1),          # DeterministicGoTPolicy
    (3, (20, 12, 1), 32, 64, 4, 2, 2),      # odd widths everywhere
])
def test_fused_mlp_head_forward_backward(amd, B, ks, n1, n2, n3, towers, heads3):
    """dgvit_mlp_head_forward / _backward against fp64 PyTorch arithmetic: outputs, input gradients (per concatenated piece) and
    every parameter gradient, for every head shape of the reference's networks."""
    from dgvit_amd import functional as F
    g = torch.Generator().manual_seed(B + sum(ks) + n2)
    K0 = sum(ks)
    xs = [torch.randn(B, k, generator=g) for k in ks]
    tw = []
    for _ in range(towers):
        w1, b1 = torch.randn(n1, K0, generator=g) / K0 ** 0.5, torch.randn(n1, generator=g) * 0.1
        w2, b2 = torch.randn(n2, n1, generator=g) / n1 ** 0.5, torch.randn(n2, generator=g) * 0.1
        l3 = [(torch.randn(n3, n2, generator=g) / n2 ** 0.5, torch.randn(n3, generator=g) * 0.1) for _ in range(heads3)]
        tw.append((w1, b1, w2, b2, l3))
    dy = torch.randn(towers, heads3, B, n3, generator=g)
    # reference in fp64
    rx = [x.double().requires_grad_(True) for x in xs]
    rt = [(w1.double().requires_grad_(True), b1.double().requires_grad_(True), w2.double().requires_grad_(True), b2.double().requires_grad_(True),
           [(w3.double().requires_grad_(True), b3.double().requires_grad_(True)) for w3, b3 in l3]) for w1, b1, w2, b2, l3 in tw]
    ry = _ref_head(rx, rt)
    sum((ry[t][j] * dy[t, j].double()).sum() for t in range(towers) for j in range(heads3)).backward()
    # HIP
    def lin(w, b):
        m = torch.nn.Linear(w.shape[1], w.shape[0])
        m.weight.data.copy_(w); m.bias.data.copy_(b)
        return m.cuda()
    mods = [(lin(w1, b1), lin(w2, b2), [lin(w3, b3) for w3, b3 in l3]) for w1, b1, w2, b2, l3 in tw]
    hx = [x.cuda().requires_grad_(True) for x in xs]
    assert F.mlp_head_supported(hx, mods)
    y = F.mlp_head(hx, mods)
    assert len(y) == towers and all(len(r) == heads3 and r[0].shape == (B, n3) for r in y)
    sum((y[t][j] * dy[t, j].cuda()).sum() for t in range(towers) for j in range(heads3)).backward()
    for t in range(towers):
        for j in range(heads3):
            np.testing.assert_allclose(y[t][j].detach().cpu().numpy(), ry[t][j].detach().numpy(), rtol=0, atol=2e-5)
    scale = max(1.0, B ** 0.5)
    for a, b in zip(hx, rx):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=0, atol=5e-5)
    for t in range(towers):
        l1, l2, l3s = mods[t]
        r = rt[t]
        for ours, ref in ((l1.weight, r[0]), (l1.bias, r[1]), (l2.weight, r[2]), (l2.bias, r[3])):
            np.testing.assert_allclose(ours.grad.cpu().numpy(), ref.grad.numpy(), rtol=0, atol=5e-5 * scale)
        for j in range(heads3):
            np.testing.assert_allclose(l3s[j].weight.grad.cpu().numpy(), r[4][j][0].grad.numpy(), rtol=0, atol=5e-5 * scale)
            np.testing.assert_allclose(l3s[j].bias.grad.cpu().numpy(), r[4][j][1].grad.numpy(), rtol=0, atol=5e-5 * scale)


def test_fused_mlp_head_frozen_inputs_and_canaries(amd):
    """Raw C-ABI call with guard zones around h1, h2, y, the input gradients, the parameter gradients and the scratch; NULL slots
    (frozen parameters / inputs without gradient) are skipped."""
    from dgvit_amd import _lib
    lib = _lib.load()
    B, ks, n1, n2, n3, towers, heads3 = 45, (64, 2), 128, 32, 2, 2, 1
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(B, k, generator=g).cuda() for k in ks]
    K0 = sum(ks)
    shapes = [(n1, K0), (n1,), (n2, n1), (n2,), (n3, n2), (n3,)] * towers
    params = [torch.randn(*s, generator=g).cuda() * 0.1 for s in shapes]
    d = _lib.dgvit_mlp_desc()
    d.batch, d.nseg, d.n1, d.n2, d.n3, d.towers, d.heads3 = B, len(ks), n1, n2, n3, towers, heads3
    for i, x in enumerate(xs):
        d.kx[i], d.ldx[i] = x.shape[1], x.stride(0)
    tab = lambda ts: (ctypes.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])
    h1, h2, y = Guarded(towers * B * n1), Guarded(towers * B * n2), Guarded(towers * heads3 * B * n3)
    _lib.check(lib.dgvit_mlp_head_forward(ctypes.byref(d), tab(xs), tab(params), _p(h1.t), _p(h2.t), _p(y.t), _st()), "fwd")
    torch.cuda.synchronize()
    for gd, what in ((h1, "h1"), (h2, "h2"), (y, "y")):
        gd.check(what)
        assert torch.isfinite(gd.t).all()
    nsc = lib.dgvit_mlp_head_backward_scratch_floats(ctypes.byref(d))
    sc = Guarded(nsc)
    dins = [Guarded(B * ks[0]), None]                     # no gradient for the action piece
    dpars = [Guarded(p.numel()) for p in params]
    dpars[0] = None                                       # tower 0's first weight frozen
    dpars[7] = None                                       # tower 1's first bias frozen
    dy = [torch.randn(B, n3, generator=g).cuda() for _ in range(towers * heads3)]
    rc = lib.dgvit_mlp_head_backward(ctypes.byref(d), tab(xs), tab(params), _p(h1.t), _p(h2.t), tab(dy), tab([None if q is None else q.t for q in dins]),
                                     tab([None if q is None else q.t for q in dpars]), _p(sc.t), nsc, _st())
    _lib.check(rc, "bwd")
    torch.cuda.synchronize()
    sc.check("scratch")
    for q in dins + dpars:
        if q is not None:
            q.check("gradient")
            assert torch.isfinite(q.t).all()


@pytest.mark.parametrize("B,A,scalar", [(1, 2, True), (37, 2, True), (300, 3, False)])
def test_fused_tanh_gaussian_sample(amd, B, A, scalar):
    """dgvit_tanh_gaussian_forward / _backward against the reference's own op sequence (got_sac_network.py:238-251) in fp64
    autograd, including log_std values outside the clamp range and per-action scale / bias."""
    from dgvit_amd import functional as F
    g = torch.Generator().manual_seed(B)
    mean, raw, eps = torch.randn(B, A, generator=g), torch.randn(B, A, generator=g) * 3 - 1, torch.randn(B, A, generator=g)
    raw[0, 0], raw[-1, -1] = 5.0, -30.0
    scale = torch.tensor(1.0) if scalar else torch.rand(A, generator=g) + 0.5
    bias = torch.tensor(0.0) if scalar else torch.randn(A, generator=g)
    w = [torch.randn(B, A, generator=g), torch.randn(B, 1, generator=g), torch.randn(B, A, generator=g)]
    m, r = mean.double().requires_grad_(True), raw.double().requires_grad_(True)
    ls = torch.clamp(r, min=-20, max=2)
    normal = torch.distributions.Normal(m, ls.exp())
    x_t = m + ls.exp() * eps.double()
    y_t = torch.tanh(x_t)
    act = y_t * scale.double() + bias.double()
    lp = (normal.log_prob(x_t) - torch.log(scale.double() * (1 - y_t.pow(2)) + 1e-6)).sum(1, keepdim=True)
    tm = torch.tanh(m) * scale.double() + bias.double()
    (act * w[0].double() + tm * w[2].double()).sum().add((lp * w[1].double()).sum()).backward()
    hm, hr = mean.cuda().requires_grad_(True), raw.cuda().requires_grad_(True)
    a2, lp2, tm2 = F.tanh_gaussian_sample(hm, hr, eps.cuda(), scale.cuda(), bias.cuda(), -20, 2)
    ((a2 * w[0].cuda()).sum() + (lp2 * w[1].cuda()).sum() + (tm2 * w[2].cuda()).sum()).backward()
    np.testing.assert_allclose(a2.detach().cpu().numpy(), act.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(tm2.detach().cpu().numpy(), tm.detach().numpy(), rtol=0, atol=2e-6)
    # where tanh saturates, 1 - y^2 is a few fp32 ulps against the 1e-6 epsilon (ill-conditioned in the reference's own fp32
    # arithmetic, got_sac_network.py:248): strict comparison on rows with every |y| < 0.999, a loose bound on the others
    ok = (y_t.detach().abs() < 0.999).all(dim=1).numpy()
    assert ok.sum() >= B // 2
    np.testing.assert_allclose(lp2.detach().cpu().numpy()[ok], lp.detach().numpy()[ok], rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(lp2.detach().cpu().numpy()[~ok], lp.detach().numpy()[~ok], rtol=0.05, atol=0.5)
    np.testing.assert_allclose(hm.grad.cpu().numpy()[ok], m.grad.numpy()[ok], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(hr.grad.cpu().numpy()[ok], r.grad.numpy()[ok], rtol=1e-4, atol=1e-4)
    assert bool(torch.isfinite(hm.grad).all()) and bool(torch.isfinite(hr.grad).all())
    assert hr.grad[0, 0].item() == 0.0 and hr.grad[-1, -1].item() == 0.0      # clamped entries pass no gradient


def test_replay_add_batch_is_one_staged_copy_and_wraps(amd):
    """DeviceReplayBuffer.add_batch (expert demonstrations, DRL.py:469-478) assembles all transitions on the host and moves them
    with one host->device copy; contents, ring wrap and the n > size case must match per-transition adds."""
    from dgvit_amd.replay import DeviceReplayBuffer
    rs = np.random.RandomState(1)
    n = 21
    demo = dict(obs=rs.rand(n, 12, 10).astype(np.float32), pobs=rs.rand(n, 2), act=rs.rand(n, 2), rew=np.arange(n, dtype=np.float32),
                next_obs=rs.rand(n, 12, 10).astype(np.float32), next_pobs=rs.rand(n, 2), done=(np.arange(n) % 2).astype(np.float32))
    a, b = DeviceReplayBuffer(16, obs_shape=(12, 10), seed=0), DeviceReplayBuffer(16, obs_shape=(12, 10), seed=0)
    for j in range(5):                                   # both start 5 transitions into the ring
        for rb in (a, b):
            rb.add(**{k: v[j] for k, v in demo.items()})
    a.add_batch(**{k: v[5:] for k, v in demo.items()})   # 16 more: wraps
    for j in range(5, n):
        b.add(**{k: v[j] for k, v in demo.items()})
    assert a.get_stored_size() == b.get_stored_size() == 16 and a.next_index == b.next_index
    for k in a.store:
        assert torch.equal(a.store[k], b.store[k]), k
    c = DeviceReplayBuffer(8, obs_shape=(12, 10))
    c.add_batch(**demo)                                  # more transitions than slots: the last 8 survive
    assert c.get_stored_size() == 8
    got = c.sample(8, indices=torch.arange(8))
    assert sorted(got["rew"].reshape(-1).tolist()) == list(range(n - 8, n))


@pytest.mark.parametrize("unit_offset", [False, True])
def test_standalone_rmsnorm_with_unit_offset(amd, unit_offset):
    """RMSNorm(dim, unit_offset) of GoalFormer.py:107-122 called on its own (round 1 refused unit_offset=True)."""
    from dgvit_amd.goalformer import RMSNorm
    m = RMSNorm(96, unit_offset=unit_offset).cuda()
    assert float(m.g.detach().abs().max()) == (0.0 if unit_offset else 1.0)
    with torch.no_grad():
        m.g.add_(torch.randn(96, device="cuda") * 0.1)
    x = torch.randn(5, 7, 96, device="cuda", requires_grad=True)
    y = m(x)
    xr, gr = x.detach().double().cpu().requires_grad_(True), m.g.detach().double().cpu().requires_grad_(True)
    ref = torch.nn.functional.normalize(xr, dim=-1) * 96 ** 0.5 * (gr + float(unit_offset))
    w = torch.randn(5, 7, 96)
    (y * w.cuda()).sum().backward(); (ref * w.double()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(m.g.grad.cpu().numpy(), gr.grad.numpy(), rtol=0, atol=1e-4)


# ------------------------------------------------------------------------------------------------ small-batch (per-frame) forward
def _small_path(on):
    """the per-frame experiment lives in the diagnostic library only (frame.hip is not part of the product)"""
    return knobs(force_diag=True, small_batch_path=(1 if on else 0, 0))


@pytest.mark.parametrize("name,cls", [("policy_native_shipped", "policy"), ("policy_native_small", "policy"), ("policy_c2", "policy"),
                                      ("detpolicy_native_shipped", "det")])
def test_small_batch_path_matches_reference_goldens(amd, name, cls):
    """no_grad forward (what SAC.choose_action runs, DRL.py:170-185) on the opt-in two-launches-per-block path: outputs must
    equal the reference's own (fixtures from the unmodified reference classes at 128x160) within 1e-4."""
    fx = load_fixture(name)
    cfg = fixture_cfg(fx)
    batch, seed = int(fx["meta/batch"]), int(fx["meta/seed"])
    kw = dict(image_size=cfg.image, patch_size=cfg.patch)
    if cls == "policy":
        m = _load_state(amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw), O.make_params(O.policy_param_spec(cfg), seed)).eval().to("cuda")
    else:
        m = _load_state(amd.DeterministicGoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, **kw), O.make_params(O.detpolicy_param_spec(cfg), seed)).eval().to("cuda")
    img, pstate, _, _ = O.make_inputs(cfg, batch, seed)
    with _small_path(True), torch.no_grad():
        out = m([img.cuda(), pstate.cuda()])
    if cls == "policy":
        np.testing.assert_allclose(out[0].cpu().numpy(), fx["mean"], rtol=0, atol=OUT_TOL)
        np.testing.assert_allclose(out[1].cpu().numpy(), fx["log_std"], rtol=0, atol=OUT_TOL)
    else:
        np.testing.assert_allclose(out.cpu().numpy(), fx["mean"], rtol=0, atol=OUT_TOL)


@pytest.mark.parametrize("image,patch,dim,depth,heads,dim_head,B", [
    ((128, 160), (16, 20), 64, 4, 4, 64, 1), ((128, 160), (16, 20), 64, 4, 4, 64, 2), ((128, 160), (16, 20), 64, 4, 4, 64, 32),
    ((84, 84), (12, 12), 256, 2, 8, 64, 3),      # DGViT-small width, N = 50
    ((84, 84), (14, 14), 96, 2, 3, 32, 5),       # N = 37, dim_head 32, odd widths
    ((84, 84), (7, 7), 64, 1, 2, 64, 2),         # N = 145 > 128: not eligible, must fall through to the large-batch schedule
    ((24, 24), (24, 24), 32, 2, 1, 64, 4),       # N = 2
])
def test_small_batch_path_equals_large_batch_schedule(amd, image, patch, dim, depth, heads, dim_head, B):
    """Same module, same inputs, eval and train mode (same dropout seed): the per-frame kernels and the GEMM schedule must agree
    to fp32 summation-order rounding, and both with the oracle."""
    cfg = O.GoTConfig(image=image, patch=patch, dim=dim, depth=depth, heads=heads, dim_head=dim_head, mlp_dim=256 if dim < 64 else 2048)
    params = O.make_params(O.got_param_spec(cfg, prefix=""), 17)
    m = amd.GoT(image_size=image, patch_size=patch, num_classes=2, dim=dim, depth=depth, heads=heads, mlp_dim=cfg.mlp_dim, channels=1, dim_head=dim_head)
    m.load_state_dict(params, strict=True)
    m = m.cuda()
    img, _, _, _ = O.make_inputs(cfg, B, 17)
    goal = torch.randn(B, dim, generator=torch.Generator().manual_seed(3))
    ref = O.got_forward(params, img, goal, cfg, prefix="") if hasattr(O, "got_forward") else None
    outs = {}
    for mode in ("eval", "train"):
        m.train(mode == "train")
        for on in (True, False):
            with _small_path(on):
                torch.manual_seed(5)      # same dropout seed draw for both paths
                with torch.no_grad():
                    outs[(mode, on)] = m(img.cuda(), goal.cuda()).cpu()
        np.testing.assert_allclose(outs[(mode, True)].numpy(), outs[(mode, False)].numpy(), rtol=0, atol=5e-5)
    if ref is not None:
        np.testing.assert_allclose(outs[("eval", True)].numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    assert (outs[("train", True)] - outs[("eval", True)]).abs().max().item() > 1e-3      # dropout was live in train mode


@pytest.mark.parametrize("image,patch", [((84, 84), (12, 12)), ((128, 160), (16, 20)), ((224, 224), (16, 16)), ((36, 60), (12, 12)),
                                          ((84, 84), (14, 14)), ((84, 84), (7, 7))])
def test_patch_gather_in_the_gemm_loader(amd, image, patch):
    """Inference forwards take the depth patches straight from the image in the patch-embedding GEMM's tile loader
    (GoalFormer.py:138 folded into the loader; eligible when patch and image widths are multiples of 4); training forwards keep
    the materialised patch matrix.  Same weights, same frames: both must give the same features (and match the oracle)."""
    cfg = O.GoTConfig(image=image, patch=patch, dim=64, depth=1, heads=2)
    params = O.make_params(O.got_param_spec(cfg, prefix=""), 23)
    m = amd.GoT(image_size=image, patch_size=patch, num_classes=2, dim=64, depth=1, heads=2, mlp_dim=cfg.mlp_dim, channels=1)
    m.load_state_dict(params, strict=True)
    m = m.cuda().eval()
    B = 70 if image == (36, 60) else 5                    # 70 frames x 15 patches: several row tiles, a ragged last one
    img, _, _, _ = O.make_inputs(cfg, B, 23)
    goal = torch.randn(B, 64, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        f_gather = m(img.cuda(), goal.cuda()).cpu()                          # save = 0: gather path where eligible
    gg = goal.cuda().requires_grad_(True)
    f_copy = m(img.cuda(), gg).detach().cpu()                                # save = 1: patchify + GEMM
    np.testing.assert_allclose(f_gather.numpy(), f_copy.numpy(), rtol=0, atol=2e-5)
    ref = O.got_forward(params, img, goal, cfg, prefix="")
    np.testing.assert_allclose(f_gather.numpy(), ref.detach().numpy(), rtol=0, atol=OUT_TOL)
    # an image view that is not 16-byte aligned falls back to the copy (and still gives the same result)
    big = torch.zeros(B * image[0] * image[1] + 1, device="cuda")
    off = big[1:].view(B, *image)
    off.copy_(img)
    assert off.data_ptr() % 16 != 0
    with torch.no_grad():
        f_off = m(off, goal.cuda()).cpu()
    np.testing.assert_allclose(f_off.numpy(), f_copy.numpy(), rtol=0, atol=2e-5)


# ---------------------------------------------------------------- single-pass attention backward (32 < N <= 64)
@pytest.mark.gpu
@pytest.mark.parametrize("B,N,H,dh", [(5, 50, 8, 64), (3, 37, 2, 64), (2, 64, 4, 32), (2, 33, 1, 64)])
def test_attention_backward_single_pass_equals_two_phase(B, N, H, dh):
    """The single-pass kernel (every query-tile x key-tile pair computed once, P^T / dS^T crossing LDS) and the two-phase kernel
    it replaces for these shapes are the same function: equal up to fp32 summation order."""
    import dgvit_amd
    from dgvit_amd import functional as F
    dgvit_amd.load_library()
    g = torch.Generator().manual_seed(N * 7 + H)
    qkv = torch.randn(B, N, 3 * H * dh, generator=g).cuda()
    dout = torch.randn(B, N, H * dh, generator=g).cuda()
    out, lse = F.op_attention_fwd(qkv, H, dh)
    with knobs(attention_bwd_single_pass=0):
        two_phase = F.op_attention_bwd(qkv, out, dout, lse, H, dh).clone()
    single = F.op_attention_bwd(qkv, out, dout, lse, H, dh)
    assert torch.isfinite(single).all()
    torch.testing.assert_close(single, two_phase, atol=2e-5, rtol=1e-5)
    again = F.op_attention_bwd(qkv, out, dout, lse, H, dh)
    assert torch.equal(single, again), "fixed summation order: bit-identical from run to run"


# ---------------------------------------------------------------- persistent fp32 GEMM (tile loop in the workgroup)
@pytest.mark.gpu
@pytest.mark.parametrize("layout,epi,M,N,K,tile,wgs", [
    (0, 0, 1000, 384, 256, 64128016, 8),       # NT, bias + residual, ragged M, tiles walk through several rounds per workgroup
    (0, 0, 1000, 384, 256, 64128016, 0),
    (0, 1, 777, 512, 256, 64128016, 16),       # NT, gelu2 (two outputs)
    (0, 3, 520, 256, 512, 64064032, 8),        # NT, relu, 32-deep k-tiles
    (1, 0, 1000, 264, 512, 64064032, 8),       # NN plain, ragged N (multiple of 4, not of the tile)
    (1, 2, 900, 512, 256, 64128016, 24),       # NN, dgelu operand
    (1, 4, 333, 128, 512, 64064032, 8),        # NN, drelu operand, one MFMA column tile per wave
    (0, 0, 300, 256, 256, 64128016, 2),        # two workgroups walk over all ten tiles
    (1, 2, 25600, 2048, 256, 0, 0),            # the C3 shapes themselves, automatic tile and grid
    (0, 1, 25600, 2048, 256, 0, 0),
])
def test_persistent_gemm_equals_per_tile_kernel(layout, epi, M, N, K, tile, wgs):
    """The persistent kernel keeps every tile's k order: results are bit-identical to the per-tile kernel's, for every epilogue,
    ragged edges included."""
    import dgvit_amd
    from dgvit_amd import functional as F
    g = torch.Generator().manual_seed(M + N + K + epi)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) if layout == 0 else torch.randn(K, N, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda() if layout == 0 else None
    res = torch.randn(M, N, generator=g).cuda() if (layout == 0 and epi == 0) else None
    aux = torch.randn(M, N, generator=g).cuda() if epi in (2, 4) else None

    def run(mode):
        # (split tiles sum their k-slices in another order: not the comparison made here)
        with knobs(force_diag=True, gemm_tile=tile, gemm_persistent=(mode, wgs), gemm_split=0) as lib:
            before = lib.dgvit_gemm_persistent_launches()
            out = F.op_gemm(layout, epi, A, B, M, N, K, bias=bias, res=res, aux=aux, want_c2=(epi == 1))
            torch.cuda.synchronize()
            assert lib.dgvit_gemm_persistent_launches() == before + (1 if mode else 0), "the launch did not take the pipelined kernel"
        return out if isinstance(out, tuple) else (out,)

    ref = run(0)
    got = run(2)
    for r, o in zip(ref, got):
        assert torch.isfinite(o).all()
        assert torch.equal(r, o), f"max diff {(r - o).abs().max().item()}"


@pytest.mark.gpu
def test_persistent_gemm_writes_nothing_outside_c():
    """Direct accumulator stores with the descriptor's range check: a C window inside a larger canary buffer stays intact around
    the matrix (rows past M and the columns between N and ldc)."""
    import dgvit_amd
    from dgvit_amd import _lib
    M, N, K, ldc = 203, 132, 256, 160
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).cuda()
    B = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    pad = 4096
    buf = torch.full((pad + M * ldc + pad,), 777.0, device="cuda")
    C = buf[pad:pad + M * ldc].view(M, ldc)
    with knobs(force_diag=True, gemm_tile=64128016, gemm_persistent=(2, 3)) as lib:
        nsc = lib.dgvit_gemm_scratch_floats(0, M, N, K)
        scratch = torch.zeros(max(nsc, 4), device="cuda")
        before = lib.dgvit_gemm_persistent_launches()
        rc = lib.dgvit_gemm(0, 0, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), ldc, M, N, K, bias.data_ptr(), None, 0, None, 0, None, 0,
                            scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "dgvit_gemm")
        assert lib.dgvit_gemm_persistent_launches() == before + 1
    torch.cuda.synchronize()
    ref = (A.double() @ B.double().t() + bias.double()).float()
    torch.testing.assert_close(C[:, :N], ref, atol=2e-4, rtol=1e-5)
    assert (C[:, N:] == 777.0).all(), "columns between N and ldc were written"
    assert (buf[:pad] == 777.0).all() and (buf[pad + M * ldc:] == 777.0).all(), "wrote outside C"


@pytest.mark.gpu
@pytest.mark.parametrize("layout,epi,M,N,K,ldc", [(0, 0, 203, 132, 256, 160), (0, 1, 77, 260, 64, 264), (1, 2, 130, 136, 96, 200), (1, 0, 65, 68, 512, 68)])
def test_direct_gemm_epilogue_stays_inside_c(layout, epi, M, N, K, ldc):
    """The direct (register) epilogue of the per-tile fp32 GEMM - the default for the NT / NN forms - writes through a range-checked
    buffer descriptor: a C window with ldc >= N inside a canary buffer keeps its padding columns, the rows past M and everything
    around it; the values equal an fp64 reference and the LDS-image epilogue (dgvit_set_gemm_diagnostics(8))."""
    import math
    import dgvit_amd
    from dgvit_amd import _lib
    lib = dgvit_amd.load_library()
    g = torch.Generator().manual_seed(M * 3 + N)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) if layout == 0 else torch.randn(K, N, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda() if layout == 0 else None
    res = torch.randn(M, N, generator=g).cuda() if (layout == 0 and epi == 0) else None
    aux = torch.randn(M, N, generator=g).cuda() if epi == 2 else None
    pad = 2048
    nsc = lib.dgvit_gemm_scratch_floats(layout, M, N, K)
    scratch = torch.zeros(max(nsc, 4), device="cuda")
    outs = []
    for diag in (0, 8):
        buf = torch.full((pad + M * ldc + pad,), 555.0, device="cuda")
        buf2 = torch.full((pad + M * ldc + pad,), 555.0, device="cuda")
        C, C2 = buf[pad:pad + M * ldc].view(M, ldc), buf2[pad:pad + M * ldc].view(M, ldc)
        # diag 0: the direct epilogue, diag 8: the LDS-image epilogue forced; both in the diagnostic library with the in-launch
        # split off (a split tile sums its k-slices in another order, which is not the comparison made here)
        with knobs(force_diag=True, gemm_diagnostics=diag, gemm_split=0) as klib:
            rc = klib.dgvit_gemm(layout, epi, A.data_ptr(), K, B.data_ptr(), K if layout == 0 else N, C.data_ptr(), ldc, M, N, K,
                                 bias.data_ptr() if bias is not None else None, res.data_ptr() if res is not None else None, N,
                                 C2.data_ptr() if epi == 1 else None, ldc, aux.data_ptr() if aux is not None else None, N,
                                 scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "dgvit_gemm")
        torch.cuda.synchronize()
        for b_, c_ in ((buf, C),) + (((buf2, C2),) if epi == 1 else ()):
            assert (c_[:, N:] == 555.0).all(), "columns between N and ldc were written"
            assert (b_[:pad] == 555.0).all() and (b_[pad + M * ldc:] == 555.0).all(), "wrote outside C"
        outs.append((C[:, :N].clone(), C2[:, :N].clone()))
    acc = A.double() @ (B.double().t() if layout == 0 else B.double())
    if epi == 0:
        ref = acc + (bias.double() if bias is not None else 0) + (res.double() if res is not None else 0)
    elif epi == 1:
        ref = acc + bias.double()
    else:
        x = aux.double()
        ref = acc * (0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-x * x / 2) / math.sqrt(2 * math.pi))
    torch.testing.assert_close(outs[0][0], ref.float(), atol=3e-4, rtol=1e-5)
    assert torch.equal(outs[0][0], outs[1][0]), "direct and LDS-image epilogues differ"
    if epi == 1:
        torch.testing.assert_close(outs[0][1], torch.nn.functional.gelu(ref).float(), atol=3e-4, rtol=1e-5)
        assert torch.equal(outs[0][1], outs[1][1])


# ---------------------------------------------------------------- LayerNorm fused into the producing GEMM's epilogue (dim 64)
@pytest.mark.gpu
@pytest.mark.parametrize("B,split", [(1, 1), (7, 1), (32, 1), (700, 1), (32, 0), (700, 0)])
def test_layernorm_fused_into_gemm_epilogue_is_bit_identical(B, split):
    """dim == 64: to_out / fc2 compute the following LayerNorm in their epilogue (split-K last arriver for few rows, LDS-image
    epilogue for many).  Outputs and every gradient equal the separate-launch schedule bit for bit, in eval and in train mode."""
    import dgvit_amd
    lib = dgvit_amd.load_library()
    torch.manual_seed(11)
    model = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).cuda()
    g = torch.Generator().manual_seed(B)
    img = torch.rand(B, 128, 160, generator=g).cuda()
    ps = torch.rand(B, 2, generator=g).cuda()

    def run(fused, train):
        with knobs(ln_fusion=fused, gemm_split=split):     # (1, 1) = the product library itself
            model.train(train)
            torch.manual_seed(5)           # same embedding-dropout mask in both schedules
            for p_ in model.parameters():
                p_.grad = None
            mean, log_std = model([img, ps])
            if train:
                (mean.square().mean() + log_std.square().mean()).backward()
            torch.cuda.synchronize()
            grads = [p_.grad.clone() for p_ in model.parameters() if p_.grad is not None]
            return mean.detach().clone(), log_std.detach().clone(), grads

    for train in (False, True):
        a, b = run(1, train), run(0, train)
        assert torch.isfinite(a[0]).all()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "outputs differ between fused and separate LayerNorm"
        assert len(a[2]) == len(b[2])
        for x, y in zip(a[2], b[2]):
            assert torch.equal(x, y), "a gradient differs between fused and separate LayerNorm"


# ---------------------------------------------------------------- implicit-GEMM convolution (conv2 / conv3 forward)
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(1, 128, 160), (5, 128, 160), (32, 128, 160), (3, 84, 84)])
def test_conv_forward_gather_equals_im2col(B, H, W):
    """conv2 / conv3 read their 5x5xC windows through the GEMM's A-tile loader (no column matrix): same k order, so the CNN critic's
    outputs and gradients equal the im2col schedule bit for bit."""
    import dgvit_amd
    lib = dgvit_amd.load_library()
    torch.manual_seed(4)
    net = dgvit_amd.QNetwork(2, 2).cuda()
    g = torch.Generator().manual_seed(B + H)
    img = torch.rand(B, H, W, generator=g).cuda()
    ps = torch.rand(B, 2, generator=g).cuda()
    act = (torch.rand(B, 2, generator=g) * 2 - 1).cuda()

    def run(on):
        # (k-slices off: since round 4 the gathered form is split inside the launch at small batches, another summation order;
        #  tests/test_gpu_round4.py::test_implicit_gemm_convolutions_take_k_slices_at_small_batches covers that against this one)
        with knobs(conv_gather=on, gemm_split=0):
            for p_ in net.parameters():
                p_.grad = None
            q1, q2 = net([img, ps, act])
            (q1.square().mean() + q2.mean()).backward()
            torch.cuda.synchronize()
            return q1.detach().clone(), q2.detach().clone(), [p_.grad.clone() for p_ in net.parameters() if p_.grad is not None]

    a, b = run(1), run(0)
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)
