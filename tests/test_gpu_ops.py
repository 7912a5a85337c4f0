"""GPU: operator-level parity of every HIP kernel against fp64 CPU arithmetic on the same inputs.
Tolerances are written per test; values are O(1) so abs tolerances are meaningful."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import O, knobs  # noqa: E402


@pytest.fixture(scope="module")
def F():
    import dgvit_amd
    dgvit_amd.load_library()
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return dgvit_amd.functional


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def dev(t):
    return t.float().cuda()


def close(got, ref, atol, rtol=0.0, msg=""):
    np.testing.assert_allclose(got.detach().double().cpu().numpy(), ref.double().numpy(), rtol=rtol, atol=atol, err_msg=msg)


# ---------------------------------------------------------------- GEMM: layouts x tiles x ragged shapes
GEMM_SHAPES = [(128, 128, 32), (256, 384, 64), (200, 136, 100), (50, 64, 256), (1, 2, 128), (37, 130, 258), (300, 7, 49),
               (65, 768, 64), (512, 256, 2048)]


@pytest.mark.parametrize("tile", [0, 64, 128])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt_bias_residual(F, M, N, K, tile):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    with knobs(gemm_tile=tile):
        y = F.op_gemm(0, 0, dev(x), dev(w), M, N, K, bias=dev(b), res=dev(r))
    close(y, x @ w.T + b + r, atol=2e-4 * math.sqrt(K), msg=f"NT {M}x{N}x{K}")


@pytest.mark.parametrize("tile", [0, 64])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nn(F, M, N, K, tile):
    a, b = rnd(M, K, seed=5), rnd(K, N, seed=6)
    with knobs(gemm_tile=tile):
        y = F.op_gemm(1, 0, dev(a), dev(b), M, N, K)
    close(y, a @ b, atol=2e-4 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(128, 128, 4096), (1536, 256, 3000), (64, 320, 128), (2, 128, 512), (130, 258, 777), (256, 2048, 25600)])
def test_gemm_tn_splitk(F, M, N, K):
    a, b = rnd(K, M, seed=7), rnd(K, N, seed=8)
    y = F.op_gemm(2, 0, dev(a), dev(b), M, N, K)
    close(y, a.T @ b, atol=3e-4 * math.sqrt(K))


def test_gemm_epilogues(F):
    M, N, K = 192, 256, 96
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    h = x @ w.T + b
    c, c2 = F.op_gemm(0, 1, dev(x), dev(w), M, N, K, bias=dev(b), want_c2=True)
    close(c, h, atol=1e-4)
    close(c2, 0.5 * h * (1 + torch.erf(h / math.sqrt(2))), atol=1e-4)
    y = F.op_gemm(0, 3, dev(x), dev(w), M, N, K, bias=dev(b))
    close(y, torch.relu(h), atol=1e-4)
    # dgelu: C = (A B) * gelu'(aux)
    a, bm, aux = rnd(M, K, seed=4), rnd(K, N, seed=5, scale=0.2), rnd(M, N, seed=6)
    gp = 0.5 * (1 + torch.erf(aux / math.sqrt(2))) + aux * torch.exp(-0.5 * aux * aux) / math.sqrt(2 * math.pi)
    y = F.op_gemm(1, 2, dev(a), dev(bm), M, N, K, aux=dev(aux))
    close(y, (a @ bm) * gp, atol=1e-4)
    y = F.op_gemm(1, 4, dev(a), dev(bm), M, N, K, aux=dev(aux))
    close(y, (a @ bm) * (aux > 0), atol=1e-4)


def test_gemm_mfma_layout_identity(F):
    """A = I with an asymmetric B catches a transposed accumulator map (cdna guide section 3)."""
    n = 128
    b = torch.arange(n * n, dtype=torch.float64).reshape(n, n) / 1000.0
    y = F.op_gemm(1, 0, dev(torch.eye(n, dtype=torch.float64)), dev(b), n, n, n)
    close(y, b, atol=1e-5)
    y = F.op_gemm(0, 0, dev(torch.eye(n, dtype=torch.float64)), dev(b), n, n, n)
    close(y, b.T, atol=1e-5)


# ---------------------------------------------------------------- norms
@pytest.mark.parametrize("T,D", [(7, 32), (1000, 256), (50, 64), (33, 768), (5, 1024)])
def test_layernorm_fwd_bwd(F, T, D):
    x, g, b, dy, dres = rnd(T, D, seed=1, scale=2.0) + 0.5, 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3), rnd(T, D, seed=4), rnd(T, D, seed=5)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    ref.backward(dy)
    y, mean, rstd = F.op_layernorm_fwd(dev(x), dev(g), dev(b))
    close(y, ref.detach(), atol=2e-5)
    dx, dg, db = F.op_layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dres=dev(dres))
    close(dx, xr.grad + dres, atol=5e-5)
    close(dg, gr.grad, atol=1e-4 * math.sqrt(T))
    close(db, br.grad, atol=1e-4 * math.sqrt(T))


@pytest.mark.parametrize("B,D", [(3, 32), (512, 256), (70, 768)])
def test_rmsnorm_fwd_bwd(F, B, D):
    x, g, dy = rnd(B, D, seed=1), 1 + 0.1 * rnd(D, seed=2), rnd(B, D, seed=3)
    xr, gr = x.clone().requires_grad_(True), g.clone().requires_grad_(True)
    ref = O.rms_norm(xr, gr)
    ref.backward(dy)
    close(F.op_rmsnorm_fwd(dev(x), dev(g)), ref.detach(), atol=1e-5)
    dx, dg = F.op_rmsnorm_bwd(dev(dy), dev(x), dev(g))
    close(dx, xr.grad, atol=2e-5)
    close(dg, gr.grad, atol=1e-4 * math.sqrt(B))


# ---------------------------------------------------------------- attention
def _attn_ref(qkv, H, dh):
    B, N, _ = qkv.shape
    I = H * dh
    q, k, v = (qkv[..., j * I:(j + 1) * I].reshape(B, N, H, dh).permute(0, 2, 1, 3) for j in range(3))
    p = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, N, I)


@pytest.mark.parametrize("B,N,H,dh", [(3, 50, 8, 64), (2, 65, 4, 64), (2, 7, 2, 32), (1, 32, 1, 64), (2, 37, 2, 64), (1, 145, 2, 64),
                                      (1, 197, 3, 64), (2, 64, 2, 32), (1, 1, 2, 64), (1, 224, 1, 64), (2, 100, 3, 32), (1, 96, 2, 64),
                                      (2, 33, 2, 64), (1, 64, 3, 64), (2, 50, 2, 32), (1, 63, 1, 32),
                                      (2, 225, 2, 64), (1, 257, 3, 64), (2, 288, 2, 64), (1, 257, 2, 32), (1, 288, 1, 32)])
def test_attention_fwd_bwd(F, B, N, H, dh):
    qkv = rnd(B, N, 3 * H * dh, seed=N)
    dout = rnd(B, N, H * dh, seed=N + 1)
    qr = qkv.clone().requires_grad_(True)
    ref = _attn_ref(qr, H, dh)
    ref.backward(dout)
    out, lse = F.op_attention_fwd(dev(qkv), H, dh)
    close(out, ref.detach(), atol=2e-5)
    I = H * dh
    q, k = (qkv[..., j * I:(j + 1) * I].reshape(B, N, H, dh).permute(0, 2, 1, 3) for j in range(2))
    lse_ref = torch.logsumexp((q @ k.transpose(-1, -2)) * dh ** -0.5, -1) / math.log(2.0)     # base-2 units
    close(lse, lse_ref, atol=2e-5)
    dqkv = F.op_attention_bwd(dev(qkv), out, dev(dout), lse, H, dh)
    close(dqkv, qr.grad, atol=1e-4)


def test_attention_peaked_softmax(F):
    """Large logits: the max-subtraction path must hold (one key dominates each row)."""
    B, N, H, dh = 1, 50, 2, 64
    qkv = rnd(B, N, 3 * H * dh, seed=3) * 6.0
    ref = _attn_ref(qkv, H, dh)
    close(F.op_attention_fwd(dev(qkv), H, dh)[0], ref, atol=5e-4)


# ---------------------------------------------------------------- token assembly pieces
@pytest.mark.parametrize("shape,patch", [((3, 84, 84), (12, 12)), ((2, 128, 160), (16, 20)), ((2, 84, 84), (7, 7)), ((1, 16, 24), (8, 8))])
def test_patchify(F, shape, patch):
    img = rnd(*shape, seed=9)
    cfg = O.GoTConfig(image=shape[1:], patch=patch)
    close(F.op_patchify(dev(img), patch), O.patchify(img.float(), cfg), atol=0.0)  # pure data movement: bit exact


def test_dropout_statistics_and_replay(F):
    n = 1 << 20
    x = torch.ones(n, device="cuda")
    F.op_dropout_(x, 1234, 0.9)
    kept = (x != 0).float().mean().item()
    assert abs(kept - 0.9) < 3e-3, kept
    vals = torch.unique(x).cpu().numpy()
    np.testing.assert_allclose(sorted(vals), [0.0, 1.0 / 0.9], rtol=1e-6)
    y = torch.ones(n, device="cuda")
    F.op_dropout_(y, 1234, 0.9)
    assert torch.equal(x, y), "same seed must replay the same mask (backward relies on it)"
    z = torch.ones(n, device="cuda")
    F.op_dropout_(z, 1235, 0.9)
    assert not torch.equal(x, z)
    # neighbouring elements uncorrelated
    m = (x != 0).float()
    c = ((m[:-1] - 0.9) * (m[1:] - 0.9)).mean().item() / (0.9 * 0.1)
    assert abs(c) < 0.01, c


# ---------------------------------------------------------------- head Linear autograd node
@pytest.mark.parametrize("M,N,K,relu", [(512, 128, 256, True), (4, 2, 128, False), (33, 128, 258, True), (1, 64, 2, False), (512, 32, 128, True)])
def test_linear_autograd(F, M, N, K, relu):
    x, w, b, dy = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.3), rnd(N, seed=3), rnd(M, N, seed=4)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = xr @ wr.T + br
    if relu:
        ref = torch.relu(ref)
    ref.backward(dy)
    xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
    y = F.linear(xg, wg, bg, relu=relu)
    y.backward(dev(dy))
    close(y, ref.detach(), atol=1e-4)
    close(xg.grad, xr.grad, atol=2e-4)
    close(wg.grad, wr.grad, atol=2e-4 * math.sqrt(M))
    close(bg.grad, br.grad, atol=2e-4 * math.sqrt(M))


def test_live_profile_sampling(F):
    """dgvit_profile_*: every launch timed (stride 1) or every 3rd; totals count all launches either way."""
    import ctypes
    import dgvit_amd
    from dgvit_amd import _lib
    lib = dgvit_amd.load_library()
    A, B = torch.randn(256, 128, device="cuda"), torch.randn(192, 128, device="cuda")
    kinds = _lib.PROFILE_KINDS
    for stride, timed in ((1, 10), (3, 4)):
        assert lib.dgvit_profile_sampling(stride) == 0
        assert lib.dgvit_profile_start(64) == 0
        for _ in range(10):
            F.op_gemm(0, 0, A, B, 256, 192, 128)
        ms, work, cnt = (ctypes.c_double * kinds)(), (ctypes.c_double * kinds)(), (ctypes.c_longlong * kinds)()
        assert lib.dgvit_profile_stop(ms, work, cnt) == 0
        wall, call = (ctypes.c_double * kinds)(), (ctypes.c_longlong * kinds)()
        assert lib.dgvit_profile_totals(wall, call) == 0
        flops = 2.0 * 256 * 192 * 128
        assert cnt[0] == timed and call[0] == 10
        assert work[0] == timed * flops and wall[0] == 10 * flops
        assert 0.0 < ms[0] < 50.0
    lib.dgvit_profile_sampling(1)
    assert lib.dgvit_profile_sampling(0) < 0
