"""GPU: SURVEY 8(f4) depth-frame preprocessing kernels against the oracle's restatement of OpenCV's formulas.
PARITY UNPINNED against OpenCV itself (cv2 is not installed; the reference holds no fixtures for these steps): what is checked
is HIP == oracle on the same inputs (fp32 pixel arithmetic: 1e-3 on 0..255 values, i.e. 4e-6 of the range), invariants that
do not depend on the restatement (constant images, integer outputs, value ranges, identity resize), the resize against torch's
independent bilinear interpolation, and the statistics of the device noise generator."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import O  # noqa: E402


@pytest.fixture(scope="module")
def P():
    import dgvit_amd
    dgvit_amd.load_library()
    assert torch.cuda.is_available()
    from dgvit_amd import preprocess
    return preprocess


def depth_frames(b, h, w, seed):
    rs = np.random.RandomState(seed)
    d = rs.uniform(0.3, 9.5, (b, h, w)).astype(np.float32)
    d[:, : h // 3] += np.linspace(0, 3, w, dtype=np.float32)      # smooth structure on top of the noise
    return d


@pytest.mark.parametrize("b,h,w", [(3, 440, 640), (1, 128, 160), (2, 37, 52)])
def test_stages_match_the_oracle(P, b, h, w):
    d = depth_frames(b, h, w, 1)
    noise = (np.random.RandomState(2).standard_normal(d.shape) * 50).astype(np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()
    u = P.depth_to_uint8(dev(d)).cpu().numpy()
    ref_u = O.f4_depth_to_uint8(d)
    assert (u == np.trunc(u)).all() and u.min() == 0 and u.max() in (254, 255)     # max * a + b in fp32 can land just below 255
    assert np.abs(u - ref_u).max() <= 1.0 and (u != ref_u).mean() < 1e-3      # a value on an integer boundary may truncate either way
    a = P.add_nose(dev(ref_u), noise_level=50, noise=dev(noise)).cpu().numpy()
    np.testing.assert_allclose(a, O.f4_add_nose(ref_u, noise), atol=1e-3)
    ref_a = O.f4_add_nose(ref_u, noise)
    bl = P.blurring(dev(ref_a)).cpu().numpy()
    np.testing.assert_allclose(bl, O.f4_blurring(ref_a), atol=1e-3)
    y1, y2 = P.get_center_band(torch.empty(h, w))
    assert (bl[:, :y1] == ref_a[:, :y1]).all() and (bl[:, y2:] == ref_a[:, y2:]).all()       # outside the band nothing changes
    ref_b = O.f4_blurring(ref_a)
    s = P.resize_state(dev(ref_b)).cpu().numpy()
    np.testing.assert_allclose(s, O.f4_resize_to_state(ref_b), atol=1e-5)
    t = torch.nn.functional.interpolate(torch.from_numpy(ref_b)[:, None], size=(128, 160), mode="bilinear", align_corners=False)[:, 0] / 255
    np.testing.assert_allclose(s, t.numpy(), atol=1e-5)


def test_whole_chain_and_single_frame(P):
    d = depth_frames(2, 440, 640, 3)
    noise = (np.random.RandomState(4).standard_normal(d.shape) * 50).astype(np.float32)
    ref = O.f4_pipeline(d, noise)
    got = P.depth_to_state(torch.from_numpy(d).cuda(), noise_level=50, noise=torch.from_numpy(noise).cuda()).cpu().numpy()
    assert got.shape == (2, 128, 160) and got.min() >= 0 and got.max() <= 1
    # (a pixel whose normalised value sits on an integer boundary may truncate differently: one grey level, smoothed by two blurs)
    assert np.abs(got - ref).max() < 2e-3 and np.abs(got - ref).mean() < 1e-5
    one = P.depth_to_state(torch.from_numpy(d[0]).cuda(), noise_level=50, noise=torch.from_numpy(noise[0]).cuda()).cpu().numpy()
    assert one.shape == (128, 160) and np.array_equal(one, got[0])


def test_invariants(P):
    c = torch.full((2, 60, 80), 137.0).cuda()
    assert torch.equal(P.add_nose(c, 0.0, noise=torch.zeros_like(c)), c)            # blur of a constant image
    assert (P.blurring(c) - c).abs().max() < 1e-4
    assert (P.resize_state(c) * 255 - 137).abs().max() < 1e-4
    x = torch.rand(1, 128, 160).cuda() * 255
    torch.testing.assert_close(P.resize_state(x) * 255, x, atol=1e-4, rtol=0)       # identity size
    flat = torch.full((1, 16, 16), 3.25).cuda()
    assert float(P.depth_to_uint8(flat).abs().max()) == 0.0                         # max == min -> scale 0 (OpenCV's rule)


def test_device_noise_statistics(P):
    z = torch.full((4, 440, 640), 128.0).cuda()
    lib = __import__("dgvit_amd").load_library()
    import ctypes
    out = torch.empty_like(z)
    lib.dgvit_noise_clip(ctypes.c_void_p(z.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), z.numel(), 20.0, 1234, None)
    torch.cuda.synchronize()
    n = (out - 128.0).double().cpu()
    assert abs(float(n.mean())) < 0.1 and abs(float(n.std()) - 20.0) < 0.1          # 1.1 M samples of N(0, 20), clipping negligible
    assert abs(float((n.abs() < 20).double().mean()) - 0.6827) < 5e-3
    out2 = torch.empty_like(z)
    lib.dgvit_noise_clip(ctypes.c_void_p(z.data_ptr()), None, ctypes.c_void_p(out2.data_ptr()), z.numel(), 20.0, 1234, None)
    lib.dgvit_noise_clip(ctypes.c_void_p(z.data_ptr()), None, ctypes.c_void_p(z.data_ptr()), z.numel(), 20.0, 1235, None)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and not torch.equal(out, z)                       # reproducible per seed, different across seeds
