"""the depth-frame preprocessing in front of the encoder, on the device.

The reference does this on the host with OpenCV for every camera message and every environment step
(``env_lab.py``: ``listener_callback`` :420-434, ``add_nose`` :78-89, ``blurring`` :69-76, ``get_center_band`` :33-39,
``cv2.resize(..., (160, 128))`` and ``/ 255`` :295-299 / :348-349).  Here the frames stay in HBM as fp32 ``(B, H, W)``
(or ``(H, W)``) tensors and every stage is a HIP kernel of ``libdgvit_hip.so``; function names follow the reference.
Parity against OpenCV itself is unpinned (cv2 is not installed in the build image; see oracle/dgvit_oracle.py f4_*).
"""
import ctypes

import torch

from . import _lib
from .functional import _dev, _ptr, _stream


def _frames(img):
    img = _dev(img, "image")
    if img.dim() == 2:
        return img.unsqueeze(0), True
    if img.dim() != 3:
        raise _lib.DgvitError(f"image must be (H, W) or (B, H, W), got {tuple(img.shape)}")
    return img, False


def get_center_band(image):
    """(y1, y2) of the horizontal centre band of height H // 5 (env_lab.py:33-39)."""
    h = image.shape[-2]
    band = h // 5
    y1 = h // 2 - band // 2
    return y1, y1 + band


def depth_to_uint8(depth):
    """float depth -> 0..255 integers (``cv2.normalize(..., 0, 255, NORM_MINMAX).astype(np.uint8)``, env_lab.py:424-426), kept as fp32."""
    lib = _lib.load()
    x, squeeze = _frames(depth)
    B, H, W = x.shape
    out = torch.empty_like(x)
    scratch = torch.empty(128 * B, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.dgvit_depth_normalize_u8(_ptr(x), _ptr(out), _ptr(scratch), scratch.numel(), B, H, W, _stream()), "dgvit_depth_normalize_u8")
    return out[0] if squeeze else out


def _blur(x, ksize, y0, y1):
    lib = _lib.load()
    B, H, W = x.shape
    out = x.clone()
    tmp = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.dgvit_gaussian_blur(_ptr(x), _ptr(out), _ptr(tmp), B, H, W, ksize, y0, y1, _stream()), "dgvit_gaussian_blur")
    return out


def add_nose(image, noise_level=0.02, noise=None, seed=None):
    """``clip(image + N(0, noise_level), 0, 255)`` then a 5x5 Gaussian blur (env_lab.py:78-89).  ``noise``: the draw to add (same
    shape; parity tests); otherwise it is drawn on the device from ``seed`` (default: torch's CPU generator)."""
    lib = _lib.load()
    x, squeeze = _frames(image)
    B, H, W = x.shape
    if x.numel() % 4:
        raise _lib.DgvitError("add_nose: the number of pixels must be a multiple of 4")
    if noise is not None:
        noise = _dev(noise, "noise").reshape(x.shape)
    elif seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.dgvit_noise_clip(_ptr(x), _ptr(noise), _ptr(y), x.numel(), float(noise_level), int(seed or 0), _stream()), "dgvit_noise_clip")
    out = _blur(y, 5, 0, H)
    return out[0] if squeeze else out


def blurring(image):
    """11x11 Gaussian blur of the horizontal centre band (env_lab.py:69-76)."""
    x, squeeze = _frames(image)
    y1, y2 = get_center_band(x)
    out = _blur(x, 11, y1, y2)
    return out[0] if squeeze else out


def resize_state(image, size=(128, 160)):
    """``cv2.resize(image, (160, 128)) / 255`` (env_lab.py:295,299 and :348-349); ``size`` = (height, width)."""
    lib = _lib.load()
    x, squeeze = _frames(image)
    B, H, W = x.shape
    out = torch.empty(B, size[0], size[1], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.dgvit_resize_bilinear(_ptr(x), _ptr(out), B, H, W, size[0], size[1], 1.0 / 255.0, _stream()), "dgvit_resize_bilinear")
    return out[0] if squeeze else out


def depth_to_state(depth, noise_level=50, noise=None, seed=None, size=(128, 160)):
    """The whole chain of the reference in one call: listener_callback (normalise -> uint8 -> add_nose(50) -> blurring) followed by
    the resize + /255 of ``step`` / ``reset``: depth ``(B, H, W)`` -> encoder frames ``(B, size[0], size[1])`` in [0, 1]."""
    lib = _lib.load()
    x, squeeze = _frames(depth)
    B, H, W = x.shape
    if noise is not None:
        noise = _dev(noise, "noise").reshape(x.shape)
    elif seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    n = lib.dgvit_depth_preprocess_scratch_floats(B, H, W)
    scratch = torch.empty(n, dtype=torch.float32, device=x.device)
    state = torch.empty(B, size[0], size[1], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.dgvit_depth_to_state(_ptr(x), _ptr(noise), float(noise_level), int(seed or 0), _ptr(state), _ptr(scratch), n, B, H, W,
                                      size[0], size[1], _stream())
    _lib.check(rc, "dgvit_depth_to_state")
    return state[0] if squeeze else state
