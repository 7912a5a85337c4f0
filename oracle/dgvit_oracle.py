"""CPU oracle for the DGViT hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch CPU restatement (plain torch tensor arithmetic, no
nn.Module, no einops) of the reference's depth/goal Vision-Transformer encoder
and its SAC heads.  It exists to *check* the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * the product package (``dgvit_amd``) never imports it and has no CPU
    fallback -- it raises if the HIP library is missing.

Parity pin: the reference ships no tests for this path, so the
oracle is pinned against outputs of the reference itself, generated in the
build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference/src/vis_nav/vis_nav/{GoalFormer,got_sac_network}.py``) and
committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` holds the
oracle to those vectors (<= 2e-6 abs in fp32).

Every function cites the reference lines it restates (paths relative to
``/root/reference/src/vis_nav/vis_nav/``).

Parameters are passed as a flat ``dict[str, Tensor]`` keyed exactly like the
reference ``state_dict``, so the same dict can be loaded into
the reference modules, the product modules and this oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

LOG_SIG_MAX = 2.0    # got_sac_network.py:18
LOG_SIG_MIN = -20.0  # got_sac_network.py:19
EPSILON = 1e-6       # got_sac_network.py:20


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class GoTConfig:
    """Shape of one encoder (GoalFormer.py:124 ctor arguments).

    The reference hard-wires patch 16x20 / 320 pixels (GoalFormer.py:137-139);
    the oracle honours ``patch`` so the 84x84 and 224x224 BASELINE configs can
    be expressed ("Oracle for shapes the reference classes
    cannot take").
    """
    image: Tuple[int, int] = (128, 160)
    patch: Tuple[int, int] = (16, 20)
    dim: int = 64
    depth: int = 4
    heads: int = 4
    dim_head: int = 64        # GoalFormer.py:124 default
    mlp_dim: int = 2048       # got_sac_network.py:86,183,400
    num_classes: int = 2      # unused mlp_head width (got_sac_network.py:82)

    @property
    def grid(self) -> Tuple[int, int]:
        return self.image[0] // self.patch[0], self.image[1] // self.patch[1]

    @property
    def num_patches(self) -> int:
        return self.grid[0] * self.grid[1]

    @property
    def tokens(self) -> int:
        return self.num_patches + 1

    @property
    def patch_dim(self) -> int:
        return self.patch[0] * self.patch[1]

    @property
    def inner(self) -> int:
        return self.heads * self.dim_head

    @property
    def project_out(self) -> bool:
        """GoalFormer.py:56: ``project_out = not (heads == 1 and dim_head == dim)``; False => ``to_out = nn.Identity()`` (:66-69)."""
        return not (self.heads == 1 and self.dim_head == self.dim)

    def fwd_flops_per_frame(self) -> float:
        """GEMM-only forward FLOPs per frame (SURVEY.md section 8(d) formula)."""
        P, pd, D, L = self.num_patches, self.patch_dim, self.dim, self.depth
        N, I, M = self.tokens, self.inner, self.mlp_dim
        return 2.0 * P * pd * D + L * (2.0 * N * D * 3 * I + 4.0 * N * N * I
                                       + 2.0 * N * I * D + 4.0 * N * D * M)


# --------------------------------------------------------------------------
# parameter inventory (the checkpoint ABI)
# --------------------------------------------------------------------------
def got_param_spec(cfg: GoTConfig, prefix: str = "trans.") -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, init-kind) in the reference's registration order."""
    D, I, M, N = cfg.dim, cfg.inner, cfg.mlp_dim, cfg.tokens
    s: List[Tuple[str, Tuple[int, ...], str]] = [
        (prefix + "pos_embedding", (1, N, D), "randn"),               # GoalFormer.py:142
        (prefix + "cls_token", (1, 1, D), "randn"),                   # GoalFormer.py:143 (unused)
        (prefix + "layer_norm.g", (D,), "gain"),                      # GoalFormer.py:117-118
        (prefix + "to_patch_embedding.1.weight", (D, cfg.patch_dim), "xavier"),  # :139
        (prefix + "to_patch_embedding.1.bias", (D,), "bias"),
    ]
    for i in range(cfg.depth):
        lp = f"{prefix}transformer.layers.{i}."
        s += [
            (lp + "0.norm.weight", (D,), "gain"),                     # GoalFormer.py:34
            (lp + "0.norm.bias", (D,), "lnbias"),
            (lp + "0.fn.to_qkv.weight", (3 * I, D), "xavier"),        # GoalFormer.py:64
        ]
        if cfg.project_out:                                           # GoalFormer.py:56,66-69: nn.Identity() has no parameters
            s += [(lp + "0.fn.to_out.0.weight", (D, I), "xavier"),
                  (lp + "0.fn.to_out.0.bias", (D,), "bias")]
        s += [
            (lp + "1.norm.weight", (D,), "gain"),
            (lp + "1.norm.bias", (D,), "lnbias"),
            (lp + "1.fn.net.0.weight", (M, D), "xavier"),             # GoalFormer.py:43
            (lp + "1.fn.net.0.bias", (M,), "bias"),
            (lp + "1.fn.net.3.weight", (D, M), "xavier"),             # GoalFormer.py:46
            (lp + "1.fn.net.3.bias", (D,), "bias"),
        ]
    s += [
        (prefix + "mlp_head.0.weight", (D,), "gain"),                 # GoalFormer.py:151-154 (unused)
        (prefix + "mlp_head.0.bias", (D,), "lnbias"),
        (prefix + "mlp_head.1.weight", (cfg.num_classes, D), "xavier"),
        (prefix + "mlp_head.1.bias", (cfg.num_classes,), "bias"),
    ]
    return s


def _lin(name: str, out_f: int, in_f: int):
    return [(name + ".weight", (out_f, in_f), "xavier"), (name + ".bias", (out_f,), "bias")]


def policy_param_spec(cfg: GoTConfig, nb_actions: int = 2, nb_pstate: int = 2):
    """GoTPolicy keys (got_sac_network.py:173-192)."""
    s = got_param_spec(cfg)
    s += _lin("fc_embed", cfg.dim, nb_pstate)
    s += _lin("fc1", 128, cfg.dim) + _lin("fc2", 128, 128)
    s += _lin("mean_linear", nb_actions, 128) + _lin("log_std_linear", nb_actions, 128)
    return s


def qnet_param_spec(cfg: GoTConfig, nb_actions: int = 2, nb_pstate: int = 2):
    """GoTQNetwork keys (got_sac_network.py:76-103), dead conv1-3 included."""
    s = got_param_spec(cfg)
    s += [("conv1.weight", (16, 4, 5, 5), "conv"), ("conv1.bias", (16,), "bias"),
          ("conv2.weight", (64, 16, 5, 5), "conv"), ("conv2.bias", (64,), "bias"),
          ("conv3.weight", (256, 64, 5, 5), "conv"), ("conv3.bias", (256,), "bias")]
    s += _lin("fc1", 128, cfg.dim + nb_actions) + _lin("fc2", 32, 128) + _lin("fc3", nb_actions, 32)
    s += _lin("fc_embed", cfg.dim, nb_pstate)
    s += _lin("fc11", 128, cfg.dim + nb_actions) + _lin("fc21", 32, 128) + _lin("fc31", nb_actions, 32)
    return s


def detpolicy_param_spec(cfg: GoTConfig, nb_actions: int = 2, nb_pstate: int = 2):
    """DeterministicGoTPolicy keys (got_sac_network.py:390-413)."""
    s = got_param_spec(cfg)
    s += _lin("fc_embed", cfg.dim, nb_pstate)
    s += _lin("fc1", 128, cfg.dim) + _lin("fc2", 32, 128)
    s += _lin("mean_linear", nb_actions, 32) + _lin("log_std_linear", nb_actions, 32)
    return s


def make_params(spec, seed: int, dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic, portable parameter fill from ``numpy.random.RandomState``.

    Distributions follow the reference's init (Xavier-uniform gain 1 on Linear
    weights, got_sac_network.py:30-33; randn pos-embedding, GoalFormer.py:142;
    PyTorch-default uniform biases) but gains/LN-biases are perturbed away from
    1/0 so that a kernel ignoring them cannot pass parity.
    """
    rs = np.random.RandomState(seed)
    out: Dict[str, Tensor] = {}
    for key, shape, kind in spec:
        if kind == "randn":
            a = rs.standard_normal(shape)
        elif kind == "xavier":
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rs.uniform(-bound, bound, shape)
        elif kind == "conv":
            fan = shape[1] * shape[2] * shape[3]
            a = rs.uniform(-1.0, 1.0, shape) / math.sqrt(fan)
        elif kind == "bias":
            a = rs.uniform(-0.05, 0.05, shape)
        elif kind == "gain":
            a = 1.0 + rs.uniform(-0.2, 0.2, shape)
        elif kind == "lnbias":
            a = rs.uniform(-0.1, 0.1, shape)
        else:
            raise ValueError(kind)
        out[key] = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return out


def make_inputs(cfg: GoTConfig, batch: int, seed: int, dtype=torch.float32):
    """Synthetic frames/goals/actions per SURVEY.md section 8(d) "Synthetic inputs"."""
    rs = np.random.RandomState(seed + 100003)
    img = rs.random_sample((batch,) + tuple(cfg.image))
    pstate = np.stack([rs.uniform(0.0, 1.0, batch), rs.uniform(-1.0, 1.0, batch)], 1)
    act = rs.uniform(-1.0, 1.0, (batch, 2))
    tgt = rs.standard_normal((batch, 1))
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return f(img), f(pstate), f(act), f(tgt)


# --------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------
def patchify(img: Tensor, cfg: GoTConfig) -> Tensor:
    """'b (h p1) (w p2) -> b (h w) (p1 p2)'  (GoalFormer.py:138).

    Pixel order inside a patch is row-major (p1 * pw + p2); patch order is
    h * Wp + w.
    """
    B = img.shape[0]
    gh, gw = cfg.grid
    ph, pw = cfg.patch
    x = img.reshape(B, gh, ph, gw, pw).permute(0, 1, 3, 2, 4)
    return x.reshape(B, gh * gw, ph * pw)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """nn.Linear: y = x W^T + b."""
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm(dim) of PreNorm (GoalFormer.py:34,37): biased variance, eps 1e-5."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * gamma + beta


def gelu_exact(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (GoalFormer.py:44)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def rms_norm(x: Tensor, g: Tensor) -> Tensor:
    """RMSNorm.forward (GoalFormer.py:120-122): F.normalize(x) * sqrt(D) * g, eps 1e-12."""
    n = torch.sqrt((x * x).sum(-1, keepdim=True)).clamp_min(1e-12)
    return x / n * math.sqrt(x.shape[-1]) * g


def attention(x: Tensor, w_qkv: Tensor, w_out: Optional[Tensor], b_out: Optional[Tensor], heads: int, dim_head: int) -> Tensor:
    """Attention.forward (GoalFormer.py:71-82).  ``w_out is None``: ``to_out`` is ``nn.Identity()`` (GoalFormer.py:56,66-69).

    to_qkv rows are ordered [q(h0..hH-1) | k | v], 64 columns per head
    (chunk(3) then 'b n (h d) -> b h n d', GoalFormer.py:72-73).
    """
    B, N, _ = x.shape
    I = heads * dim_head
    qkv = linear(x, w_qkv)                                   # :72
    q, k, v = (qkv[..., j * I:(j + 1) * I].reshape(B, N, heads, dim_head).permute(0, 2, 1, 3)
               for j in range(3))                            # :73
    dots = (q @ k.transpose(-1, -2)) * (dim_head ** -0.5)    # :75, scale :59
    attn = torch.softmax(dots, dim=-1)                       # :77
    out = attn @ v                                           # :80
    out = out.permute(0, 2, 1, 3).reshape(B, N, I)           # :81
    if w_out is None:                                        # :66-69 project_out False (heads == 1 and dim_head == dim)
        return out
    return linear(out, w_out, b_out)                         # :82


def feed_forward(x: Tensor, w1, b1, w2, b2) -> Tensor:
    """FeedForward.forward (GoalFormer.py:42-50), dropout p=0."""
    return linear(gelu_exact(linear(x, w1, b1)), w2, b2)


# --------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------
def got_embed(p: Dict[str, Tensor], img: Tensor, goal: Tensor, cfg: GoTConfig,
              drop_mask: Optional[Tensor] = None, drop_p: float = 0.1, prefix: str = "trans.") -> Tensor:
    """Token assembly: GoT.forward lines GoalFormer.py:157-163.

    ``drop_mask`` (B,N,D) of {0,1} reproduces train-mode ``nn.Dropout(0.1)``
    (keep -> x/(1-p)); ``None`` is eval mode.
    """
    x = linear(patchify(img, cfg), p[prefix + "to_patch_embedding.1.weight"],
               p[prefix + "to_patch_embedding.1.bias"])                       # :157
    x = torch.cat([goal.unsqueeze(1), x], dim=1)                              # :160-161
    x = x + p[prefix + "pos_embedding"][:, :x.shape[1]]                       # :162
    if drop_mask is not None:                                                 # :163
        x = x * drop_mask / (1.0 - drop_p)
    return x


def got_block(p: Dict[str, Tensor], x: Tensor, i: int, cfg: GoTConfig, prefix: str = "trans.") -> Tensor:
    """One Transformer layer: x = attn(LN(x)) + x; x = ff(LN(x)) + x (GoalFormer.py:101-105)."""
    lp = f"{prefix}transformer.layers.{i}."
    h = layer_norm(x, p[lp + "0.norm.weight"], p[lp + "0.norm.bias"])
    x = attention(h, p[lp + "0.fn.to_qkv.weight"], p.get(lp + "0.fn.to_out.0.weight"),
                  p.get(lp + "0.fn.to_out.0.bias"), cfg.heads, cfg.dim_head) + x
    h = layer_norm(x, p[lp + "1.norm.weight"], p[lp + "1.norm.bias"])
    x = feed_forward(h, p[lp + "1.fn.net.0.weight"], p[lp + "1.fn.net.0.bias"],
                     p[lp + "1.fn.net.3.weight"], p[lp + "1.fn.net.3.bias"]) + x
    return x


def got_forward(p: Dict[str, Tensor], img: Tensor, goal: Tensor, cfg: GoTConfig,
                drop_mask: Optional[Tensor] = None, prefix: str = "trans.",
                return_tokens: bool = False, pool: str = "cls"):
    """GoT.forward (GoalFormer.py:156-171): pool='cls' takes token 0, pool='mean' the token mean (:167) -> RMSNorm."""
    x = got_embed(p, img, goal, cfg, drop_mask, prefix=prefix)
    toks = [x]
    for i in range(cfg.depth):                                                # :165
        x = got_block(p, x, i, cfg, prefix)
        toks.append(x)
    pooled = x.mean(dim=1) if pool == "mean" else x[:, 0]                       # :167
    feat = rms_norm(pooled, p[prefix + "layer_norm.g"])                       # :170
    return (feat, toks) if return_tokens else feat


# --------------------------------------------------------------------------
# bf16 configuration (BASELINE config 5): the same GoT.forward with the STORAGE roundings of the HIP bf16 path
# modelled (every GEMM operand -- patches, LayerNorm outputs, q/k/v, attention probabilities and output, GELU
# output, the to_out / fc2 branch outputs, the GEMM weights -- rounded to bf16; residual stream, statistics, biases, softmax and all sums in the
# working dtype).  Pinned two ways in tests/: against the fp32 restatement above (distance = the precision cost of
# bf16 storage) and against the reference run under torch.autocast(bfloat16) (tests/golden/make_golden_bf16.py).
# --------------------------------------------------------------------------
def rb(x: Tensor) -> Tensor:
    """round to bf16 (nearest even) and back"""
    return x.to(torch.bfloat16).to(x.dtype)


def got_forward_bf16(p: Dict[str, Tensor], img: Tensor, goal: Tensor, cfg: GoTConfig,
                     drop_mask: Optional[Tensor] = None, prefix: str = "trans.", pool: str = "cls",
                     drop_p: float = 0.1) -> Tensor:
    x = linear(rb(patchify(img, cfg)), rb(p[prefix + "to_patch_embedding.1.weight"]),
               p[prefix + "to_patch_embedding.1.bias"])                       # GoalFormer.py:157
    x = torch.cat([goal.unsqueeze(1), x], dim=1)                              # :160-161
    x = x + p[prefix + "pos_embedding"][:, :x.shape[1]]                       # :162
    if drop_mask is not None:                                                 # :163
        x = x * drop_mask / (1.0 - drop_p)
    B, N, _ = x.shape
    H, dh = cfg.heads, cfg.dim_head
    I = H * dh
    for i in range(cfg.depth):                                                # :165, 101-105
        lp = f"{prefix}transformer.layers.{i}."
        h = rb(layer_norm(x, p[lp + "0.norm.weight"], p[lp + "0.norm.bias"]))
        qkv = rb(linear(h, rb(p[lp + "0.fn.to_qkv.weight"])))                 # :72
        q, k, v = (qkv[..., j * I:(j + 1) * I].reshape(B, N, H, dh).permute(0, 2, 1, 3) for j in range(3))
        dots = (q @ k.transpose(-1, -2)) * (dh ** -0.5)                       # :75
        e = torch.exp(dots - dots.amax(-1, keepdim=True))
        out = (rb(e) @ v) / e.sum(-1, keepdim=True)                           # :77-80 (probabilities stored bf16, sum fp32)
        out = rb(out.permute(0, 2, 1, 3).reshape(B, N, I))                    # :81
        x = rb(linear(out, rb(p[lp + "0.fn.to_out.0.weight"]), p[lp + "0.fn.to_out.0.bias"])) + x   # branch output stored bf16
        h = rb(layer_norm(x, p[lp + "1.norm.weight"], p[lp + "1.norm.bias"]))
        a = rb(gelu_exact(linear(h, rb(p[lp + "1.fn.net.0.weight"]), p[lp + "1.fn.net.0.bias"])))
        x = rb(linear(a, rb(p[lp + "1.fn.net.3.weight"]), p[lp + "1.fn.net.3.bias"])) + x
    pooled = x.mean(dim=1) if pool == "mean" else x[:, 0]                     # :167
    return rms_norm(pooled, p[prefix + "layer_norm.g"])                       # :170


# --------------------------------------------------------------------------
# heads
# --------------------------------------------------------------------------
def policy_forward(p, istate, pstate, cfg: GoTConfig, drop_mask=None):
    """GoTPolicy.forward (got_sac_network.py:221-236): fc_embed has NO activation."""
    goal = linear(pstate, p["fc_embed.weight"], p["fc_embed.bias"])           # :226
    feat = got_forward(p, istate, goal, cfg, drop_mask)                       # :228
    x = torch.relu(linear(feat, p["fc1.weight"], p["fc1.bias"]))              # :230
    x = torch.relu(linear(x, p["fc2.weight"], p["fc2.bias"]))                 # :231
    mean = linear(x, p["mean_linear.weight"], p["mean_linear.bias"])          # :233
    log_std = linear(x, p["log_std_linear.weight"], p["log_std_linear.bias"])  # :234
    return mean, log_std.clamp(LOG_SIG_MIN, LOG_SIG_MAX)                      # :235


def policy_sample(p, istate, pstate, cfg: GoTConfig, noise: Tensor, drop_mask=None,
                  action_scale: float = 1.0, action_bias: float = 0.0):
    """GoTPolicy.sample (got_sac_network.py:238-251) with the N(0,1) draw injected."""
    mean, log_std = policy_forward(p, istate, pstate, cfg, drop_mask)
    std = log_std.exp()                                                       # :240
    x_t = mean + std * noise                                                  # :242 rsample
    y_t = torch.tanh(x_t)                                                     # :243
    action = y_t * action_scale + action_bias                                 # :245
    log_prob = -((x_t - mean) ** 2) / (2 * std * std) - log_std - math.log(math.sqrt(2 * math.pi))  # :246
    log_prob = log_prob - torch.log(action_scale * (1 - y_t * y_t) + EPSILON)  # :248
    log_prob = log_prob.sum(1, keepdim=True)                                  # :249
    return action, log_prob, torch.tanh(mean) * action_scale + action_bias    # :250-251


def qnet_forward(p, istate, pstate, a, cfg: GoTConfig, drop_mask=None):
    """GoTQNetwork.forward (got_sac_network.py:107-123): fc_embed WITH relu, twin MLPs."""
    goal = torch.relu(linear(pstate, p["fc_embed.weight"], p["fc_embed.bias"]))  # :111
    feat = got_forward(p, istate, goal, cfg, drop_mask)                       # :112
    x = torch.cat([feat, a], dim=1)                                           # :114
    q1 = torch.relu(linear(x, p["fc1.weight"], p["fc1.bias"]))                # :115
    q1 = torch.relu(linear(q1, p["fc2.weight"], p["fc2.bias"]))               # :116
    q1 = linear(q1, p["fc3.weight"], p["fc3.bias"])                           # :117
    q2 = torch.relu(linear(x, p["fc11.weight"], p["fc11.bias"]))              # :119
    q2 = torch.relu(linear(q2, p["fc21.weight"], p["fc21.bias"]))             # :120
    q2 = linear(q2, p["fc31.weight"], p["fc31.bias"])                         # :121
    return q1, q2


def detpolicy_forward(p, istate, pstate, cfg: GoTConfig, drop_mask=None,
                      action_scale: float = 1.0, action_bias: float = 0.0):
    """DeterministicGoTPolicy.forward (got_sac_network.py:425-436)."""
    goal = linear(pstate, p["fc_embed.weight"], p["fc_embed.bias"])           # :429
    feat = got_forward(p, istate, goal, cfg, drop_mask)                       # :430
    x = torch.relu(linear(feat, p["fc1.weight"], p["fc1.bias"]))              # :433
    x = torch.relu(linear(x, p["fc2.weight"], p["fc2.bias"]))                 # :434
    return torch.tanh(linear(x, p["mean_linear.weight"], p["mean_linear.bias"])) * action_scale + action_bias  # :435


# --------------------------------------------------------------------------
# CNN critic / actor
# --------------------------------------------------------------------------
def cnn_qnet_param_spec(nb_actions: int = 2, nb_pstate: int = 2):
    """QNetwork keys (got_sac_network.py:126-144)."""
    s = [("conv1.weight", (16, 1, 5, 5), "conv"), ("conv1.bias", (16,), "bias"),
         ("conv2.weight", (64, 16, 5, 5), "conv"), ("conv2.bias", (64,), "bias"),
         ("conv3.weight", (256, 64, 5, 5), "conv"), ("conv3.bias", (256,), "bias")]
    s += _lin("fc1", 128, 256 + 32 + nb_actions) + _lin("fc2", 32, 128) + _lin("fc3", nb_actions, 32)
    s += _lin("fc_embed", 32, nb_pstate)
    s += _lin("fc11", 128, 256 + 32 + nb_actions) + _lin("fc21", 32, 128) + _lin("fc31", nb_actions, 32)
    return s


def cnn_policy_param_spec(nb_actions: int = 2, nb_pstate: int = 2):
    """GaussianPolicy keys (got_sac_network.py:259-274)."""
    s = [("conv1.weight", (16, 1, 5, 5), "conv"), ("conv1.bias", (16,), "bias"),
         ("conv2.weight", (64, 16, 5, 5), "conv"), ("conv2.bias", (64,), "bias"),
         ("conv3.weight", (256, 64, 5, 5), "conv"), ("conv3.bias", (256,), "bias")]
    s += _lin("fc_embed", 32, nb_pstate) + _lin("fc1", 128, 256 + 32) + _lin("fc2", 32, 128)
    s += _lin("mean_linear", nb_actions, 32) + _lin("log_std_linear", nb_actions, 32)
    return s


def cnn_features(p, istate):
    """unsqueeze(1) -> relu(conv1) -> relu(conv2) -> relu(conv3) -> global average (got_sac_network.py:150-155)."""
    x = istate.unsqueeze(1)
    for i in (1, 2, 3):
        x = torch.relu(torch.nn.functional.conv2d(x, p[f"conv{i}.weight"], p[f"conv{i}.bias"], stride=2))
    return x.mean(dim=(2, 3))


def cnn_qnet_forward(p, istate, pstate, a):
    """QNetwork.forward (got_sac_network.py:146-170)."""
    x1 = cnn_features(p, istate)
    x2 = torch.relu(linear(pstate, p["fc_embed.weight"], p["fc_embed.bias"]))            # :158
    x = torch.cat([x1, x2, a], dim=1)                                                      # :160
    q1 = torch.relu(linear(x, p["fc1.weight"], p["fc1.bias"]))
    q1 = torch.relu(linear(q1, p["fc2.weight"], p["fc2.bias"]))
    q1 = linear(q1, p["fc3.weight"], p["fc3.bias"])
    q2 = torch.relu(linear(x, p["fc11.weight"], p["fc11.bias"]))
    q2 = torch.relu(linear(q2, p["fc21.weight"], p["fc21.bias"]))
    q2 = linear(q2, p["fc31.weight"], p["fc31.bias"])
    return q1, q2


def cnn_policy_forward(p, istate, pstate):
    """GaussianPolicy.forward (got_sac_network.py:288-308): goal embedding WITHOUT activation."""
    x = torch.cat([cnn_features(p, istate), linear(pstate, p["fc_embed.weight"], p["fc_embed.bias"])], dim=1)
    x = torch.relu(linear(x, p["fc1.weight"], p["fc1.bias"]))
    x = torch.relu(linear(x, p["fc2.weight"], p["fc2.bias"]))
    mean = linear(x, p["mean_linear.weight"], p["mean_linear.bias"])
    log_std = linear(x, p["log_std_linear.weight"], p["log_std_linear.bias"]).clamp(LOG_SIG_MIN, LOG_SIG_MAX)
    return mean, log_std


# --------------------------------------------------------------------------
# SAC loss arithmetic the path is differentiated through (DRL.py:390-432)
# --------------------------------------------------------------------------
def sac_critic_loss(q1: Tensor, q2: Tensor, y: Tensor) -> Tensor:
    """qf_loss = mse(q1, y) + mse(q2, y)  (DRL.py:397-399); y (B,1) broadcasts over the 2-wide Q."""
    yb = y.expand_as(q1)
    return ((q1 - yb) ** 2).mean() + ((q2 - yb) ** 2).mean()


def sac_actor_loss(alpha: float, log_pi: Tensor, q1_pi: Tensor, q2_pi: Tensor) -> Tensor:
    """policy_loss = (alpha * log_pi - min(q1_pi, q2_pi)).mean()  (DRL.py:408-410)."""
    return (alpha * log_pi - torch.minimum(q1_pi, q2_pi)).mean()


def soft_update(target: Dict[str, Tensor], source: Dict[str, Tensor], tau: float) -> None:
    """utils.py:31-33: target <- target*(1-tau) + source*tau."""
    for k in target:
        target[k].mul_(1.0 - tau).add_(source[k], alpha=tau)


# --------------------------------------------------------------------------
# depth-frame preprocessing in front of the path (env_lab.py:420-434 listener_callback, :78-89 add_nose,
# :69-76 blurring, :33-39 get_center_band, :295-299 / :348-349 resize + /255).
# PARITY UNPINNED: the reference does this with OpenCV (cv2.normalize / GaussianBlur / resize), which is not installed
# here and has no fixtures in the reference; the functions below restate OpenCV's published semantics for float32 images
#   normalize(NORM_MINMAX, 0, 255): dst = src * a + b, a = 255 / (max - min) (0 when max - min <= DBL_EPSILON), b = -min * a
#   .astype(np.uint8): truncation toward zero
#   GaussianBlur(ksize, sigma 0): separable, BORDER_REFLECT_101; ksize 5 uses the fixed kernel [1, 4, 6, 4, 1] / 16, ksize 11
#       sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8 = 2.0 with exp(-(i - c)^2 / (2 sigma^2)) normalised to sum 1
#   resize(INTER_LINEAR): source coordinate (dst + 0.5) * scale - 0.5, index clamped to the image, weights (1 - f, f)
# and are checked only for self-consistency (and, for the resize, against torch's independent bilinear interpolation).
# --------------------------------------------------------------------------
def f4_gaussian_kernel(ksize: int) -> np.ndarray:
    if ksize == 5:
        return np.array([0.0625, 0.25, 0.375, 0.25, 0.0625], np.float32)
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.abs(i) % p
    return np.where(i >= n, p - i, i)


def f4_gaussian_blur(img: np.ndarray, ksize: int) -> np.ndarray:
    """cv2.GaussianBlur(img, (ksize, ksize), 0) on (..., H, W) float32 images."""
    k = f4_gaussian_kernel(ksize)
    r = ksize // 2
    H, W = img.shape[-2:]
    cols = _reflect101(np.arange(W)[:, None] + np.arange(-r, r + 1)[None, :], W)      # (W, ksize)
    rows = _reflect101(np.arange(H)[:, None] + np.arange(-r, r + 1)[None, :], H)
    tmp = (img[..., :, cols] * k).sum(-1, dtype=np.float32)                           # horizontal pass
    return (tmp[..., rows, :] * k[None, :, None]).sum(-2, dtype=np.float32)           # vertical pass


def f4_depth_to_uint8(img: np.ndarray) -> np.ndarray:
    """listener_callback for float depth (env_lab.py:424-426): MINMAX-normalise to 0..255, truncate to integers (kept as float32)."""
    img = img.astype(np.float32)
    lo, hi = img.min(axis=(-2, -1), keepdims=True), img.max(axis=(-2, -1), keepdims=True)
    d = hi.astype(np.float64) - lo
    a = np.where(d > np.finfo(np.float64).eps, 255.0 / np.where(d > 0, d, 1.0), 0.0)
    b = -lo * a
    return np.trunc((img * a.astype(np.float32) + b.astype(np.float32)).astype(np.float32))


def f4_add_nose(img: np.ndarray, noise: np.ndarray) -> np.ndarray:
    """add_nose (env_lab.py:78-89) with the Gaussian draw supplied (noise = N(0, noise_level)): clip to 0..255, 5x5 blur."""
    return f4_gaussian_blur(np.clip(img.astype(np.float32) + noise.astype(np.float32), 0, 255), 5)


def f4_blurring(img: np.ndarray) -> np.ndarray:
    """blurring (env_lab.py:69-76): 11x11 Gaussian blur of the horizontal centre band of height H // 5 (borders reflect inside the band)."""
    H = img.shape[-2]
    bh = H // 5
    y1 = H // 2 - bh // 2
    out = img.astype(np.float32).copy()
    out[..., y1:y1 + bh, :] = f4_gaussian_blur(out[..., y1:y1 + bh, :], 11)
    return out


def f4_resize_to_state(img: np.ndarray, size=(128, 160)) -> np.ndarray:
    """cv2.resize(img, (160, 128)) / 255 (env_lab.py:295,299): bilinear, half-pixel centres, clamped."""
    Hd, Wd = size
    Hs, Ws = img.shape[-2:]

    def axis(nd, ns):
        f = (np.arange(nd, dtype=np.float32) + 0.5) * np.float32(ns / nd) - 0.5
        i0 = np.floor(f).astype(np.int64)
        w = (f - i0).astype(np.float32)
        w = np.where(i0 < 0, 0.0, w).astype(np.float32)
        i0 = np.clip(i0, 0, ns - 1)
        w = np.where(i0 >= ns - 1, 0.0, w).astype(np.float32)
        return i0, np.minimum(i0 + 1, ns - 1), w

    x0, x1, wx = axis(Wd, Ws)
    y0, y1, wy = axis(Hd, Hs)
    img = img.astype(np.float32)
    h = img[..., :, x0] * (1 - wx) + img[..., :, x1] * wx
    out = h[..., y0, :] * (1 - wy)[:, None] + h[..., y1, :] * wy[:, None]
    return (out / np.float32(255)).astype(np.float32)


def f4_pipeline(depth: np.ndarray, noise: np.ndarray, size=(128, 160)) -> np.ndarray:
    """sensor depth frame -> encoder input: listener_callback (:420-434) then the resize of step() / reset() (:295-299)."""
    return f4_resize_to_state(f4_blurring(f4_add_nose(f4_depth_to_uint8(depth), noise)), size)
