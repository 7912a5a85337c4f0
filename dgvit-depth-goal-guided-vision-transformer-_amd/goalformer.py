"""MI355X-native twin of the reference encoder module ``GoalFormer.py``.

Same public class (``GoT``), constructor signature, attribute names and ``state_dict`` keys as the
reference (GoalFormer.py:123-171), so checkpoints and callers (got_sac_network.py:79-88,176-185) carry
over.  The sub-modules below are parameter containers only: ``GoT.forward`` hands all tensors to one
fused HIP forward/backward (functional.got_encoder); none of them runs PyTorch math.

Differences from the reference, on purpose:
  * ``patch_size`` is honoured (the reference hard-wires 16x20 / Linear(320, dim), GoalFormer.py:137-139);
    with patch_size=(16, 20) the parameter shapes are identical;
  * transformer ``dropout`` must be 0 (the only value the reference's nets ever use); ``pool`` 'cls' and 'mean' both work;
  * ``heads == 1 and dim_head == dim`` gives the reference's projection-less attention (``to_out = nn.Identity()``) on the fp32 path;
  * ``RMSNorm`` also works stand-alone (``unit_offset`` included).
"""
import math

import torch
from torch import nn

from . import functional as F_


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


class _Holder(nn.Module):
    """A sub-module that only owns parameters; calling it is a bug (the fused path reads the tensors)."""

    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} holds parameters for the fused DGViT HIP encoder and is not callable")


class RMSNorm(nn.Module):
    """GoalFormer.py:107-122.  Inside ``GoT`` it is a parameter holder (the fused encoder applies it to the pooled token); called
    on its own it runs the HIP RMSNorm kernels, ``unit_offset`` included (gain stored as g, applied as g + 1, init g = 0)."""

    def __init__(self, dim, unit_offset=False):
        super().__init__()
        self.unit_offset = unit_offset
        self.scale = dim ** 0.5
        self.g = nn.Parameter(torch.zeros(dim))
        nn.init.constant_(self.g, 1. - float(unit_offset))

    def forward(self, x):
        gain = self.g + float(self.unit_offset) if self.unit_offset else self.g
        return F_.rms_norm(x, gain)


class Patchify(_Holder):
    """Slot 0 of ``to_patch_embedding`` (einops Rearrange in the reference: no parameters)."""

    def __init__(self, p1, p2):
        super().__init__()
        self.p1, self.p2 = p1, p2


class Attention(_Holder):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)      # GoalFormer.py:56
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        # GoalFormer.py:66-69: without a projection ``to_out`` is nn.Identity() -- no ``to_out`` keys in the state_dict, and the fused
        # encoder adds the head's output straight into the residual stream (param_table passes None for the two slots)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout)) if project_out else nn.Identity()


class FeedForward(_Holder):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden_dim, dim),
                                 nn.Dropout(dropout))


class PreNorm(_Holder):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class Transformer(_Holder):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                           PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))])
            for _ in range(depth)])


def _weights_loaded(module, incompatible_keys):
    """load_state_dict post-hook (a module-level function: the module must stay picklable, attention_imitating.py:199)."""
    module._bf16_weights.invalidate()


class GoT(nn.Module):
    """Goal-guided ViT encoder: (B, H, W) depth frames + (B, dim) goal embedding -> (B, dim) features."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls', channels=3,
                 dim_head=64, dropout=0., emb_dropout=0.1):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(patch_size)
        self.layer_norm = RMSNorm(dim)
        assert image_height % patch_height == 0 and image_width % patch_width == 0, \
            'Image dimensions must be divisible by the patch size.'
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        if dropout != 0.:
            raise NotImplementedError("transformer dropout must be 0 (the reference never sets it)")
        num_patches = (image_height // patch_height) * (image_width // patch_width)
        # `channels` is accepted and ignored exactly like the reference (frames are single-channel, 3-D input)
        self.to_patch_embedding = nn.Sequential(Patchify(patch_height, patch_width),
                                                nn.Linear(patch_height * patch_width, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))   # unused, kept for checkpoint compatibility
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))  # unused, kept for checkpoints
        self._cfg = (image_height, image_width, patch_height, patch_width, dim, depth, heads, dim_head, mlp_dim,
                     1 if pool == 'mean' else 0, 0)       # last entry: schedule flags (set_schedule)
        self.compute_dtype = torch.float32
        self._bf16_weights = F_.Bf16Weights()
        self._grad_hook = None     # set by parallel.GradSync(overlap=True): called inside the backward with the gradient-ready events
        self.register_load_state_dict_post_hook(_weights_loaded)

    def __deepcopy__(self, memo):
        # the bf16 weight arena is a cache of THIS module's parameters: a copy (DRL.py:169 deep-copies the policy) starts its own
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = F_.Bf16Weights() if k == "_bf16_weights" else copy.deepcopy(v, memo)
        new._bf16_weights.frozen = self._bf16_weights.frozen
        return new

    def freeze_bf16_weights(self, frozen: bool = True):
        """bf16 configuration only.  By default the bf16 copies of the GEMM weights are re-packed from the fp32 masters before
        every forward (nothing cheaper can see a fused optimiser step or a ``param.data.copy_``).  ``freeze_bf16_weights()``
        declares that the weights no longer change (serving): the copies are packed once and reused.  Parameter writers of this
        package (FlatAdam, soft_update, hard_update, GradSync.broadcast_parameters, load_state_dict) still invalidate them;
        after any OTHER in-place write call ``freeze_bf16_weights()`` again (it drops the cached copies)."""
        self._bf16_weights.frozen = bool(frozen)
        self._bf16_weights.invalidate()
        return self

    def set_schedule(self, dense_last_block: bool = False, wgrad_overlap: bool = False):
        """Per-module schedule options, passed to the C ABI with every call (``dgvit_config.flags``; nothing global).
        ``dense_last_block``: run the whole last block instead of its token-0 rows only (identical results; A/B measurements).
        ``wgrad_overlap``: the backward runs the weight-gradient GEMMs on a helper stream beside the data-gradient chain
        (+3..5 % frames/s at BASELINE config 3; per-kernel timings stop being interpretable)."""
        from ._lib import FLAG_DENSE_LAST_BLOCK, FLAG_WGRAD_OVERLAP
        flags = (FLAG_DENSE_LAST_BLOCK if dense_last_block else 0) | (FLAG_WGRAD_OVERLAP if wgrad_overlap else 0)
        self._cfg = (*self._cfg[:10], flags)
        return self

    def set_compute_dtype(self, dtype):
        """torch.float32 (default: exact fp32 MFMA path) or torch.bfloat16 (BASELINE config 5: bf16 storage for the GEMM
        operands, fp32 residual stream / statistics / accumulation; parameters stay fp32 masters)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"compute dtype {dtype} unsupported (torch.float32 or torch.bfloat16)")
        self.compute_dtype = dtype
        return self

    def param_table(self):
        """Parameters in the order of include/dgvit_hip.h's table."""
        t = [self.pos_embedding, self.to_patch_embedding[1].weight, self.to_patch_embedding[1].bias, self.layer_norm.g]
        for attn, ff in self.transformer.layers:
            out = attn.fn.to_out[0] if isinstance(attn.fn.to_out, nn.Sequential) else None    # None: nn.Identity(), the slots stay empty
            t += [attn.norm.weight, attn.norm.bias, attn.fn.to_qkv.weight, out.weight if out else None, out.bias if out else None,
                  ff.norm.weight, ff.norm.bias, ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias]
        return t

    @staticmethod
    def draw_dropout_seed() -> int:
        """The Philox seed of one train-mode forward, from torch's CPU generator: follows ``torch.manual_seed`` (ranks seeded with
        base + rank draw different masks, a rank's sequence repeats) and needs no device sync."""
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def forward(self, img, goal):
        keep, seed = 1.0, 0
        if self.training and self.dropout.p > 0:
            keep = 1.0 - self.dropout.p
            if img.is_cuda and torch.cuda.is_current_stream_capturing():
                # inside a HIP-graph capture a host seed would be frozen into every replay: draw it on the device
                # (graph-safe generator) and let the dropout kernel read it from memory
                seed = torch.empty(1, dtype=torch.int64, device=img.device).random_()
            else:
                seed = self.draw_dropout_seed()
        params = self.param_table()
        if self.compute_dtype == torch.bfloat16:
            return F_.got_encoder_bf16(img, goal, self._cfg, params, self._bf16_weights, keep, seed, grad_hook=self._grad_hook)
        return F_.got_encoder(img, goal, self._cfg, params, keep, seed, grad_hook=self._grad_hook)
