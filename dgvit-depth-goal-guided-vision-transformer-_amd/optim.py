"""Flat-buffer optimiser step and Polyak target update for the DGViT networks.

The reference calls ``torch.optim.Adam`` over ~70 tensors per network (DRL.py:126-168, 401-403, 412-414) and a
per-parameter Python loop ``soft_update`` (utils.py:31-33).  Here the parameters of a network are re-homed into
flat fp32 buffers (each ``nn.Parameter`` keeps its identity, its ``.data`` becomes a view), and both updates are
one HBM-bound HIP kernel per buffer (``dgvit_adam_step`` / ``dgvit_soft_update``).

The fused encoder backward already delivers all encoder gradients as views of one flat buffer laid out in
parameter-table order; when ``FlatAdam`` finds that layout it consumes the buffer in place, otherwise it gathers
the gradients with one multi-tensor copy.
"""
import ctypes
from typing import Iterable, List

import torch

from . import _lib
from .goalformer import GoT


def _al4(n: int) -> int:
    return (n + 3) & ~3


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Block:
    """A group of parameters living back to back (4-float aligned) in one flat buffer."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += _al4(p.numel())
        self.numel = off
        ref = params[0]
        if not ref.is_cuda:
            raise _lib.DgvitError("flat parameter buffers need the module on a ROCm device first (call .to(device) before)")
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=ref.device)
        for p, o in zip(params, self.offsets):
            v = self.flat[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.gflat = None

    def intact(self) -> bool:
        base = self.flat.data_ptr()
        return all(p.data.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def gather_grads(self) -> torch.Tensor:
        """The gradients as one flat tensor with this block's layout (zero copy when they already are)."""
        g0 = self.params[0].grad
        st = g0.untyped_storage().data_ptr()
        o0 = g0.storage_offset()
        if all(p.grad.untyped_storage().data_ptr() == st and p.grad.storage_offset() - o0 == o and p.grad.is_contiguous()
               for p, o in zip(self.params, self.offsets)) and o0 % 4 == 0:
            return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), o0, (self.numel,))
        if self.gflat is None:
            self.gflat = torch.zeros_like(self.flat)
        views = [self.gflat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        torch._foreach_copy_(views, [p.grad for p in self.params])
        return self.gflat


class FlatAdam:
    """``torch.optim.Adam`` semantics (amsgrad=False) with one HIP kernel per flat parameter block.

    ``modules``: the networks to optimise.  Every ``GoT`` encoder inside becomes one block in parameter-table
    order (matching the fused backward's gradient buffer); all remaining parameters that receive gradients form
    one more block, built on the first ``step()``.  Parameters that never get a gradient are left alone, like
    torch's optimisers do.  Build it after ``module.to(device)``.  ``capturable=True`` keeps the step counter in device
    memory (like torch's ``capturable`` optimisers) so that ``step()`` can be recorded into a HIP graph.
    """

    def __init__(self, modules: Iterable[torch.nn.Module], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 capturable: bool = False):
        self.capturable, self._step_dev = bool(capturable), None
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.modules = list(modules) if not isinstance(modules, torch.nn.Module) else [modules]
        self.step_count = 0
        self.blocks: List[_Block] = []
        taken = set()
        for m in self.modules:
            for sub in m.modules():
                if isinstance(sub, GoT):
                    table = [p for p in sub.param_table() if p.requires_grad]
                    if table and all(id(p) not in taken for p in table):
                        self.blocks.append(_Block(table))
                        taken.update(id(p) for p in table)
        self._taken = taken
        self._rest_built = False

    def _all_params(self):
        seen = set()
        for m in self.modules:
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    yield p

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self._all_params():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _build_rest(self) -> None:
        rest = [p for p in self._all_params() if id(p) not in self._taken and p.grad is not None]
        if rest:
            self.blocks.append(_Block(rest))
            self._taken.update(id(p) for p in rest)
        self._rest_built = True

    @torch.no_grad()
    def step(self) -> None:
        lib = _lib.load()
        if not self._rest_built:
            self._build_rest()
        self.step_count += 1
        step_ptr = ctypes.c_void_p(0)
        if self.capturable:
            if self._step_dev is None:
                self._step_dev = torch.full((1,), self.step_count - 1, dtype=torch.int64, device=self.blocks[0].flat.device)
            self._step_dev += 1                      # a device op: replayed with the graph
            step_ptr = ctypes.c_void_p(self._step_dev.data_ptr())
        for b in self.blocks:
            if any(p.grad is None for p in b.params):
                raise _lib.DgvitError("FlatAdam: a parameter of a flat block has no gradient this step")
            if not b.intact():
                raise _lib.DgvitError("FlatAdam: parameter storage was replaced (e.g. by .to()); rebuild the optimiser")
            g = b.gather_grads()
            with torch.cuda.device(b.flat.device):
                rc = lib.dgvit_adam_step(ctypes.c_void_p(b.flat.data_ptr()), ctypes.c_void_p(g.data_ptr()),
                                         ctypes.c_void_p(b.exp_avg.data_ptr()), ctypes.c_void_p(b.exp_avg_sq.data_ptr()), b.numel,
                                         self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, step_ptr, _stream())
            _lib.check(rc, "dgvit_adam_step")

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                "exp_avg": [b.exp_avg.clone() for b in self.blocks], "exp_avg_sq": [b.exp_avg_sq.clone() for b in self.blocks]}

    def load_state_dict(self, sd) -> None:
        if len(sd["exp_avg"]) != len(self.blocks):
            raise ValueError("optimizer state does not match the parameter blocks (run one step first to build them)")
        self.step_count = int(sd["step"])
        for b, m, v in zip(self.blocks, sd["exp_avg"], sd["exp_avg_sq"]):
            b.exp_avg.copy_(m)
            b.exp_avg_sq.copy_(v)


def flatten_parameters(module: torch.nn.Module) -> torch.Tensor:
    """Re-home ALL parameters of ``module`` (in ``parameters()`` order) into one flat buffer; returns it and
    remembers it on the module for ``soft_update``.  Call after ``.to(device)`` and before building optimisers."""
    blk = _Block(list(module.parameters()))
    module._dgvit_flat = blk
    return blk.flat


@torch.no_grad()
def soft_update(target: torch.nn.Module, source: torch.nn.Module, tau: float) -> None:
    """target <- target*(1-tau) + source*tau (utils.py:31-33).  One HIP kernel when both modules went through
    ``flatten_parameters``; otherwise the reference's per-parameter loop."""
    tb, sb = getattr(target, "_dgvit_flat", None), getattr(source, "_dgvit_flat", None)
    if tb is not None and sb is not None and tb.numel == sb.numel and tb.intact() and sb.intact():
        lib = _lib.load()
        with torch.cuda.device(tb.flat.device):
            rc = lib.dgvit_soft_update(ctypes.c_void_p(tb.flat.data_ptr()), ctypes.c_void_p(sb.flat.data_ptr()), tb.numel, float(tau),
                                       _stream())
        _lib.check(rc, "dgvit_soft_update")
        return
    for tp, sp in zip(target.parameters(), source.parameters()):
        tp.data.mul_(1.0 - tau).add_(sp.data, alpha=tau)


def hard_update(target: torch.nn.Module, source: torch.nn.Module) -> None:
    """utils.py:35-37."""
    soft_update(target, source, 1.0)
