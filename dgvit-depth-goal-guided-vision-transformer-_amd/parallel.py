"""Batch-sharded data parallelism for the DGViT networks: one process per GPU, gradients all-reduced with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for tests).

Every frame is encoded independently, so rank r simply takes its own frames; the only
exchange is one all-reduce(sum)/world of the gradients after backward.  The fused encoder backward already
writes all its parameter gradients into ONE flat fp32 buffer (``functional._GoTEncoder.backward``); ``sync()``
finds such shared buffers through the gradients' storages and all-reduces them in place -- a few large
collectives sized for xGMI's per-link ring bandwidth, no flatten/unflatten copies.  The remaining small
gradients (head Linears) are coalesced into one extra buffer.  Parameters that never get a gradient
(``cls_token``, ``mlp_head.*``, the dead ``conv1-3``) simply have ``grad is None`` and are skipped.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, modules: Iterable[torch.nn.Module], process_group=None, bucket_bytes: int = 64 << 20,
                 force_collective: bool = False):
        """``force_collective``: issue the all-reduce even in a world of one rank (a 1-rank RCCL group reduces a buffer onto
        itself) -- lets a single-GPU box exercise the exact code path the 8-GPU run takes."""
        self.params: List[torch.nn.Parameter] = []
        self.modules = list(modules)
        self.force_collective = bool(force_collective)
        seen = set()
        for m in self.modules:
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    self.params.append(p)
        self.group = process_group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._last_numel = 0

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every rank start from rank `src`'s weights (DRL.py builds nets from a per-process seed)."""
        if self.world == 1 and not self.force_collective:
            return
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradSync: torch.distributed is not initialised")
        from .optim import _HOME_OF
        done = set()
        for p in self.params:
            ent = _HOME_OF.get(p)
            if ent is not None and ent[0].intact():          # a flattened network travels as ONE buffer
                if id(ent[0]) not in done:
                    done.add(id(ent[0]))
                    dist.broadcast(ent[0].flat, src=src, group=self.group)
            else:
                dist.broadcast(p.data, src=src, group=self.group)
        # `.data` / flat-buffer writes do not move autograd's version counters: tell the bf16 weight caches
        from . import functional as F_
        F_.notify_parameters_changed(self.modules)

    def zero_grad(self) -> None:
        """Drop the gradients (set to None): the next backward's tensors are adopted as .grad without any
        accumulate or fill kernel.  Use instead of optimizer.zero_grad()."""
        for p in self.params:
            p.grad = None

    def _regions(self):
        """Group live gradients by storage; a group covering one contiguous range is reduced in place."""
        by_storage = {}
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            by_storage.setdefault(g.untyped_storage().data_ptr(), []).append(g)
        shared, loose = [], []
        for gs in by_storage.values():
            if len(gs) == 1:
                loose.append(gs[0])
                continue
            lo = min(g.storage_offset() for g in gs)
            hi = max(g.storage_offset() + g.numel() for g in gs)
            flat = torch.empty(0, dtype=gs[0].dtype, device=gs[0].device).set_(gs[0].untyped_storage(), lo, (hi - lo,))
            shared.append(flat)
        return shared, loose

    def sync(self) -> None:
        """All-reduce(sum)/world the gradients; call once after backward."""
        shared, loose = self._regions()
        self._last_numel = sum(p.grad.numel() for p in self.params if p.grad is not None)
        if self._last_numel == 0:
            raise RuntimeError("GradSync: no parameter has a gradient; call after backward()")
        w = self.world
        if w == 1 and not self.force_collective:
            return
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradSync: torch.distributed is not initialised")
        handles, bufs = [], list(shared)
        small = None
        if loose:
            small = torch.cat([g.reshape(-1) for g in loose])
            bufs.append(small)
        for buf in bufs:
            n = buf.numel()
            for off in range(0, n, self.bucket_elems):
                chunk = buf[off:min(n, off + self.bucket_elems)]
                handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        inv = 1.0 / w
        for buf in shared:
            buf.mul_(inv)
        if small is not None:
            small.mul_(inv)
            off = 0
            for g in loose:
                g.copy_(small[off:off + g.numel()].view_as(g))
                off += g.numel()

    def grad_numel(self) -> int:
        """Number of gradient elements exchanged by the last sync()."""
        return self._last_numel
