"""Batch-sharded data parallelism for the DGViT networks: one process per GPU, gradients all-reduced with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for tests).

Every frame is encoded independently, so rank r simply takes its own frames; the only
exchange is one all-reduce(sum)/world of the gradients after backward.  The gradients of all parameters that
receive one live in ONE flat fp32 buffer (``p.grad`` are views into it), so the exchange is a few large
bucketed all-reduces -- sized for xGMI's per-link ring bandwidth -- with no flatten/unflatten copies.
Parameters that never get a gradient (``cls_token``, ``mlp_head.*``, the dead ``conv1-3``) are
discovered on the first call and left out.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, modules: Iterable[torch.nn.Module], process_group=None, bucket_bytes: int = 64 << 20):
        self.params: List[torch.nn.Parameter] = []
        seen = set()
        for m in modules:
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    self.params.append(p)
        self.group = process_group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.flat: Optional[torch.Tensor] = None
        self.live: List[torch.nn.Parameter] = []
        self.views: List[torch.Tensor] = []

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every rank start from rank `src`'s weights (DRL.py builds nets from a per-process seed)."""
        if self.world == 1:
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)

    def _adopt(self) -> None:
        """Move the existing .grad tensors into one flat buffer and re-point .grad at views of it."""
        self.live = [p for p in self.params if p.grad is not None]
        total = sum(p.numel() for p in self.live)
        if total == 0:
            raise RuntimeError("GradSync: no parameter has a gradient; call after backward()")
        ref = self.live[0]
        self.flat = torch.zeros(total, dtype=ref.grad.dtype, device=ref.grad.device)
        self.views, off = [], 0
        for p in self.live:
            v = self.flat[off:off + p.numel()].view_as(p)
            v.copy_(p.grad)
            p.grad = v
            self.views.append(v)
            off += p.numel()

    def _intact(self) -> bool:
        return self.flat is not None and all(p.grad is v for p, v in zip(self.live, self.views)) and \
            all(p.grad is None for p in self.params if all(p is not q for q in self.live))

    def zero_grad(self) -> None:
        """Zero the flat buffer in one kernel, keeping .grad views alive (use instead of optimizer.zero_grad())."""
        if self.flat is None:
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()

    def sync(self) -> None:
        """All-reduce(sum)/world the gradients; call once after backward."""
        if self.flat is None or not self._intact():
            self._adopt()
        w = self.world
        if w == 1:
            return
        handles = []
        n = self.flat.numel()
        for off in range(0, n, self.bucket_elems):
            chunk = self.flat[off:min(n, off + self.bucket_elems)]
            handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        self.flat.div_(w)

    def grad_numel(self) -> int:
        return 0 if self.flat is None else self.flat.numel()
