"""Batch-sharded data parallelism for the DGViT networks: one process per GPU, gradients all-reduced with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU for tests).

Every frame is encoded independently, so rank r simply takes its own frames; the only
exchange is one all-reduce(sum)/world of the gradients after backward.  The fused encoder backward already
writes all its parameter gradients into ONE flat fp32 buffer (``functional._GoTEncoder.backward``); ``sync()``
finds such shared buffers through the gradients' storages and all-reduces them in place -- a few large
collectives sized for xGMI's per-link ring bandwidth, no flatten/unflatten copies.  The remaining small
gradients (head Linears) are coalesced into one extra buffer.  Parameters that never get a gradient
(``cls_token``, ``mlp_head.*``, the dead ``conv1-3``) simply have ``grad is None`` and are skipped.

``overlap=True``: the encoder backward records a HIP event where each transformer block's gradients are final
(include/dgvit_hip.h: dgvit_grad_events; blocks finish last-to-first).  GradSync then queues that block's all-reduce on a side
stream behind its event while the backward's kernels for the earlier blocks are still running, and ``sync()`` only exchanges
what is left (embedding, final norm, heads) and makes the caller's stream wait for everything.  On RCCL the reduction is
``ReduceOp.AVG`` (no separate 1/world pass); gloo has no AVG, so CPU runs keep sum + scale.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, modules: Iterable[torch.nn.Module], process_group=None, bucket_bytes: int = 64 << 20,
                 force_collective: bool = False, overlap: bool = False):
        """``force_collective``: issue the all-reduce even in a world of one rank (a 1-rank RCCL group reduces a buffer onto
        itself) -- lets a single-GPU box exercise the exact code path the 8-GPU run takes.
        ``overlap``: start each transformer block's all-reduce from inside the encoder backward (see the module docstring).
        ``zero_grad()`` of this class ARMS the hook and ``sync()`` disarms it: only the backward between the two starts early
        all-reduces.  Any other backward through the same encoder (DRL.py:407-413: the policy loss is back-propagated through the
        critic, whose gradients nobody wants) is ignored, so it can neither launch collectives nor leave stale entries behind."""
        self.params: List[torch.nn.Parameter] = []        # trainable: their gradients are exchanged
        self.all_params: List[torch.nn.Parameter] = []    # every parameter, frozen ones too: broadcast_parameters sends them all
        self.modules = list(modules)
        self.force_collective = bool(force_collective)
        seen = set()
        for m in self.modules:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    self.all_params.append(p)
                    if p.requires_grad:
                        self.params.append(p)
        self.group = process_group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._last_numel = 0
        self.overlap = bool(overlap)
        self._side = {}          # device -> side stream the early all-reduces are queued on
        self._early = []         # (flat buffer, lo, hi, work handle) of this step's early all-reduces; holding the buffer keeps its address
        #                          from being handed to another tensor while the entry lives (entries are matched by storage address)
        self._armed = False      # overlap: set by zero_grad(), cleared by sync() -- backward passes outside that window are ignored
        self.early_launches = 0  # all-reduces started from inside a backward so far (tests, timelines)
        self.on_block_queued = None   # optional callable(side_stream): after each block's all-reduces were queued (tools/overlap_timeline.py)
        if self.overlap:
            import os
            import warnings
            if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
                # measured (tools/overlap_queue_probe.py): with an RCCL process group alive and the HIP runtime's default of 4 hardware
                # queues, work queued behind a gradient-ready event runs only after the whole backward -- correct, but not overlapped
                warnings.warn("GradSync(overlap=True): export GPU_MAX_HW_QUEUES=8 (before the first HIP call) or the collectives' stream "
                              "shares a hardware queue with the compute stream and nothing overlaps", RuntimeWarning, stacklevel=2)
            from .goalformer import GoT
            for m in self.modules:
                for sub in m.modules():
                    if isinstance(sub, GoT):
                        sub._grad_hook = self._on_grads_ready

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every rank start from rank `src`'s weights (DRL.py builds nets from a per-process seed) -- frozen parameters
        included: a frozen encoder under a heads-only optimiser (DRL.py:145-148) must be the same network on every rank too."""
        if self.world == 1 and not self.force_collective:
            return
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradSync: torch.distributed is not initialised")
        from .optim import _HOME_OF
        done = set()
        for p in self.all_params:
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
        if self.overlap:
            self._drop_early()
            self._armed = True

    def _drop_early(self) -> None:
        """Early all-reduces nobody collected (a backward that was never followed by sync()): wait for them, then forget them."""
        stale, self._early = self._early, []
        for _, _, _, h in stale:
            h.wait()

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

    def _active(self) -> bool:
        return self.world > 1 or self.force_collective

    def _reduce_op(self, on_device: bool = True):
        """(op, scale afterwards?) for gradients on the GPU (``on_device``) or on the host: RCCL averages inside the collective, gloo
        can only sum.  A group with one backend per device type reports e.g. "cpu:gloo,cuda:nccl": the device's entry decides."""
        backend = str(dist.get_backend(self.group)).lower()
        per_device = dict(e.split(":", 1) for e in backend.split(",") if ":" in e)
        mine = per_device.get("cuda" if on_device else "cpu", backend if not per_device else "")
        if mine == "nccl":
            return dist.ReduceOp.AVG, False
        return dist.ReduceOp.SUM, True

    def _on_grads_ready(self, flat, ranges, events) -> None:
        """functional's gradient-ready hook: runs inside the encoder backward, after its kernels were queued.  ``ranges[k]`` of
        ``flat`` is final once ``events[k]`` has happened; its all-reduce goes to the side stream behind that event."""
        if not self._active() or not self._armed:
            return          # not this GradSync's backward (see __init__): nothing is launched, nothing is remembered
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradSync: torch.distributed is not initialised")
        key = flat.untyped_storage().data_ptr()
        if any(e[0].untyped_storage().data_ptr() == key for e in self._early):
            raise RuntimeError("GradSync(overlap=True): this gradient buffer is already being reduced")
        from . import _lib
        lib = _lib.load()
        dev = flat.device
        side = self._side.get(dev)
        if side is None:
            side = self._side[dev] = torch.cuda.Stream(device=dev)
        op, _ = self._reduce_op(flat.is_cuda)
        with torch.cuda.device(dev), torch.cuda.stream(side):
            for (lo, hi), ev in zip(ranges, events):
                _lib.check(lib.dgvit_stream_wait_event(side.cuda_stream, ev), "dgvit_stream_wait_event")
                for off in range(lo, hi, self.bucket_elems):
                    end = min(hi, off + self.bucket_elems)
                    h = dist.all_reduce(flat[off:end], op=op, group=self.group, async_op=True)   # RCCL's stream waits for `side`
                    self._early.append((flat, off, end, h))
                    self.early_launches += 1
                if self.on_block_queued is not None:
                    self.on_block_queued(side)

    def sync(self) -> None:
        """All-reduce the gradients to their mean over the ranks; call once after backward.  With ``overlap`` the transformer
        blocks' shares are already on their way (``_on_grads_ready``): only the rest is exchanged here, then the caller's
        stream waits for all of it."""
        self._armed = False
        shared, loose = self._regions()
        self._last_numel = sum(p.grad.numel() for p in self.params if p.grad is not None)
        if self._last_numel == 0:
            raise RuntimeError("GradSync: no parameter has a gradient; call after backward()")
        if not self._active():
            return
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradSync: torch.distributed is not initialised")
        some = shared[0] if shared else loose[0]
        op, scale = self._reduce_op(some.is_cuda)
        early, self._early = [(f.untyped_storage().data_ptr(), lo, hi, h) for f, lo, hi, h in self._early], []
        live = {g.untyped_storage().data_ptr() for g in shared}
        for key, lo, hi, h in early:
            if key not in live:     # autograd accumulated the new gradients into older .grad tensors: the early reduce missed them
                raise RuntimeError("GradSync(overlap=True): a reduced gradient buffer is not the parameters' .grad -- call "
                                   "GradSync.zero_grad() (grads set to None) before every backward")
        handles, bufs = [h for _, _, _, h in early], []
        for buf in shared:
            key, base = buf.untyped_storage().data_ptr(), buf.storage_offset()
            done = sorted((lo, hi) for k, lo, hi, _ in early if k == key)
            pos = base
            for lo, hi in done + [(base + buf.numel(), base + buf.numel())]:     # the gaps between the early ranges
                if lo > pos:
                    bufs.append(buf[pos - base:lo - base])
                pos = max(pos, hi)
        small = None
        if loose:
            small = torch.cat([g.reshape(-1) for g in loose])
            bufs.append(small)
        for buf in bufs:
            n = buf.numel()
            for off in range(0, n, self.bucket_elems):
                chunk = buf[off:min(n, off + self.bucket_elems)]
                handles.append(dist.all_reduce(chunk, op=op, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        if scale:
            inv = 1.0 / self.world
            for buf in shared:
                buf.mul_(inv)
            if small is not None:
                small.mul_(inv)
        if small is not None:
            off = 0
            for g in loose:
                g.copy_(small[off:off + g.numel()].view_as(g))
                off += g.numel()

    def grad_numel(self) -> int:
        """Number of gradient elements exchanged by the last sync()."""
        return self._last_numel
