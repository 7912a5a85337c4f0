"""HIP-graph capture of launch-bound steps.

At the reference's shipped training size (batch 32, L4/H4/D64 actor + CNN critic, config.yaml:11,58-63) one SAC
``learn()`` step is ~250 small kernels: on MI355X the GPU finishes them faster than Python can enqueue them
(measured: 5.0 ms per step, 4.8 ms of it host time).  Everything on the path is capturable -- the C library never
allocates or synchronises, the dropout seed and the Adam step counter can live in device memory (ABI v2) -- so a
whole step (forward, backward, ``FlatAdam(capturable=True).step()``, Polyak update) records into ONE graph and
replays with a single launch.
"""
from typing import Callable

import torch


class GraphedStep:
    """``GraphedStep(fn)`` runs ``fn()`` a few times eagerly (lazy initialisation, allocator warm-up), records it into
    a HIP graph and replays it on ``__call__``.

    ``fn`` takes no arguments and must read its inputs from pre-allocated device tensors that the caller refills
    in place (``static_img.copy_(batch)``) before each replay; whatever ``fn`` returns are static output tensors
    that every replay overwrites.  Inside ``fn`` use ``FlatAdam(..., capturable=True)``, ``zero_grad(set_to_none=True)``
    and no host synchronisation (``.item()``, ``.cpu()``, ``Normal(..., validate_args=True)``).  Modules in train mode
    draw their dropout seed on the device while capturing, so every replay uses a fresh mask.
    """

    def __init__(self, fn: Callable[[], object], warmup: int = 3):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a ROCm device")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def __call__(self):
        self.graph.replay()
        return self.outputs
