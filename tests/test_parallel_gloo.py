"""CPU: the data-parallel gradient exchange (dgvit_amd.parallel.GradSync) with world_size 2 over gloo.
Each rank back-propagates its half of a batch through the CPU oracle (stand-in for the HIP modules, which
need a GPU); after sync() every rank must hold the gradient of the full-batch mean loss."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, ROOT


class OracleNet(torch.nn.Module):
    """Parameters named like GoTPolicy, forward through the oracle; includes never-used parameters."""

    def __init__(self, cfg, seed):
        super().__init__()
        self.cfg = cfg
        self.keys = []
        for k, v in O.make_params(O.policy_param_spec(cfg), seed).items():
            name = k.replace(".", "__")
            self.register_parameter(name, torch.nn.Parameter(v))
            self.keys.append((k, name))

    def forward(self, img, pstate):
        p = {k: getattr(self, n) for k, n in self.keys}
        return O.policy_forward(p, img, pstate, self.cfg)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        import dgvit_amd  # noqa: F401
        from dgvit_amd.parallel import GradSync
        cfg = O.GoTConfig(image=(16, 24), patch=(8, 8), dim=32, depth=1, heads=2, dim_head=32, mlp_dim=64)
        net = OracleNet(cfg, seed=100 + rank)          # different init per rank on purpose
        sync = GradSync([net], bucket_bytes=4096)      # tiny buckets: several all-reduces
        sync.broadcast_parameters(0)
        B = 8
        img, pstate, _, _ = O.make_inputs(cfg, B, 7)
        sl = slice(rank * B // world, (rank + 1) * B // world)
        for it in range(2):                             # second round exercises the flat-buffer views
            sync.zero_grad()
            mean, log_std = net(img[sl], pstate[sl])
            ((mean ** 2).mean() + (log_std ** 2).mean()).backward()
            sync.sync()
        grads = {k: getattr(net, n).grad.clone() if getattr(net, n).grad is not None else None for k, n in net.keys}
        state = {k: getattr(net, n).detach().clone() for k, n in net.keys}
        n_live = sync.grad_numel()
        # shared-storage path: gradients that are views of one flat buffer (what the fused HIP backward
        # produces) must be all-reduced IN PLACE, together with a few loose tensors
        plist = list(net.parameters())
        for q in plist:
            q.grad = None
        members, loose = plist[:6], plist[6:9]
        flat = torch.full((sum(q.numel() for q in members),), float(rank + 1))
        off = 0
        for q in members:
            q.grad = flat[off:off + q.numel()].view_as(q)
            off += q.numel()
        for q in loose:
            q.grad = torch.full_like(q, 10.0 * (rank + 1))
        sync.sync()
        assert torch.all(flat == 1.5), "flat buffer was not reduced in place"
        assert all(q.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for q in members)
        assert all(torch.all(q.grad == 15.0) for q in loose)
        assert sync.grad_numel() == flat.numel() + sum(q.numel() for q in loose)
        # overlapped form, host logic only (the HIP events need a GPU: tests/test_gpu_ddp.py): two middle ranges of the flat buffer were
        # reduced "early" (as GradSync._on_grads_ready does from inside a backward); sync() must exchange exactly the three gaps
        # around them plus the loose tensors, wait for everything, and scale the whole buffer once
        flat.fill_(float(rank + 1))
        for q in loose:
            q.grad.fill_(10.0 * (rank + 1))
        key, n = flat.untyped_storage().data_ptr(), flat.numel()
        a, b, c, d = n // 5, 2 * n // 5, 3 * n // 5, 4 * n // 5
        for lo, hi in ((c, d), (a, b)):                  # last block first, as the backward finishes them
            sync._early.append((key, lo, hi, dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)))
        sync.sync()
        assert torch.all(flat == 1.5) and all(torch.all(q.grad == 15.0) for q in loose) and not sync._early
        # a reduced buffer that is not the parameters' .grad (autograd accumulated into older tensors) is refused
        other = torch.ones(8)
        sync._early.append((other.untyped_storage().data_ptr(), 0, 8, dist.all_reduce(other, async_op=True)))
        try:
            sync.sync()
            raise AssertionError("sync() accepted an early reduction of a foreign buffer")
        except RuntimeError as e:
            assert "zero_grad" in str(e)
        # numpy payloads: torch tensors would travel as shared-memory handles that die with the worker
        out.put((rank, {k: (None if g is None else g.numpy().copy()) for k, g in grads.items()},
                 {k: v.numpy().copy() for k, v in state.items()}, n_live))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradsync_world2_matches_full_batch():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda t: t[0])
    (_, g0, s0, n0), (_, g1, s1, n1) = results
    t = lambda d: {k: (None if v is None else torch.from_numpy(v)) for k, v in d.items()}
    g0, s0, g1, s1 = t(g0), t(s0), t(g1), t(s1)
    # weights were broadcast from rank 0
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    # single-process reference: full batch, mean loss, rank-0 weights
    cfg = O.GoTConfig(image=(16, 24), patch=(8, 8), dim=32, depth=1, heads=2, dim_head=32, mlp_dim=64)
    p = {k: v.clone().requires_grad_(True) for k, v in s0.items()}
    img, pstate, _, _ = O.make_inputs(cfg, 8, 7)
    mean, log_std = O.policy_forward(p, img, pstate, cfg)
    ((mean ** 2).mean() + (log_std ** 2).mean()).backward()
    live = 0
    for k in p:
        if p[k].grad is None:
            assert g0[k] is None and g1[k] is None, f"{k} must be skipped by the exchange (never gets a gradient)"
            continue
        live += p[k].numel()
        assert torch.allclose(g0[k], p[k].grad, rtol=1e-4, atol=1e-6), k
        assert torch.equal(g0[k], g1[k]), f"{k}: ranks disagree after all-reduce"
    assert n0 == n1 == live
    assert any(p[k].grad is None for k in p), "unused parameters (cls_token, mlp_head) should exist"
