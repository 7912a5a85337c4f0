"""CPU: the data-parallel gradient exchange (dgvit_amd.parallel.GradSync) with world_size 2 and 4 over gloo.
Each rank back-propagates its share of a batch through the CPU oracle (stand-in for the HIP modules, which
need a GPU); after sync() every rank must hold the gradient of the full-batch mean loss.  Buckets are 1024 floats, so
the larger parameters (to_qkv 6144, fc1 / fc2 2048 floats) are split across several all-reduces; one sub-module is frozen
(requires_grad off, the heads-only optimiser of DRL.py:145-148 freezes the encoder the same way) and must stay out of the exchange."""
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
        for k, n in net.keys:                          # frozen on every rank alike (same layout everywhere): no gradient, not exchanged
            if "to_patch_embedding" in k:
                getattr(net, n).requires_grad_(False)
        sync = GradSync([net], bucket_bytes=4096)      # tiny buckets: several all-reduces, parameters split across them
        assert all("to_patch_embedding" not in k for k, n in net.keys if any(getattr(net, n) is q for q in sync.params))
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
        plist = list(sync.params)                       # (the trainable ones)
        for q in plist:
            q.grad = None
        members, loose = plist[:6], plist[6:9]
        mean_rank = (world + 1) / 2.0                   # mean over the ranks of (rank + 1)
        flat = torch.full((sum(q.numel() for q in members),), float(rank + 1))
        off = 0
        for q in members:
            q.grad = flat[off:off + q.numel()].view_as(q)
            off += q.numel()
        for q in loose:
            q.grad = torch.full_like(q, 10.0 * (rank + 1))
        sync.sync()
        assert torch.all(flat == mean_rank), "flat buffer was not reduced in place"
        assert all(q.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for q in members)
        assert all(torch.all(q.grad == 10.0 * mean_rank) for q in loose)
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
            sync._early.append((flat, lo, hi, dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)))
        sync.sync()
        assert torch.all(flat == mean_rank) and all(torch.all(q.grad == 10.0 * mean_rank) for q in loose) and not sync._early
        # a reduced buffer that is not the parameters' .grad (autograd accumulated into older tensors) is refused
        other = torch.ones(8)
        sync._early.append((other, 0, 8, dist.all_reduce(other, async_op=True)))
        try:
            sync.sync()
            raise AssertionError("sync() accepted an early reduction of a foreign buffer")
        except RuntimeError as e:
            assert "zero_grad" in str(e)
        # arming (overlap): zero_grad() arms the gradient-ready hook and drops early reductions nobody collected, sync() disarms it; a
        # backward outside that window (DRL.py:407-413: the policy loss back-propagated through the critic) must start nothing
        ov = GradSync([net], bucket_bytes=4096, overlap=True)
        assert not ov._armed
        ov._on_grads_ready(flat, [(0, 8)], [None])           # disarmed: ignored before it touches the events
        assert not ov._early and ov.early_launches == 0
        ov.zero_grad()
        assert ov._armed
        ov._early.append((other, 0, 8, dist.all_reduce(other, async_op=True)))    # a stale entry from a backward that was never synced
        ov.zero_grad()
        assert ov._armed and not ov._early
        for q in loose:
            q.grad = torch.full_like(q, 10.0 * (rank + 1))
        ov.sync()
        assert not ov._armed and all(torch.all(q.grad == 10.0 * mean_rank) for q in loose)
        # numpy payloads: torch tensors would travel as shared-memory handles that die with the worker
        out.put((rank, {k: (None if g is None else g.numpy().copy()) for k, g in grads.items()},
                 {k: v.numpy().copy() for k, v in state.items()}, n_live))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_gradsync_matches_full_batch(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    import queue, time
    t_end = time.time() + 240
    while len(results) < world and time.time() < t_end:
        try:
            results.append(out.get(timeout=2))
        except queue.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a worker died: " + str([p.exitcode for p in procs])
    assert len(results) == world
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda t: t[0])
    t = lambda d: {k: (None if v is None else torch.from_numpy(v)) for k, v in d.items()}
    (_, g0, s0, n0), (_, g1, s1, n1) = results[0], results[-1]
    g0, s0, g1, s1 = t(g0), t(s0), t(g1), t(s1)
    # weights were broadcast from rank 0 (every rank started from its own seed)
    for _, _, sr, _ in results[1:]:
        for k in s0:
            assert torch.equal(s0[k], torch.from_numpy(sr[k])), k
    for _, gr, _, nr in results[1:]:
        assert nr == n0
        for k in g0:
            assert (g0[k] is None) == (gr[k] is None) and (g0[k] is None or torch.equal(g0[k], torch.from_numpy(gr[k]))), f"{k}: ranks disagree"
    # single-process reference: full batch, mean loss, rank-0 weights
    cfg = O.GoTConfig(image=(16, 24), patch=(8, 8), dim=32, depth=1, heads=2, dim_head=32, mlp_dim=64)
    p = {k: v.clone().requires_grad_("to_patch_embedding" not in k) for k, v in s0.items()}
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
    assert g0["trans.to_patch_embedding.1.weight"] is None and max(v.numel() for v in p.values()) > 1024    # frozen; buckets split parameters


def test_reduce_op_follows_the_device_backend(monkeypatch):
    """RCCL averages inside the collective (no 1/world pass); gloo can only sum.  A process group with one backend per device type
    reports "cpu:gloo,cuda:nccl": device gradients take AVG, host gradients SUM + scale -- AVG and "no scale" always go together."""
    import sys
    sys.path.insert(0, ROOT)
    import dgvit_amd  # noqa: F401
    from dgvit_amd.parallel import GradSync
    gs = GradSync([torch.nn.Linear(2, 2)])
    for backend, on_device, want in [("nccl", True, (dist.ReduceOp.AVG, False)), ("gloo", True, (dist.ReduceOp.SUM, True)),
                                     ("gloo", False, (dist.ReduceOp.SUM, True)), ("cpu:gloo,cuda:nccl", True, (dist.ReduceOp.AVG, False)),
                                     ("cpu:gloo,cuda:nccl", False, (dist.ReduceOp.SUM, True)), ("cuda:nccl", True, (dist.ReduceOp.AVG, False))]:
        monkeypatch.setattr(dist, "get_backend", lambda group=None, b=backend: b)
        assert gs._reduce_op(on_device) == want, (backend, on_device)


def test_ranks_draw_different_dropout_seeds_from_their_own_generator():
    """emb-dropout is live in training (GoalFormer.py:163): every rank seeds torch with base + rank (bench.py), GoT draws its Philox seed
    from that CPU generator -- the masks differ between ranks and repeat for a rank."""
    import sys
    sys.path.insert(0, ROOT)
    import dgvit_amd
    seeds = []
    for rank in range(4):
        torch.manual_seed(3407 + rank)
        a = dgvit_amd.GoT.draw_dropout_seed()
        torch.manual_seed(3407 + rank)
        assert dgvit_amd.GoT.draw_dropout_seed() == a
        seeds.append(a)
    assert len(set(seeds)) == 4
