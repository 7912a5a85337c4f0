"""GPU: the batch-sharded training path with the HIP kernels in it.  Two processes share the one card of the test box (gloo carries the
gradients; RCCL needs one GPU per rank and is what bench.py uses on a node): each rank runs the HIP forward/backward on its half
of the frames, GradSync all-reduces, and the averaged gradients must equal those of one process on the whole batch."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import O  # noqa: E402

pytestmark = pytest.mark.gpu

B = 8


def _cfg():
    return O.GoTConfig(image=(84, 84), patch=(12, 12), dim=64, depth=2, heads=2)


def _model_and_data(bf16=False):
    import dgvit_amd
    cfg = _cfg()
    params = O.make_params(O.policy_param_spec(cfg), 55)
    m = dgvit_amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, image_size=cfg.image, patch_size=cfg.patch)
    m.load_state_dict(params, strict=True)
    m = m.to("cuda").eval()
    if bf16:
        m.trans.set_compute_dtype(torch.bfloat16)
    img, pstate, _, _ = O.make_inputs(cfg, B, 55)
    g = torch.Generator().manual_seed(55)
    tgt = torch.randn(B, 2, generator=g)
    return m, img.cuda(), pstate.cuda(), tgt.cuda()


def _loss(m, img, pstate, tgt):
    mean, log_std = m([img, pstate])
    return ((mean - tgt) ** 2).mean() + (log_std ** 2).mean()


def _worker(rank, world, tmp, overlap, bf16):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", init_method=f"file://{tmp}/rdzv", rank=rank, world_size=world)
    try:
        from dgvit_amd.parallel import GradSync
        m, img, pstate, tgt = _model_and_data(bf16)
        per = B // world
        sl = slice(rank * per, (rank + 1) * per)
        sync = GradSync([m], overlap=overlap, bucket_bytes=64 << 10)
        for _ in range(2):      # twice: the events and the side stream are re-used by the second step
            sync.zero_grad()
            _loss(m, img[sl], pstate[sl], tgt[sl]).backward()
            launched = sync.early_launches
            sync.sync()
        torch.cuda.synchronize()
        # overlap: both transformer blocks' all-reduces (64 KB buckets, 1.2 MB of gradients per block) were queued from inside the backward
        assert (launched >= 2 * 2 * 18 and launched % 4 == 0) if overlap else launched == 0, launched
        if rank == 0:
            torch.save({k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}, os.path.join(tmp, "grads.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,bf16", [(False, False), (True, False), (True, True)], ids=["after-backward", "overlapped", "overlapped-bf16"])
def test_two_ranks_on_the_hip_path_match_one_process(overlap, bf16):
    """overlapped: GradSync(overlap=True) starts each transformer block's all-reduce from inside the backward, behind the
    gradient-ready event the C ABI records (dgvit_got_backward[_bf16]_ev); the result must not change."""
    m, img, pstate, tgt = _model_and_data(bf16)
    _loss(m, img, pstate, tgt).backward()
    torch.cuda.synchronize()
    want = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, tmp, overlap, bf16), nprocs=2, join=True)
        got = torch.load(os.path.join(tmp, "grads.pt"), weights_only=True)
    assert sorted(got) == sorted(want)
    for k in want:
        # bf16: the two half-batches round their activations separately from the whole batch
        tol = (2e-2 if bf16 else 2e-5) * float(want[k].abs().max()) + 1e-9
        assert float((got[k] - want[k]).abs().max()) <= tol, k


def _worker_sac_pattern(rank, world, tmp):
    """(a) a backward through the encoder that is NOT followed by sync() -- DRL.py:407-413 back-propagates the policy loss through the
    critic -- between two synced steps of GradSync(overlap=True); (b) ranks start from different weights and different torch seeds:
    after broadcast_parameters the weights agree, the train-mode dropout masks do not."""
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", init_method=f"file://{tmp}/rdzv", rank=rank, world_size=world)
    try:
        from dgvit_amd.parallel import GradSync
        m, img, pstate, tgt = _model_and_data(False)
        if rank:                                   # a rank-local perturbation that the broadcast must undo (frozen parameters included)
            with torch.no_grad():
                for q in m.parameters():
                    q.add_(0.01 * rank)
        m.trans.to_patch_embedding[1].requires_grad_(False)
        per = B // world
        sl = slice(rank * per, (rank + 1) * per)
        sync = GradSync([m], overlap=True, bucket_bytes=64 << 10)
        sync.broadcast_parameters(0)
        flat = torch.cat([q.detach().reshape(-1) for q in m.parameters()]).cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert all(torch.equal(both[0], b) for b in both), "weights differ after the broadcast"
        # train mode: each rank seeds torch with base + rank (bench.py) -> different Philox masks on the SAME frames
        torch.manual_seed(3407 + rank)
        m.train()
        with torch.no_grad():
            f = m.trans(img[:2], torch.zeros(2, 64, device="cuda")).cpu()
        feats = [torch.empty_like(f) for _ in range(world)]
        dist.all_gather(feats, f)
        assert float((feats[0] - feats[1]).abs().max()) > 1e-3, "ranks drew the same dropout mask"
        m.eval()
        for step in range(2):
            sync.zero_grad()
            _loss(m, img[sl], pstate[sl], tgt[sl]).backward()
            launched = sync.early_launches
            sync.sync()
            kept = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
            # the stray backward: same encoder, no zero_grad / sync around it; it accumulates into .grad like any second backward
            _loss(m, img[sl], pstate[sl], tgt[sl]).backward()
            assert sync.early_launches == launched and not sync._early, "a backward outside zero_grad()..sync() started collectives"
        torch.cuda.synchronize()
        assert launched > 0
        if rank == 0:
            torch.save({k: v.cpu() for k, v in kept.items()}, os.path.join(tmp, "grads.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_unsynced_backward_between_overlapped_steps_and_rank_local_dropout():
    m, img, pstate, tgt = _model_and_data(False)
    m.trans.to_patch_embedding[1].requires_grad_(False)
    _loss(m, img, pstate, tgt).backward()
    torch.cuda.synchronize()
    want = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    assert not any("to_patch_embedding" in k for k in want)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_sac_pattern, args=(2, tmp), nprocs=2, join=True)
        got = torch.load(os.path.join(tmp, "grads.pt"), weights_only=True)
    assert sorted(got) == sorted(want)
    for k in want:
        assert float((got[k] - want[k]).abs().max()) <= 2e-5 * float(want[k].abs().max()) + 1e-9, k
