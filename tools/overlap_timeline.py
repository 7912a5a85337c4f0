"""Does the gradient exchange start before the backward has finished?  

    GPU_MAX_HW_QUEUES=8 python tools/overlap_timeline.py [batch]

One-rank RCCL group on this GPU (a one-rank all-reduce is the identity, but it is queued, ordered and waited for exactly as on 8
ranks), BASELINE config 3 model, GradSync(overlap=True).  Device-side evidence from HIP events:
  * `reduced[k]`: recorded on GradSync's side stream right after the all-reduce of the k-th finished transformer block was queued there
    (the stream had to wait for that block's gradient-ready event first);
  * `bwd_end`: recorded on the compute stream when loss.backward() returned (= behind the last backward kernel).
A block's exchange overlaps the backward iff reduced[k] happens BEFORE bwd_end on the device clock.
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd  # noqa: E402
from dgvit_amd.parallel import GradSync  # noqa: E402
import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(3407)
model = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)).to(dev).train()
sync = GradSync([model], force_collective=True, overlap=True)
img, pstate, _, _ = (t.to(dev) for t in synthetic.make_inputs((84, 84), B, 3407))
tgt = torch.randn(B, 2, device=dev)

marks = []


def mark(side):                            # behind each block's queued all-reduces, on GradSync's side stream
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(side)
    marks.append(ev)


sync.on_block_queued = mark


def step(timed):
    sync.zero_grad()
    marks.clear()
    mean, log_std = model([img, pstate])
    loss = ((mean - tgt) ** 2).mean() + (log_std ** 2).mean()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    loss.backward()
    t1.record()
    n_early = len(marks)
    sync.sync()
    torch.cuda.synchronize()
    if timed:
        print(f"backward on the device: {t0.elapsed_time(t1):.3f} ms; {n_early} all-reduce groups were queued before loss.backward() returned")
        for k, ev in enumerate(marks):
            print(f"  block {5 - k}: gradients final and exchange released {t0.elapsed_time(ev):7.3f} ms after the backward began, "
                  f"{ev.elapsed_time(t1):7.3f} ms before its last kernel finished")


for i in range(6):
    step(i >= 4)
dist.destroy_process_group()
