"""Can work queued on a second stream behind a gradient-ready event run BESIDE the backward?  (it can only if that stream has a
hardware queue of its own)

    [NODIST=1] [GPU_MAX_HW_QUEUES=n] python tools/overlap_queue_probe.py wait|kernel|allreduce

The C3 model's backward (B = 512) records one event per transformer block (dgvit_got_backward_ev); the hook queues, on a side stream
behind each event: nothing (`wait`), a small elementwise kernel (`kernel`), or a one-rank RCCL all-reduce (`allreduce`, launches no
kernel), then a timing marker.  Printed: when each marker fired relative to the start of the backward.
Measured on MI355X / ROCm 7: markers alone fire when their block is final (0.8, 2.4, 4.0 ... ms of a 9 ms backward).  A KERNEL behind
the event does too while RCCL is not initialised; once a process group exists, the HIP runtime's default of 4 hardware queues puts the
side stream on the compute stream's queue and every kernel waits for the whole backward (all markers at ~8.9 ms); with
GPU_MAX_HW_QUEUES=8 they are back on time.  bench.py sets that variable; an integration that wants the overlap must too.
"""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgvit_amd
from dgvit_amd import _lib
import synthetic
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
if os.environ.get("NODIST") is None: dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
MODE = sys.argv[1] if len(sys.argv) > 1 else "wait"
lib = dgvit_amd.load_library()
dev = torch.device("cuda", 0)
B = 512
torch.manual_seed(3407)
model = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)).to(dev).train()
img, pstate, _, _ = (t.to(dev) for t in synthetic.make_inputs((84, 84), B, 3407))
tgt = torch.randn(B, 2, device=dev)
NS = int(os.environ.get("NSIDE", "1"))
for _ in range(NS): side = torch.cuda.Stream(device=dev)
print("side stream", side)
marks = []
handles = []
host = {}
def hook(flat, ranges, events):
    host["hook"] = time.perf_counter()
    for r, e in zip(ranges, events):
        _lib.check(lib.dgvit_stream_wait_event(side.cuda_stream, e), "wait")
        if MODE == "allreduce":
            with torch.cuda.stream(side):
                handles.append(dist.all_reduce(flat[r[0]:r[1]], op=dist.ReduceOp.AVG, async_op=True))
        elif MODE == "kernel":
            with torch.cuda.stream(side):
                flat[r[0]:r[1]].mul_(1.0)
        ev = torch.cuda.Event(enable_timing=True); ev.record(side); marks.append(ev)
for sub in model.modules():
    if hasattr(sub, "_grad_hook"):
        sub._grad_hook = hook
for it in range(5):
    for p in model.parameters(): p.grad = None
    marks.clear()
    mean, log_std = model([img, pstate])
    loss = ((mean - tgt) ** 2).mean() + (log_std ** 2).mean()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter(); t0.record()
    loss.backward()
    t1.record(); h1 = time.perf_counter()
    for h in handles: h.wait()
    handles.clear()
    torch.cuda.synchronize(); h2 = time.perf_counter()
    if it >= 3:
        print(f"host: hook entered {1e3*(host['hook']-h0):.3f} ms after backward() was called, backward() returned at {1e3*(h1-h0):.3f} ms, device idle at {1e3*(h2-h0):.3f} ms; device backward {t0.elapsed_time(t1):.3f} ms")
        print("  releases after t0 (ms):", [round(t0.elapsed_time(m), 3) for m in marks])

if os.environ.get('NODIST') is None: dist.destroy_process_group()
