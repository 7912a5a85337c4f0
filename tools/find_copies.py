"""Which host-side ops issue device-to-device copies in one C3 training step (torch.profiler, with stacks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import dgvit_amd
import synthetic
from dgvit_amd.parallel import GradSync
from dgvit_amd.optim import FlatAdam
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = dgvit_amd.GoTPolicy(2, 2, 6, 8, 256, image_size=(84, 84), patch_size=(12, 12)).to(dev).train()
sync = GradSync([model]); opt = FlatAdam([model], lr=1e-4)
img, pstate, _, _ = (t.to(dev) for t in synthetic.make_inputs((84, 84), 512, 1))
tm, tl = torch.randn(512, 2, device=dev), torch.randn(512, 2, device=dev)

def step():
    sync.zero_grad()
    mean, log_std = model([img, pstate])
    loss = ((mean - tm) ** 2).mean() + ((log_std - tl) ** 2).mean()
    loss.backward(); sync.sync(); opt.step()

for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
import collections
c = collections.Counter()
for e in prof.events():
    n = e.name
    if "emcpy" in n or "copyBuffer" in n or "emset" in n or "fillBuffer" in n:
        par = e.cpu_parent.name if getattr(e, "cpu_parent", None) is not None else None
        gp = e.cpu_parent.cpu_parent.name if par and e.cpu_parent.cpu_parent is not None else None
        c[(n[:50], str(e.device_type)[-4:], par, gp)] += 1
for k, v in c.most_common(40):
    print(v, k)
