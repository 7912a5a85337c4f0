#!/usr/bin/env python3
"""torch.profiler view of one single-frame policy.sample() (which host call issues which device op)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd, synthetic
from torch.profiler import profile, ProfilerActivity
m = dgvit_amd.GoTPolicy(2, 2, 4, 4, 64).to("cuda").eval()
img, ps, _, _ = (t.to("cuda") for t in synthetic.make_inputs((128, 160), 1, 0))
with torch.no_grad():
    for _ in range(5):
        m.sample([img, ps])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m.sample([img, ps])
        torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60, max_src_column_width=90))
