#!/usr/bin/env python3
"""Can policy.sample() for one frame be captured into a HIP graph and replayed?  Latency eager vs graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dgvit_amd
import synthetic


class C:   # shape record with the FLOP model (tools must not import oracle/)
    def __init__(self, image=(128, 160), patch=(16, 20), dim=64, depth=4, heads=4, mlp_dim=2048):
        self.image, self.patch, self.dim, self.depth, self.heads, self.mlp_dim = image, patch, dim, depth, heads, mlp_dim

    def fwd_flops_per_frame(self):
        return synthetic.fwd_flops_per_frame(self.image, self.patch, self.dim, self.depth, self.heads, 64, self.mlp_dim)


class O:   # minimal stand-in namespace used below
    GoTConfig = C

    @staticmethod
    def make_inputs(cfg, batch, seed):
        return synthetic.make_inputs(cfg.image, batch, seed)
dev = "cuda"
cfg = O.GoTConfig(dim=64, depth=4, heads=4)
torch.manual_seed(0)
m = dgvit_amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim).to(dev).eval()
img, ps, _, _ = (t.to(dev) for t in O.make_inputs(cfg, 1, 0))
simg, sps = img.clone(), ps.clone()
with torch.no_grad():
    for _ in range(5):
        m.sample([simg, sps])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            m.sample([simg, sps])
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = m.sample([simg, sps])
    torch.cuda.synchronize()
    eager_mean = m.sample([simg, sps])[2]
    g.replay(); torch.cuda.synchronize()
    print("graph mean == eager mean:", torch.allclose(out[2], eager_mean, atol=1e-6), out[2].tolist())
    a1 = out[0].clone(); g.replay(); torch.cuda.synchronize(); a2 = out[0].clone()
    print("sampled actions differ between replays (RNG advances):", not torch.equal(a1, a2))
    def t(fn, n=300):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print("eager  us/frame:", round(t(lambda: m.sample([simg, sps])), 1))
    print("graph  us/frame:", round(t(lambda: g.replay()), 1))
    def full():
        simg.copy_(img); sps.copy_(ps); g.replay(); return out[0].cpu()
    print("graph + H2D-style copies + D2H action us/frame:", round(t(full), 1))
