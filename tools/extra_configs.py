#!/usr/bin/env python3
"""Throughput / latency of the other BASELINE.json configs through the product modules (GPU box only).
Prints one JSON line per config; kept under profiles/ for DESIGN.md."""
import json, os, sys, time
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


def timed(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def policy(cfg):
    torch.manual_seed(0)
    return dgvit_amd.GoTPolicy(2, 2, cfg.depth, cfg.heads, cfg.dim, image_size=cfg.image, patch_size=cfg.patch).to(dev)


def report(name, cfg, B, dt, passes):
    fl = cfg.fwd_flops_per_frame() * passes
    print(json.dumps({"config": name, "batch": B, "ms": round(dt * 1e3, 4), "frames_per_s": round(B / dt, 1),
                      "tflops_dense": round(B / dt * fl / 1e12, 2), "frac_of_f32_mfma_peak": round(B / dt * fl / 1e12 / 157.3, 4)}), flush=True)


# C1: single frame, shipped model, 128x160, sample() (what SAC.choose_action runs, DRL.py:170-185)
cfg = C(dim=64, depth=4, heads=4)
m = policy(cfg).eval()
img, ps, _, _ = (t.to(dev) for t in O.make_inputs(cfg, 1, 0))
with torch.no_grad():
    report("C1a single frame 128x160 shipped L4/H4/D64: policy.sample()", cfg, 1, timed(lambda: m.sample([img, ps]), 10, 200), 1)
cfg = C(image=(84, 84), patch=(12, 12), dim=64, depth=4, heads=4)
m = policy(cfg).eval()
img, ps, _, _ = (t.to(dev) for t in O.make_inputs(cfg, 1, 0))
with torch.no_grad():
    report("C1b single frame 84x84@12 shipped model: policy.sample()", cfg, 1, timed(lambda: m.sample([img, ps]), 10, 200), 1)
# C2: B=256 forward only, DGViT-small 84x84@12
cfg = C(image=(84, 84), patch=(12, 12), dim=256, depth=6, heads=8)
m = policy(cfg).eval()
img, ps, _, _ = (t.to(dev) for t in O.make_inputs(cfg, 256, 0))
with torch.no_grad():
    report("C2 B=256 fwd-only DGViT-small 84x84@12 (eval)", cfg, 256, timed(lambda: m([img, ps])), 1)
m.train()
with torch.no_grad():
    report("C2 B=256 fwd-only DGViT-small 84x84@12 (train-mode dropout)", cfg, 256, timed(lambda: m([img, ps])), 1)
img, ps, _, _ = (t.to(dev) for t in O.make_inputs(cfg, 512, 0))
with torch.no_grad():
    report("C3-fwd B=512 fwd-only DGViT-small 84x84@12", cfg, 512, timed(lambda: m([img, ps])), 1)
# C0b: native 128x160, DGViT-small
cfg = C(dim=256, depth=6, heads=8)
m = policy(cfg).train()
img, ps, _, _ = (t.to(dev) for t in O.make_inputs(cfg, 256, 0))
with torch.no_grad():
    report("C0b B=256 fwd-only DGViT-small native 128x160@16x20 (N=65)", cfg, 256, timed(lambda: m([img, ps])), 1)


def fb():
    m.zero_grad(set_to_none=True)
    a, b = m([img, ps])
    ((a ** 2).mean() + (b ** 2).mean()).backward()


report("C0b B=256 fwd+bwd DGViT-small native 128x160", cfg, 256, timed(fb), 3)
# C5 shape in fp32: 224x224@16, L12 H12 D768 M3072 (bf16 path not built yet)
cfg = C(image=(224, 224), patch=(16, 16), dim=768, depth=12, heads=12, mlp_dim=3072)
torch.manual_seed(0)
g = dgvit_amd.GoT(image_size=cfg.image, patch_size=cfg.patch, num_classes=2, dim=cfg.dim, depth=cfg.depth, heads=cfg.heads,
                  mlp_dim=cfg.mlp_dim).to(dev).train()
img = torch.rand(64, 224, 224, device=dev)
goal = torch.randn(64, 768, device=dev, requires_grad=True)
with torch.no_grad():
    report("C5-shape B=64 fwd-only ViT-Base 224x224@16 in fp32", cfg, 64, timed(lambda: g(img, goal), 2, 5), 1)


def fb5():
    g.zero_grad(set_to_none=True)
    g(img, goal).square().mean().backward()


report("C5-shape B=64 fwd+bwd ViT-Base 224x224@16 in fp32", cfg, 64, timed(fb5, 2, 5), 3)

# the shipped critic (CNN QNetwork) and the shipped training step (GoT actor L4/H4/D64 + CNN critic)
q = dgvit_amd.QNetwork(2, 2).to(dev)
Bq = 512
img, ps, act, tgt = (t.to(dev) for t in O.make_inputs(C(), Bq, 0))


def qfb():
    q.zero_grad(set_to_none=True)
    q1, q2 = q([img, ps, act])
    (torch.nn.functional.mse_loss(q1, tgt.expand_as(q1)) + torch.nn.functional.mse_loss(q2, tgt.expand_as(q2))).backward()


dt = timed(qfb)
conv_flops = 2.0 * (62 * 78 * 16 * 25 + 29 * 37 * 64 * 400 + 13 * 17 * 256 * 1600)
print(json.dumps({"config": "f1 CNN QNetwork fwd+bwd B=512 128x160", "ms": round(dt * 1e3, 3), "frames_per_s": round(Bq / dt, 1),
                  "tflops_dense": round(Bq / dt * 3 * conv_flops / 1e12, 2)}), flush=True)
with torch.no_grad():
    dt = timed(lambda: q([img, ps, act]))
print(json.dumps({"config": "f1 CNN QNetwork fwd-only B=512 128x160", "ms": round(dt * 1e3, 3), "frames_per_s": round(Bq / dt, 1),
                  "tflops_dense": round(Bq / dt * conv_flops / 1e12, 2)}), flush=True)
