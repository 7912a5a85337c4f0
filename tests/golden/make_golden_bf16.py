#!/usr/bin/env python3
"""Golden vectors for the bf16 configuration (BASELINE config 5), FROM THE REFERENCE ITSELF.

Run only in the build container (needs /root/reference):

    python tests/golden/make_golden_bf16.py

The reference's GoT (unchanged code; patch embedding resized exactly as make_golden.py does) is run on CPU twice on
the same seeded weights and inputs: in fp32, and under ``torch.autocast('cpu', dtype=torch.bfloat16)`` -- the
standard way a user of the reference would run it in bf16.  Stored: both feature tensors.  The autocast output is a
bf16 *peer* of our HIP bf16 path, not an exact target (rounding points differ: autocast keeps bf16 GELU inputs and a
bf16 softmax input); tests bound our distance to it and to the fp32 output.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_golden as G            # noqa: E402  (imports the reference)
from oracle import dgvit_oracle as O   # noqa: E402


def case(name, cfg, batch, seed):
    m = G.build_got(cfg)
    G.load(m, O.got_param_spec(cfg, prefix=""), seed)
    m.eval()
    img, _, _, _ = O.make_inputs(cfg, batch, seed)
    rs = np.random.RandomState(seed + 7)
    goal = torch.from_numpy(rs.standard_normal((batch, cfg.dim))).float()
    out = G.cfg_meta(cfg, batch, seed)
    with torch.no_grad():
        out["feat_fp32"] = m(img, goal).numpy()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out["feat_autocast_bf16"] = m(img, goal).float().numpy()
    d = np.abs(out["feat_fp32"] - out["feat_autocast_bf16"])
    print(f"{name}: |fp32 - autocast| max {d.max():.4f} mean {d.mean():.5f}")
    G.save(name, out)


if __name__ == "__main__":
    # config 5 shape (224x224 @ 16, D768 H12 M3072) cut to depth 2 for fixture size / CPU time, and at full depth 12
    case("got_c5_l2_bf16", O.GoTConfig(image=(224, 224), patch=(16, 16), dim=768, depth=2, heads=12, dim_head=64, mlp_dim=3072), 2, 11)
    case("got_c5_l12_bf16", O.GoTConfig(image=(224, 224), patch=(16, 16), dim=768, depth=12, heads=12, dim_head=64, mlp_dim=3072), 2, 12)
    # DGViT-small at 84x84 @ 12 (config 2/3 shape) in bf16
    case("got_84p12_bf16", O.GoTConfig(image=(84, 84), patch=(12, 12), dim=256, depth=6, heads=8, dim_head=64, mlp_dim=2048), 4, 13)
