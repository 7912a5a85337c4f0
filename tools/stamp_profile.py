#!/usr/bin/env python3
"""Copy a JSON summary from gpurun_out/ into profiles/ with the commit it was collected at.

    python tools/stamp_profile.py gpurun_out/r4x/hbm_traffic.json profiles/r04_x_hbm_traffic.json

bench.py quotes `roofline.traffic` from such files: the commit (and the kernel-source digest tools/pmc_traffic.py stored) say which
code the bytes were measured on.  Run it in the build container (the GPU box has no .git)."""
import json
import subprocess
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "dgvit-depth-goal-guided-vision-transformer-_amd/csrc"], capture_output=True, text=True).stdout.strip()
d["collected_at_commit"] = head + ("+uncommitted csrc changes" if dirty else "")
if "kernel_sources_sha256" not in d and not dirty:
    # a summary written before tools/pmc_traffic.py recorded the digest itself: the GPU box ran a snapshot of this very tree (no csrc
    # change since), so the digest of the tree is the digest of what was measured
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from synthetic import kernel_source_digest
    d["kernel_sources_sha256"] = kernel_source_digest()
json.dump(d, open(dst, "w"), indent=1)
print(dst, d["collected_at_commit"])
