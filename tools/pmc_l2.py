#!/usr/bin/env python3
"""L2 hit rate per kernel family from one `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace` pass.

    python tools/pmc_l2.py <rocprof output dir> [out.json]
"""
import collections
import csv
import glob
import json
import sys

FAMILIES = ("gemm_bf16_stream_kernel<0", "gemm_bf16_stream_kernel<1", "gemm_bf16_stream_kernel", "gemm_bf16_ring_kernel", "attn_fwd_bf16_stream", "layernorm_fwd_bf16",
            "gemm_f32_kernel", "attn_fwd_kernel", "attn_bwd64_kernel")


def family(name):
    for key in FAMILIES:
        if key in name:
            return key
    return None


f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    fam = family(r["Kernel_Name"])
    if fam:
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[fam].add(r["Dispatch_Id"])
out = {}
for fam, c in acc.items():
    hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    n = len(seen[fam])
    out[fam] = {"launches_profiled": n, "l2_requests_per_launch": round((hit + miss) / max(n, 1)), "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
    print(fam, out[fam])
if len(sys.argv) > 2:
    json.dump({"source": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace (own pass)", "kernels": out}, open(sys.argv[2], "w"), indent=1)
