#!/usr/bin/env python3
"""Raw per-kernel-family sums of whatever counters a rocprofv3 --pmc pass collected (per launch), for one-off questions
(LDS bank conflicts, wait reasons): python tools/pmc_raw.py <dir> [substring of the kernel name ...]"""
import collections, csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
keys = sys.argv[2:] or ["gemm_f32_kernel"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    for k in keys:
        if k in r["Kernel_Name"]:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
for k in keys:
    print(k, "launches", len(n[k]))
    for c, v in sorted(acc[k].items()):
        print(f"   {c:34s} {v / max(1, len(n[k])):16.1f} per launch")
