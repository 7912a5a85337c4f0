#!/usr/bin/env python3
"""Per-shape L2 hit rate and bytes beyond the L2s of the stream GEMM from three `rocprofv3 --pmc` passes of
`tools/bf16_walk_budget.py pmc <budget> <batch> <reps>` (TCC_HIT_sum TCC_MISS_sum | FETCH_SIZE | WRITE_SIZE; dispatches are grouped by
launch order: reps launches per shape, shapes in the tool's order).

    python tools/pmc_walk.py <dir hit/miss> <dir fetch> <dir write> <reps> <batch> [out.json]
"""
import collections
import csv
import glob
import json
import sys

SHAPES = ["qkv", "out", "fc1", "fc2"]


def per_dispatch(d, names):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "gemm_bf16_stream_kernel" not in r["Kernel_Name"] or r["Counter_Name"] not in names:
            continue
        acc.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    return [acc[k] for k in sorted(acc)]


hm, fe, wr = per_dispatch(sys.argv[1], ("TCC_HIT_sum", "TCC_MISS_sum")), per_dispatch(sys.argv[2], ("FETCH_SIZE",)), per_dispatch(sys.argv[3], ("WRITE_SIZE",))
reps, B = int(sys.argv[4]), int(sys.argv[5])
T, D, M = B * 197, 768, 3072
dims = {"qkv": (T, 3 * D, D), "out": (T, D, D), "fc1": (T, M, D), "fc2": (T, D, M)}
assert len(hm) == len(fe) == len(wr) == reps * len(SHAPES), (len(hm), len(fe), len(wr))
out = {}
for i, s in enumerate(SHAPES):
    sl = slice(i * reps + 1, (i + 1) * reps)      # the first launch of a shape also pays the cold caches: left out
    n = reps - 1
    hit = sum(d["TCC_HIT_sum"] for d in hm[sl]) / n
    miss = sum(d["TCC_MISS_sum"] for d in hm[sl]) / n
    rd = 2 * 1024 * sum(d["FETCH_SIZE"] for d in fe[sl]) / n       # gfx950: FETCH_SIZE reads half of a wide streaming read (MI355X_MICROARCH.md, HBM)
    wb = 1024 * sum(d["WRITE_SIZE"] for d in wr[sl]) / n
    m, nn, k = dims[s]
    alg = 2 * (m * k + nn * k + m * nn)
    out[s] = {"l2_hit_rate": round(hit / (hit + miss), 4), "read_bytes_beyond_l2": round(rd), "write_bytes": round(wb), "bytes_beyond_l2": round(rd + wb),
              "algorithmic_bytes": alg, "ratio": round((rd + wb) / alg, 3)}
    print(s, out[s])
if len(sys.argv) > 6:
    json.dump({"source": "rocprofv3 --pmc passes of tools/bf16_walk_budget.py pmc (TCC_HIT_sum TCC_MISS_sum | FETCH_SIZE | WRITE_SIZE), FETCH x2 (gfx950)",
               "batch": B, "shapes": out}, open(sys.argv[6], "w"), indent=1)
