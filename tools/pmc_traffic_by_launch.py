#!/usr/bin/env python3
"""HBM bytes per launch of the fp32 GEMM, broken down by instantiation and grid (= by GEMM shape of the training step).

    python tools/pmc_traffic_by_launch.py gpurun_out/pf gpurun_out/pw   (the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh)
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 correction, tools/pmc_traffic.py)."""
import collections
import csv
import glob
import re
import sys


def load(d, counter):
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "gemm_f32_kernel" not in r["Kernel_Name"]:
            continue
        m = re.search(r"TileCfg<(\d+), (\d+), (\d+), \d+, \d+>, (\d), (\d), (\d+)", r["Kernel_Name"])
        key = (m.group(0) if m else r["Kernel_Name"][:60], r.get("Grid_Size", ""), r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1] += 1
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    nf, nw = fetch[k][1], write.get(k, [0, 1])[1]
    rd = 2.0 * fetch[k][0] * 1024 / max(nf, 1)
    wr = write.get(k, [0.0, 1])[0] * 1024 / max(nw, 1)
    rows.append((nf * (rd + wr), k, nf, rd, wr))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{'instantiation (tile, layout, vec, epilogue)':48s} {'grid':>22s} {'launches':>8s} {'read MB':>9s} {'write MB':>9s} {'share':>6s}")
for t, k, n, rd, wr in rows:
    grid = "x".join(v for v in k[2:] if v) or k[1]
    print(f"{k[0]:48s} {grid:>22s} {n:8d} {rd / 1e6:9.2f} {wr / 1e6:9.2f} {t / tot:6.3f}")
