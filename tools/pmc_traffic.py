#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel family.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [out.json]

HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, and on
gfx950 FETCH_SIZE reports exactly half of the bytes of a 16-B-per-lane coalesced stream
(MI355X_MICROARCH.md, HBM section), which is how every operand here is read (buffer_load_dwordx4 / float4).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthetic import kernel_source_digest  # noqa: E402


def family(name):
    for key in ("gemm_f32_kernel", "gemm_bf16_stream_kernel", "gemm_bf16_ring_kernel", "attn_fwd_bf16_stream", "attn_fwd_pipe_kernel", "attn_fwd_kernel", "attn_bwd_kernel", "attn_bwd64_kernel", "attn_q1_fwd_kernel", "attn_q1_bwd_kernel",
                "reduce_slabs", "layernorm_fwd", "layernorm_bwd"):
        if key in name:
            return key
    return None


def load(d, counter):
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        if fam:
            acc[fam][0] += float(r["Counter_Value"])
            acc[fam][1] += 1
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for fam in fetch:
        nf, nw = fetch[fam][1], write[fam][1]
        rd = 2.0 * fetch[fam][0] * 1024 / max(nf, 1)
        wr = write[fam][0] * 1024 / max(nw, 1)
        out[fam] = {"launches_profiled": nf, "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
                    "hbm_bytes_per_launch": round(rd + wr)}
        print(f"{fam:18s} launches={nf:5d}  read {rd/1e6:9.2f} MB  write {wr/1e6:9.2f} MB  total {(rd+wr)/1e6:9.2f} MB per launch")
    if len(sys.argv) > 3:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --steps 3 --warmup 1; "
                             "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 read correction)",
                   # bench.py quotes these bytes only while the GEMM kernel sources they were measured on are unchanged (digest of
                   # csrc/gemm*.hip + headers); tools/stamp_profile.py adds the commit when the file is copied into profiles/
                   "kernel_sources_sha256": kernel_source_digest(), "kernels": out},
                  open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
