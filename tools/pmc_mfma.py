#!/usr/bin/env python3
"""MFMA utilisation per kernel family from one rocprofv3 --pmc pass (SQ + GRBM counters, no trace domains besides the kernel trace).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_BF16 \
              SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
              --kernel-trace -d gpurun_out/pmc_mfma -o x --output-format csv -- python3 bench.py ...
    python tools/pmc_mfma.py gpurun_out/pmc_mfma [out.json]

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles the kernel is on the chip), cycles taken two ways: GRBM_GUI_ACTIVE / 8 (the
counter is summed over the 8 XCDs; it ticks at ~2.3-2.4 GHz whatever the kernel) and launch duration x the 2.4 GHz of the peak figures
(this one equals achieved / peak FLOP/s when no MFMA work is wasted).  Cross-check: busy cycles per MFMA instruction (64 for v_mfma_f32_32x32x2_f32, 32 for
v_mfma_f32_32x32x16_bf16, 16 for v_mfma_f32_16x16x32_bf16; MI355X_MICROARCH.md, cycle table) is printed next to it, and the FLOP/s
implied by MOPS (512 FLOP each) over the profiled kernel time.
"""
import collections
import csv
import glob
import json
import sys

FAMILIES = ("gemm_f32_kernel", "gemm_bf16_stream_kernel", "gemm_bf16_ring_kernel", "gemm_bf16_kernel", "attn_fwd_pipe_kernel", "attn_fwd_kernel", "attn_bwd_kernel", "attn_bwd64_kernel",
            "attn_fwd_bf16_stream", "attn_fwd_bf16",
            "attn_bwd_dq_bf16", "attn_bwd_dkv_bf16")
SIMDS = 256 * 4


def family(name):
    for key in FAMILIES:
        if key in name:
            return key
    return None


def main():
    f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if not fam:
            continue
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[fam]:
            seen[fam].add(r["Dispatch_Id"])
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[fam] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for fam, c in acc.items():
        n = len(seen[fam])
        insts = c.get("SQ_INSTS_VALU_MFMA_F32", 0.0) + c.get("SQ_INSTS_VALU_MFMA_BF16", 0.0)
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        row = {"launches_profiled": n, "mfma_instructions_per_launch": round(insts / max(n, 1)),
               "mfma_busy_cycles_per_instruction": round(busy / insts, 2) if insts else None,
               "mfma_utilisation_gui_cycles": round(busy / (gui / 8 * SIMDS), 4) if gui else None,
               "mfma_utilisation_at_2p4GHz": round(busy / (SIMDS * dur[fam] * 2.4), 4) if dur[fam] else None,
               "gui_clock_GHz": round(gui / 8 / dur[fam], 3) if dur[fam] and gui else None,
               "avg_launch_us_under_pmc": round(dur[fam] / max(n, 1) / 1e3, 1) if dur[fam] else None,
               "tflops_from_mops": round(mops * 512 / dur[fam] / 1e3, 1) if dur[fam] and mops else None,
               "wait_inst_share_of_wave_cycles": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3) if c.get("SQ_WAVE_CYCLES") else None,
               "raw": {k: v for k, v in c.items()}}
        out[fam] = row
        print(fam, {k: v for k, v in row.items() if k != "raw"})
    if len(sys.argv) > 2:
        json.dump({"source": "one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_*, GRBM_GUI_ACTIVE, ...); "
                             "utilisation = MFMA busy cycles / (1024 SIMDs x kernel cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs, or duration x 2.4 GHz", "kernels": out},
                  open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
