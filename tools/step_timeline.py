#!/usr/bin/env python3
"""In-situ timeline of ONE training step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 2 ...
    python tools/step_timeline.py gpurun_out/trace [step_index_from_end=1] [> profiles/rNN_step_timeline.txt]

Prints every dispatch of the chosen step in order: start offset, duration, gap to the previous kernel's end, kernel name
(shortened) and grid; then totals per kernel family, the sum of gaps, and the step's wall span.  A step is delimited by the
adam_kernel launches (the last kernel of a step).
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"gemm_f32_kernel<TileCfg<(\d+), (\d+), (\d+), \d+, \d+>, (\d), (\d), (\d)>", name)
    if m:
        lay = {"0": "NT", "1": "NN", "2": "TN"}[m.group(4)]
        return f"gemm_f32 {lay} {m.group(1)}x{m.group(2)}x{m.group(3)} v{m.group(5)} epi{m.group(6)}"
    name = re.sub(r"at::native::", "", name)
    return name.split("(")[0][:70]


def family(s):
    for k in ("gemm_f32", "gemm_bf16", "attn_fwd", "attn_bwd", "layernorm_fwd", "layernorm_bwd", "reduce_slabs", "reduce_group", "adam", "copyBuffer",
              "fillBuffer", "im2col", "col2im", "small_", "elementwise", "reduce_kernel"):
        if k in s:
            return k
    return "other"


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    marker = sys.argv[3] if len(sys.argv) > 3 else "adam_kernel"     # the kernel that ends a step
    f = (glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")), r.get("Workgroup_Size_X", "?")))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if marker in r[2]]
    # a step ends at the LAST adam launch of a cluster of adam launches (encoder run, heads run: a few dispatches apart)
    step_ends = [i for j, i in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - i > 6]
    hi = step_ends[-back]
    lo = step_ends[-back - 1] + 1
    step = rows[lo:hi + 1]
    t0 = step[0][0]
    fam_t, fam_n = defaultdict(float), defaultdict(int)
    gaps = 0.0
    prev_end = None
    print(f"# step of {len(step)} dispatches; columns: start_us  dur_us  gap_us  kernel  grid/wg")
    for s, e, name, grid, wg in step:
        sn = short(name)
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        gaps += max(gap, 0.0)
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {sn:60s} {grid}/{wg}")
        fam_t[family(sn)] += (e - s) / 1e3
        fam_n[family(sn)] += 1
        prev_end = e if prev_end is None else max(prev_end, e)
    span = (step[-1][1] - t0) / 1e3
    print(f"# span {span:.1f} us, sum of kernel durations {sum(fam_t.values()):.1f} us, sum of positive gaps {gaps:.1f} us")
    for k in sorted(fam_t, key=lambda k: -fam_t[k]):
        print(f"# {k:16s} {fam_n[k]:4d} launches {fam_t[k]:9.1f} us")


if __name__ == "__main__":
    main()
