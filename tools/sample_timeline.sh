# ordered dispatches of the LAST of 20 eager single-frame (or small-batch) policy.sample() calls on the product library
# bash tools/sample_timeline.sh [B]
B=${1:-1}
mkdir -p gpurun_out/r4s && export TMPDIR=/tmp
O=gpurun_out/r4s
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 tools/small_batch_ab.py trace $B > /dev/null 2> $O/t.err
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:110]))
for f in glob.glob("$O/t/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# one pass = from one patch GEMM / first kernel to the tanh_gaussian kernel: take the last 2 passes
idx = [i for i, r in enumerate(rows) if "tanh_gaussian" in r[2]]
lo = idx[-3] + 1 if len(idx) >= 3 else 0
prev = None
for s, e, n in rows[lo:]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - rows[lo][0]) / 1e3:9.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:6.1f}  {n}")
    prev = e
PY
rm -rf $O/t
