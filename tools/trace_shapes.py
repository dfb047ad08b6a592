"""Per-shape device time from a rocprofv3 kernel trace of tools/run_shapes.py.
    python tools/trace_shapes.py <kernel_trace.csv> REPS M,N,K [M,N,K ...]"""
import csv
import sys

OURS = ("gemm_skinny", "gemm_stream", "gemv_kernel", "gemm_mfma", "gemm_generic", "gemm_rows", "splitk_reduce", "add_rows")
MARK = "FillFunctor<int>"   # tools/run_shapes.py launches one int32 fill in front of every call
rows = [r for r in csv.DictReader(open(sys.argv[1])) if MARK in r["Kernel_Name"] or any(t in r["Kernel_Name"] for t in OURS)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
reps = int(sys.argv[2])
calls = []  # each call: list of (name, dur): the product's kernel(s), then a split-K reduce if there is one
for r in rows:
    if MARK in r["Kernel_Name"]:
        calls.append([])
        continue
    if not calls:
        continue   # (kernels before the first marker: none in run_shapes.py)
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].split("(")[0]
    calls[-1].append(("reduce" if "splitk_reduce" in name else name[:60], d))
calls = [c for c in calls if c]
i = 0
for spec in sys.argv[3:]:
    m, n, k = (int(a) for a in spec.split(","))
    mine = calls[i:i + reps]
    i += reps
    tot = sorted(sum(d for _, d in c) for c in mine)
    med = tot[len(tot) // 2] / 1e3
    main = sorted(c[0][1] for c in mine)[len(mine) // 2] / 1e3
    byts = m * k * 2 + n * k // 4 + 8 * n + m * n * 2
    kern = "rows" if "gemm_rows" in mine[0][0][0] else "gemv" if "gemv" in mine[0][0][0] else "stream" if "stream" in mine[0][0][0] else "skinny" if "skinny" in mine[0][0][0] else ("mfma" if "mfma" in mine[0][0][0] else "generic")
    print(f"{spec:>18s}  {kern:7s} kernels/call {len(mine[0])}  median total {med:8.2f} us (main {main:8.2f})  "
          f"{2.0 * m * n * k / med / 1e6:9.1f} TFLOP/s  {byts / med / 1e3:8.1f} GB/s ({byts / med / 1e3 / 80:5.1f} % of 8 TB/s)")
