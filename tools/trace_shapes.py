"""Per-shape device time from a rocprofv3 kernel trace of tools/run_shapes.py.
    python tools/trace_shapes.py <kernel_trace.csv> REPS M,N,K [M,N,K ...]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(t in r["Kernel_Name"] for t in ("gemm_skinny", "gemm_stream", "gemv_kernel", "gemm_mfma", "gemm_generic", "gemm_rows", "splitk_reduce"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
reps = int(sys.argv[2])
calls = []  # each call: list of (name, dur)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if "splitk_reduce" in r["Kernel_Name"] and calls:
        calls[-1].append(("reduce", d))
    else:
        calls.append([(r["Kernel_Name"].split("(")[0][:60], d)])
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
