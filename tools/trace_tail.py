"""Average duration of N dispatches of a kernel in a rocprofv3 kernel trace: the last N, or with SKIP given the N that follow the
kernel's first SKIP dispatches (the bench's timed region = the --steps launches after its --warmup launches):
    python tools/trace_tail.py <kernel_trace.csv> <kernel name substring> N [SKIP]"""
import csv
import sys

path, needle, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
skip = int(sys.argv[4]) if len(sys.argv) > 4 else None
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        if needle in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
sel = rows[-n:] if skip is None else rows[skip:skip + n]
tail = [d for _, d in sel]
which = f"last {len(tail)}" if skip is None else f"dispatches {skip} .. {skip + len(tail) - 1}"
span = (sel[-1][0] + sel[-1][1] - sel[0][0]) / len(sel)
print(f"kernel *{needle}*: {len(rows)} dispatches in the trace; {which}: average {sum(tail)/len(tail)/1e3:.2f} us, "
      f"min {min(tail)/1e3:.2f} us, max {max(tail)/1e3:.2f} us; first start to last end / {len(sel)} = {span/1e3:.2f} us")
