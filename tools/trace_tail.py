"""Average duration of the last N dispatches of a kernel in a rocprofv3 kernel trace:
    python tools/trace_tail.py <kernel_trace.csv> <kernel name substring> N"""
import csv
import sys

path, needle, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        if needle in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
tail = [d for _, d in rows[-n:]]
print(f"kernel *{needle}*: {len(rows)} dispatches in the trace; last {len(tail)}: average {sum(tail)/len(tail)/1e3:.2f} us, "
      f"min {min(tail)/1e3:.2f} us, max {max(tail)/1e3:.2f} us")
