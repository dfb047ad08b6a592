"""Reads the output of tools/planner_audit.sh and prints, per shape, the library's own plan time, the best pinned plan and the gap."""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
t = collections.OrderedDict()
for r in rows:
    t.setdefault((r["m"], r["n"], r["k"]), {})[r["plan"]] = (r["us"], r["exact"])
worst = []
for k, v in t.items():
    auto = v["auto"][0]
    best_p, (best_u, _) = min(((p, u) for p, u in v.items() if p != "auto" and u[1]), key=lambda a: a[1][0])
    gap = auto / best_u - 1.0
    bad = [p for p, u in v.items() if not u[1]]
    worst.append((gap, k, auto, best_p, best_u, bad))
for gap, k, auto, bp, bu, bad in sorted(worst, key=lambda a: (a[1][1], a[1][2], a[1][0])):
    flag = " <<<" if gap > 0.05 else ""
    print(f"{k[0]:6d} x {k[1]:6d} x {k[2]:5d}: auto {auto:8.2f}   best pinned {bp:12s} {bu:8.2f}   auto is {100 * gap:+6.1f} %{flag}{'  WRONG: ' + str(bad) if bad else ''}")
