"""The tile planner's cost model (gemm_mfma.hip: plan_gemm, without the tail split) restated in Python, to judge it - or a change to
it - against measured timings of pinned plans (tools/planner_audit.sh) before touching the library:

    python tools/planner_model.py [-v] audit1.jsonl [audit2.jsonl ...]

Prints the mean regret (time of the model's pick over the best pinned plan's time, minus one) and, with -v, every shape."""
import collections
import json
import sys

CUS = 256
BIG, MID, SMALL = 2, 1, 0
# kind, bm, bn, occupancy, fixed, light, full, shared   (TILE_COSTS of gemm_mfma.hip)
TILE_COSTS = [
    (BIG, 256, 256, 1, 14.0, 1.20, 1.40, 1.57),
    (MID, 128, 256, 1, 7.0, 0.62, 0.84, 0.84),
    (MID, 128, 128, 2, 6.0, 0.42, 0.50, 0.49),
    (MID, 64, 128, 4, 5.0, 0.28, 0.36, 0.30),
    (SMALL, 32, 128, 3, 3.5, 0.32, 0.32, 0.21),
]


def cdiv(a, b):
    return -(-a // b)


def plan(m, n, k):
    steps = k // 64
    best, best_t = None, 1e30
    for (kind, bm, bn, occ, fixed, light, full, shared) in TILE_COSTS:
        if bm > 32 and m <= bm // 2:
            continue
        if bm == 32 and m > 48 and not (m <= 160 and cdiv(m, 64) * cdiv(n, 128) * 8 < CUS):
            continue
        tiles = cdiv(m, bm) * cdiv(n, bn)
        for s in range(1, 65):
            if s > 1 and (steps // s < 2 or kind == BIG):
                break
            per = cdiv(steps, s)
            q = cdiv(tiles * s, CUS)
            load = tiles * s / CUS
            fill = min(1.0, max(0.0, (load - 0.5) / 0.5))
            step1 = light + (full - light) * fill
            fixed1 = fixed - 4.0 * (1.0 - min(1.0, load)) if kind == BIG else fixed
            t = fixed1 + per * step1 if q == 1 else cdiv(q, occ) * fixed + per * q * shared
            if kind == BIG and q > 1:
                t = fixed + q * (per * shared + 4.0)
            if kind == MID and bn == 256 and s == 1 and tiles > CUS:
                rounds = cdiv(tiles, CUS)
                t = fixed + (rounds - 1) * per * full + per * (light if load - (rounds - 1) <= 0.6 else full) + load * 6.0
            if s > 1:
                t += 3.0 + 3.0 * s * m * n * 4.0 / 8.0e6
            if bm == 64 and m > 192:
                t *= 1.03
            if t < best_t:
                best_t, best = t, (kind, bm, bn, s)
    return best, best_t


def main():
    verbose = "-v" in sys.argv
    rows = [json.loads(l) for f in sys.argv[1:] if not f.startswith("-") for l in open(f) if l.startswith("{")]
    t = collections.OrderedDict()
    for r in rows:
        t.setdefault((r["m"], r["n"], r["k"]), {})[r["plan"]] = r["us"]
    tot, cnt, unknown = 0.0, 0, 0
    for (m, n, k), v in t.items():
        p, pt = plan(m, n, k)
        key = f"{p[0]}:{p[1]}:{p[2]}:{p[3]}"
        best_p, best_u = min(((a, b) for a, b in v.items() if a != "auto"), key=lambda a: a[1])
        got = v.get(key)
        if got is None:   # an unmeasured split factor: the mean of its measured neighbours, if both exist
            lo = max((s for s in (1, 2, 4, 8) if s <= p[3]), default=None)
            hi = min((s for s in (1, 2, 4, 8) if s >= p[3]), default=None)
            a, b = v.get(f"{p[0]}:{p[1]}:{p[2]}:{lo}"), v.get(f"{p[0]}:{p[1]}:{p[2]}:{hi}")
            if a is None or b is None:
                unknown += 1
                continue
            got = 0.5 * (a + b)
        reg = got / best_u - 1
        tot += reg
        cnt += 1
        if verbose:
            print(f"{m:6d} {n:6d} {k:5d}: picks {key:12s} {got:7.2f} (model {pt:6.1f})  library then {v['auto']:7.2f}  best pinned {best_p:12s} {best_u:7.2f}  "
                  f"regret {100 * reg:+5.1f} %{' <<<' if reg > 0.05 else ''}")
    print(f"mean regret {100 * tot / max(1, cnt):.2f} % over {cnt} shapes ({unknown} picks not measured)")


main()
