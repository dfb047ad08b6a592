"""Audit a hipcc .s file: for every kernel, list compiler-inserted register copies (v_mov) whose SOURCE is the
destination of an inline-asm global load.  Such a copy placed between the load and the hand-written
s_waitcnt reads the register before the data has landed (the compiler does not know the load is still
in flight).  Copies located after the wait are harmless; this tool prints all of them with line numbers
and the position of the nearest preceding wait so they can be checked.

    python tools/audit_asm_loads.py file.s [kernel-name-substring]
"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
names = [n for n in re.findall(r"^(_Z\S+):", s, flags=re.M) if pat in n]
total = 0
for name in names:
    start = s.index("\n" + name + ":")
    end = s.find("s_endpgm", start)
    if end < 0:
        continue
    lines = s[start:end].split("\n")
    dests = set()
    in_asm = False
    for l in lines:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm:
            m = re.search(r"global_load_dwordx[24] v\[(\d+):(\d+)\]", l)
            if m:
                dests.update(range(int(m.group(1)), int(m.group(2)) + 1))
    if not dests:
        continue
    last_wait = None
    bad = []
    state = {}  # reg -> line of the asm load that is still in flight; cleared by an asm wait
    in_asm = False
    # two passes over the listing: the second one starts with whatever was still in flight at the end of the
    # first, which approximates values carried around a loop's back-edge to its header
    for i, l in [(i, l) for _ in range(2) for i, l in enumerate(lines)]:
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        if in_asm:
            m = re.search(r"global_load_dwordx[24] v\[(\d+):(\d+)\]", l)
            if m:
                for r in range(int(m.group(1)), int(m.group(2)) + 1):
                    state[r] = i
            if "s_waitcnt vmcnt(0)" in l:
                state.clear()
                last_wait = i
            continue
        m = re.match(r"\s*v_mov_b(?:32|64)_e32 \S+ v\[?(\d+)(?::(\d+))?\]?", l)
        if m:
            lo = int(m.group(1))
            hi = int(m.group(2)) if m.group(2) else lo
            hit = [r for r in range(lo, hi + 1) if r in state]
            if hit:
                if (i, l.strip()) not in [(b[0], b[1]) for b in bad]:
                    bad.append((i, l.strip(), "load at line %d still in flight" % state[hit[0]]))
    print(f"{name[:100]}: load destinations {len(dests)} regs, suspicious copies: {len(bad)}")
    for b in bad[:6]:
        print("    ", b)
    total += len(bad)
print("TOTAL suspicious copies:", total)
sys.exit(1 if total else 0)
