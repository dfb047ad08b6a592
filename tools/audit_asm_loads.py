"""Audit a hipcc .s file: for every kernel, list compiler-emitted instructions that READ a register which is
the destination of an inline-asm global load still in flight.  hipcc does not know about such loads: a copy
(or any use) it places between the load and the hand-written s_waitcnt sees the register before the data has
landed.  Vector-memory operations are modelled as an in-order queue; s_waitcnt vmcnt(N) retires all but the
N youngest.  The listing is walked twice so that loads still in flight at a loop's end reach its header.

    python tools/audit_asm_loads.py file.s [kernel-name-substring]
"""
import re
import sys



def load_dest(line):
    """Destination registers of a global_load_dword[x2|x3|x4] (empty set for anything else, LDS-DMA included)."""
    m = re.search(r"global_load_dword(?:x[234])? (?:v\[(\d+):(\d+)\]|v(\d+)),", line)
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


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
            dests.update(load_dest(l))
    if not dests:
        continue
    last_wait = None
    bad = []
    fifo = []   # vector-memory operations in issue order: the set of registers an inline-asm load will write (empty
                # for LDS-DMA, stores and compiler-issued operations); s_waitcnt vmcnt(N) retires all but the N youngest
    in_asm = False

    def in_flight():
        regs = {}
        for line_no, rs in fifo:
            for r in rs:
                regs[r] = line_no
        return regs

    # two passes over the listing: the second one starts with whatever was still in flight at the end of the
    # first, which approximates values carried around a loop's back-edge to its header
    for i, l in [(i, l) for _ in range(2) for i, l in enumerate(lines)]:
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        mw = re.search(r"s_waitcnt .*vmcnt\((\d+)\)", l)
        if mw:
            n = int(mw.group(1))
            fifo = fifo[len(fifo) - n:] if n else []
            last_wait = i
            continue
        if re.match(r"\s*(global|buffer|flat|scratch)_(load|store|atomic)", l):
            fifo.append((i, load_dest(l) if in_asm else set()))
            continue
        if in_asm:
            continue
        # any instruction the compiler emitted that READS a register whose asm-issued load is still in flight
        mi = re.match(r"\s*((?:v|ds|global|buffer|flat)_\w+)\s+(.*)", l)
        if mi:
            ops = [o.strip() for o in mi.group(2).split(",")]
            srcs = ops if re.match(r"(ds_write|global_store|buffer_store|flat_store)", mi.group(1)) else ops[1:]
            state = in_flight()
            hit = []
            for o in srcs:
                for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", o):
                    hit += [r for r in range(int(lo), int(hi) + 1) if r in state]
                for r in re.findall(r"\bv(\d+)\b", o):
                    if int(r) in state:
                        hit.append(int(r))
            if hit:
                if (i, l.strip()) not in [(b[0], b[1]) for b in bad]:
                    bad.append((i, l.strip(), "load at line %d still in flight" % state[hit[0]]))
    print(f"{name[:100]}: load destinations {len(dests)} regs, suspicious copies: {len(bad)}")
    for b in bad[:6]:
        print("    ", b)
    total += len(bad)
print("TOTAL suspicious copies:", total)
sys.exit(1 if total else 0)
