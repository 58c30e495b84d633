"""Static check of gfx950 code for the hazard found in round 5: a packed-f32 instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32:
two passes through the vector ALU) whose result is read by the NEXT vector instruction with nothing but an s_waitcnt in between.
hipcc's hazard recogniser (ROCm 7.2) counts the s_waitcnt as the wait state such a pair needs; the hardware spends no cycle on a wait
whose condition already holds, and the consumer then reads the OLD register - timing dependent: it only happens when the awaited load
has already arrived (scripts/pk_waitcnt_hazard.hip isolates it; profiles/r05_pk_waitcnt_hazard.txt).

    python scripts/asm_pk_hazard.py file.s [kernel-name-substring]          (hipcc -S listing or llvm-objdump -d output)

tests/test_abi_cpu.py runs scan() over the disassembly of every code object of the shipped libifd.so."""
import re
import sys

_REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")
_LABEL = re.compile(r"^(?:[0-9a-f]+\s+<)?([A-Za-z_][\w.$]*)>?:")


def _regs(tok):
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan(lines, pat=""):
    """-> {kernel: [(line no of the packed producer, its text, line no of the consumer, its text), ...]}"""
    cur = None
    prev = None         # (line no, text, written registers, an s_waitcnt has followed) of the last packed instruction
    found = {}
    for i, l in enumerate(lines):
        m = _LABEL.match(l)
        if m and not m.group(1).startswith(".L"):
            cur = m.group(1) if pat in m.group(1) else None
            prev = None
            continue
        t = l.split("//")[0].strip()
        if cur is None or not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        if op == "s_waitcnt":
            if prev is not None:
                prev = (prev[0], prev[1], prev[2], True)
            continue
        if prev is not None and prev[3] and op.startswith("v_"):
            srcs = set()
            for o in t[len(op):].split(",")[1:]:
                srcs |= _regs(o)
            if srcs & prev[2]:
                found.setdefault(cur, []).append((prev[0] + 1, prev[1], i + 1, t))
        prev = None
        if op.startswith("v_pk_") and op.endswith("_f32"):
            prev = (i, t, _regs(t[len(op):].split(",")[0]), False)
    return found


if __name__ == "__main__":
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    found = scan(open(sys.argv[1]).read().split("\n"), pat)
    for k, v in found.items():
        print("%s: %d packed-f32 results read across a bare s_waitcnt" % (k[:60], len(v)))
        for a in v[:6]:
            print("   line %d: %s\n   line %d: %s" % a)
    if not found:
        print("no packed-f32 result is read across a bare s_waitcnt in kernels matching '%s'" % pat)
