#!/usr/bin/env python
"""Instruction mix per basic block of a kernel in a hipcc -S listing (which blocks hold the MFMAs, how many vector
instructions ride next to them, scratch traffic).   python scripts/asm_mix.py file.s kernel-name-substring [min_mfma]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_waitcnt",)): return "wait"
    if op.startswith(("s_nop",)): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    inside = False
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = collections.Counter()
    detail = collections.defaultdict(collections.Counter)
    for line in open(path):
        if not inside:
            if line.startswith(name) and ":" in line[:400]:
                inside = True
            continue
        s = line.strip()
        if s.startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = collections.Counter()
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        k = classify(op)
        blocks[cur][k] += 1
        detail[cur][op] += 1
    tot = collections.Counter()
    for b, c in blocks.items():
        tot.update(c)
        if c["mfma"] >= min_mfma:
            print(b, dict(c))
            print("   top valu:", [(o, n) for o, n in detail[b].most_common(40) if classify(o) in ("valu", "lane")])
    print("TOTAL", dict(tot))


if __name__ == "__main__":
    main()
