#!/usr/bin/env python
"""Where a kernel's scratch (spill) traffic sits: scratch instructions per basic block with the loop depth hipcc prints
in the block header.   python scripts/asm_spills.py file.s kernel-name-prefix"""
import re
import sys

path, name = sys.argv[1], sys.argv[2]
inside = False
cur, depth = "entry", 0
rows = []
cnt = {"ld": 0, "st": 0, "n": 0}
for line in open(path):
    if not inside:
        if line.startswith(name) and ":" in line[:400]:
            inside = True
        continue
    s = line.strip()
    if s.startswith("s_endpgm"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", s)
    if m:
        rows.append((cur, depth, dict(cnt)))
        cur = m.group(1)
        d = re.search(r"Depth=(\d+)", m.group(2))
        depth = int(d.group(1)) if d else 0
        cnt = {"ld": 0, "st": 0, "n": 0}
        continue
    if not s or s.startswith((";", ".")):
        continue
    cnt["n"] += 1
    if s.startswith("scratch_load"):
        cnt["ld"] += 1
    if s.startswith("scratch_store"):
        cnt["st"] += 1
rows.append((cur, depth, dict(cnt)))
tot = {}
for b, d, c in rows:
    t = tot.setdefault(d, [0, 0, 0])
    t[0] += c["ld"]; t[1] += c["st"]; t[2] += c["n"]
    if c["ld"] + c["st"] >= 4:
        print("%-12s depth %d: %4d loads %4d stores of %5d instructions" % (b, d, c["ld"], c["st"], c["n"]))
for d in sorted(tot):
    print("depth %d total: %d loads, %d stores, %d instructions" % (d, *tot[d]))
