"""Static instruction table of the decoder tile of ifd::optimize_kernel<8, 1> by source section, from a -gline-tables-only assembly
listing (every instruction attributed to the source line of its .loc, lines to the "[pcsamp:<section>]" tags in csrc/), next to the
cycles the tile-internal trace measured for the same sections (profiles/r04_tile_trace.txt):
    cd if-defense_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. -I../../include -gline-tables-only -x hip -S --cuda-device-only optimize.hip -o /tmp/opt_g.s
    python scripts/tile_static_table.py /tmp/opt_g.s > profiles/r04_tile_instruction_table.txt
Only the two basic blocks of the tile loop that hold the MFMAs (forward / backward) and the blocks between them are counted; the
loss-reporting branch (log1pf of the last step) is listed separately - it is skipped on 500 of 501 steps."""
import collections, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pcsamp_hist import line_tags

tags = line_tags()
src = open(sys.argv[1]).read().split("\n")
files = {}
for l in src:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(2))
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"\s*(md5.*)?$', l)
    if m and int(m.group(1)) not in files:
        files[int(m.group(1))] = os.path.basename(m.group(2))
start = next(i for i, l in enumerate(src) if l.startswith("_ZN3ifd15optimize_kernelILi8ELi1E"))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
# the tile loop: from the first block that holds >= 100 MFMAs to the end of the last one
blocks, cur = [], None
for i in range(start, end):
    s = src[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s) or re.match(r"^; %bb\.(\d+):", s)
    if m:
        cur = [m.group(0), i, 0]
        blocks.append(cur)
    elif cur is not None and s.startswith("v_mfma"):
        cur[2] += 1
big = [b for b in blocks if b[2] >= 100]
lo, hi = big[0][1], next(b[1] for b in blocks if b[1] > big[-1][1])


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return None


tab = collections.OrderedDict()
where = "untagged"
lossy = False
for i in range(lo, hi):
    s = src[i].strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        fn, ln = files.get(int(m.group(1)), "?"), int(m.group(2))
        rows = tags.get(fn)
        where = rows[ln - 1] if rows and 0 < ln <= len(rows) else fn
        continue
    if re.match(r"^; %bb\.|^\.LBB", s):
        # the two log1pf blocks (loss reporting) sit between the forward block's halves
        lossy = False
        continue
    if not s or s.startswith((";", ".", "//")):
        continue
    k = kind(s.split()[0])
    if k is None:
        continue
    if "frexp" in s or "v_cvt_f64_f32" in s:
        lossy = True
    tab.setdefault(where, collections.Counter())[k] += 1
print("decoder tile of optimize_kernel<8, 1>: instructions per 32-point tile by source section (static count of the tile loop's blocks,")
print("incl. the loss-reporting branch of the last step, ~230 vector instructions under tile.logit that 500 of 501 steps skip)")
print("%-24s %6s %6s %6s %6s %6s %6s | issue cycles at 32 / MFMA, 4 / vector, 2.5 / LDS" % ("section", "mfma", "valu", "lds", "vmem", "wait", "nop"))
tot = collections.Counter()
for sec, c in tab.items():
    tot.update(c)
    print("%-24s %6d %6d %6d %6d %6d %6d | %7.0f" % (sec, c["mfma"], c["valu"], c["lds"], c["vmem"], c["wait"], c["nop"],
                                                     32 * c["mfma"] + 4 * c["valu"] + 2.5 * c["lds"]))
print("%-24s %6d %6d %6d %6d %6d %6d | %7.0f" % ("total", tot["mfma"], tot["valu"], tot["lds"], tot["vmem"], tot["wait"], tot["nop"],
                                                 32 * tot["mfma"] + 4 * tot["valu"] + 2.5 * tot["lds"]))
