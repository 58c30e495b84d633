"""Histogram of rocprofv3 PC samples of ifd::optimize_kernel (scripts/pcsamp.sh).

    python scripts/pcsamp_hist.py <rocprofv3 output dir> <out.json> [<out.txt>]

Reads every *pc_sampling*.csv under the directory (stochastic or host_trap), keeps the samples whose dispatch is an
optimize_kernel launch (kernel trace of the same run), and counts per instruction: samples, samples in which the wave
issued, stall reasons, instruction type.  The library under test is built with -gline-tables-only, so the comment column
names the source line of every instruction; lines are grouped into the kernel's phases / tile sections by line range
(tags in optimize.hip / knn_device.h: "// [pcsamp:<name>]" opens a range that lasts to the next tag of the file).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line_tags():
    tags = {}
    for f in ("optimize.hip", "knn_device.h", "ifd_device.h"):
        cur, rows = "untagged", []
        for i, l in enumerate(open(os.path.join(ROOT, "if-defense_amd", "csrc", f)), 1):
            m = re.search(r"\[pcsamp:([a-z0-9_.-]+)\]", l)
            if m:
                cur = m.group(1)
            rows.append(cur)
        tags[f] = rows
    return tags


def main():
    d, out_json = sys.argv[1], sys.argv[2]
    out_txt = sys.argv[3] if len(sys.argv) > 3 else None
    tags = line_tags()
    disp = set()
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "optimize_kernel" in r.get("Kernel_Name", ""):
                disp.add(r.get("Dispatch_Id"))
    files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
    per = collections.defaultdict(lambda: {"n": 0, "issued": 0, "stall": collections.Counter(), "type": collections.Counter()})
    total = kept = 0
    cols = None
    for f in files:
        rd = csv.DictReader(open(f))
        cols = rd.fieldnames
        for r in rd:
            total += 1
            if disp and r.get("Dispatch_Id") not in disp:
                continue
            kept += 1
            key = (r.get("Instruction", "?"), r.get("Instruction_Comment", ""))
            e = per[key]
            e["n"] += 1
            wi = r.get("Wave_Issued_Instruction")
            if wi is not None and wi.strip() in ("1", "true", "True"):
                e["issued"] += 1
            sr = r.get("Stall_Reason")
            if sr:
                e["stall"][sr.replace("ROCPROFILER_PC_SAMPLING_INSTRUCTION_NOT_ISSUED_REASON_", "")] += 1
            it = r.get("Instruction_Type")
            if it:
                e["type"][it.replace("ROCPROFILER_PC_SAMPLING_INSTRUCTION_TYPE_", "")] += 1
    # group by source section
    sec = collections.defaultdict(lambda: {"n": 0, "issued": 0, "stall": collections.Counter(), "kind": collections.Counter()})
    by_line = collections.Counter()
    for (ins, com), e in per.items():
        m = re.search(r"([A-Za-z0-9_]+\.(?:hip|h|hpp|cpp)):(\d+)", com or "")
        name = "no-line-info"
        if m:
            fn, ln = m.group(1), int(m.group(2))
            rows = tags.get(fn)
            name = rows[ln - 1] if rows and 0 < ln <= len(rows) else fn
            by_line[(fn, ln)] += e["n"]
        s = sec[name]
        s["n"] += e["n"]
        s["issued"] += e["issued"]
        s["stall"].update(e["stall"])
        op = (ins or "?").split()[0]
        kind = ("mfma" if "mfma" in op else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "scratch_", "flat_"))
                else "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "other")
        s["kind"][kind] += e["n"]
    res = {"files": [os.path.basename(f) for f in files], "columns": cols, "samples_total": total, "samples_optimize_kernel": kept,
           "sections": {k: {"samples": v["n"], "frac": round(v["n"] / max(1, kept), 4), "issued_frac": round(v["issued"] / max(1, v["n"]), 3),
                            "stall": dict(v["stall"].most_common()), "by_kind": dict(v["kind"].most_common())}
                        for k, v in sorted(sec.items(), key=lambda kv: -kv[1]["n"])},
           "top_instructions": [{"instruction": k[0], "where": k[1], "samples": e["n"], "issued": e["issued"],
                                 "stall": dict(e["stall"].most_common(4))}
                                for k, e in sorted(per.items(), key=lambda kv: -kv[1]["n"])[:150]],
           "top_lines": [{"file": k[0], "line": k[1], "samples": n} for k, n in by_line.most_common(80)]}
    json.dump(res, open(out_json, "w"), indent=1)
    if out_txt:
        with open(out_txt, "w") as o:
            o.write("PC samples of ifd::optimize_kernel: %d of %d samples\n" % (kept, total))
            o.write("%-28s %8s %6s %6s  kinds | stall reasons of the not-issued samples\n" % ("section", "samples", "frac", "issued"))
            for k, v in res["sections"].items():
                o.write("%-28s %8d %6.3f %6.3f  %s | %s\n" % (k, v["samples"], v["frac"], v["issued_frac"],
                                                          " ".join("%s:%d" % kv for kv in v["by_kind"].items()),
                                                          " ".join("%s:%d" % kv for kv in list(v["stall"].items())[:6])))
            o.write("\ntop instructions\n")
            for t in res["top_instructions"][:80]:
                o.write("%7d %6d  %-70s %s  %s\n" % (t["samples"], t["issued"], t["instruction"][:70], t["where"][-40:], t["stall"]))
    print(json.dumps({k: res[k] for k in ("samples_total", "samples_optimize_kernel", "columns")}))


if __name__ == "__main__":
    main()
