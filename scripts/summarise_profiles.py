"""Reduce the rocprofv3 output of scripts/collect_profiles.sh to the small files committed under profiles/."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


res = {}
f = find("stats", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "%s_bench_kernel_stats.csv" % tag), "w") as o:
        o.write(open(f).read())
    for r in rows:
        if "optimize_kernel" in r["Name"] and "onet" not in r["Name"]:
            res["optimize_kernel_calls"] = int(r["Calls"])
            res["optimize_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
            res["optimize_kernel_share_pct"] = float(r["Percentage"])


n_dispatch = {}


def pmc(sub):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if f:
        ids = set()
        for r in csv.DictReader(open(f)):
            if "optimize_kernel" in r["Kernel_Name"] and "onet" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                ids.add(r["Dispatch_Id"])
        n_dispatch[sub] = len(ids)
        keep = [r for r in csv.DictReader(open(f)) if "optimize_kernel" in r["Kernel_Name"] and "onet" not in r["Kernel_Name"]]
        if keep:
            with open(os.path.join(out, "%s_pmc_%s.csv" % (tag, sub)), "w", newline="") as o:
                w = csv.DictWriter(o, fieldnames=list(keep[0].keys()))
                w.writeheader()
                w.writerows(keep)
    return acc


def source_sha():
    """Hash of the optimiser kernel's sources: bench.py only quotes a traffic figure measured on the same kernel."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("optimize.hip", "optimize_kernel.h", "knn_device.h", "ifd_device.h"):
        h.update(open(os.path.join(root, "if-defense_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def calibration():
    """bytes actually read by scripts/gather_calib (every 128-byte line once, the tile's 4-lanes-per-line pattern) over
    what FETCH_SIZE reported for it; 2.0 (the guide's factor for wide streaming reads) if the run is missing."""
    f, j = find("calib", "*counter_collection.csv"), os.path.join(out, "calib.json")
    if not f or not os.path.exists(j):
        return 2.0, None
    try:
        known = json.loads(open(j).read().strip().splitlines()[-1])["bytes_per_launch"]
        rows = [r for r in csv.DictReader(open(f)) if "gather_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
        per = {}
        for r in rows:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        kb = per[sorted(per, key=int)[-1]]
        return known / (kb * 1024.0), {"known_bytes": known, "FETCH_SIZE_KB": kb}
    except Exception as e:       # noqa: BLE001
        return 2.0, {"error": str(e)}


fetch, write, tcc = pmc("fetch"), pmc("write"), pmc("tcc")
factor, calib = calibration()
if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
    # the passes run ONE file (bench.py --steps 1 --warmup 0): its optimiser launches (one, or two with the partial round first) are
    # summed, the figures are per AVERAGE launch like bench.py's roofline.achieved
    launches = max(1, n_dispatch.get("fetch", 1))
    rd = factor * fetch["FETCH_SIZE"] * 1024.0 / launches   # gfx950: FETCH_SIZE under-reports; factor calibrated on the tile's own pattern
    wr = write["WRITE_SIZE"] * 1024.0 / launches
    res.update({
        "command": "scripts/collect_profiles.sh %s (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum, "
                   "separate passes over python bench.py --steps 1 --warmup 0 --no-cpu-baseline)" % tag,
        "kernel": "ifd::optimize_kernel; one file = 2468 clouds x 1024 points x 501 Adam steps in launches_per_file launches",
        "launches_per_file": float(launches),
        "FETCH_SIZE_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_KB": write["WRITE_SIZE"],
        "correction": "read bytes = fetch_factor * FETCH_SIZE * 1024 with fetch_factor calibrated by scripts/gather_calib "
                      "(a known byte count in the tile's gather pattern; MI355X_MICROARCH.md, HBM section: 2.0 for wide "
                      "streaming reads); WRITE_SIZE taken at face value",
        "fetch_factor": factor, "fetch_calibration": calib, "kernel_source_sha": source_sha(),
        "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
        "optimize_kernel_hbm_bytes_per_launch": rd + wr, "per_cloud_bytes": (rd + wr) * launches / 2468.0,
        "algorithmic_gather_bytes_per_cloud": 788004864,
    })
    if "TCC_HIT_sum" in tcc:
        res["L2_hit_rate"] = tcc["TCC_HIT_sum"] / (tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"])
json.dump(res, open(os.path.join(out, "roofline_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
