#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel stats + optional FETCH_SIZE / WRITE_SIZE PMC passes) into the committed summaries
under profiles/.   usage: summarize_profile.py <tag> <stats_dir> [<fetch_dir> <write_dir>] [steps_profiled]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, stats_dir = sys.argv[1], sys.argv[2]
fetch_dir = sys.argv[3] if len(sys.argv) > 4 else None
write_dir = sys.argv[4] if len(sys.argv) > 4 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


stats = one(os.path.join(stats_dir, "**", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
summary = {"tag": tag, "source": "rocprofv3 --kernel-trace --stats", "kernels": {}}
for r in rows:
    if "vnp_" in r["Name"].split("(")[0]:
        summary["kernels"][r["Name"].split("(")[0][:90]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                            "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}


def pmc(dirname, counter):
    f = one(os.path.join(dirname, "**", "*counter_collection.csv"))
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            a = acc[r["Kernel_Name"].split("(")[0][:90]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


if fetch_dir:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide
    # coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B streaming stores.
    fe, wr = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
    traffic = {}
    for k in fe:
        if "vnp_" not in k:
            continue
        n = max(fe[k][1], 1)
        rd = 2.0 * fe[k][0] * 1024 / n
        wt = wr[k][0] * 1024 / max(wr[k][1], 1) if k in wr else 0.0
        traffic[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wt, "hbm_bytes_per_launch": rd + wt}
    summary["traffic"] = traffic
    summary["traffic_note"] = "FETCH_SIZE*2*1024 + WRITE_SIZE*1024 per launch, separate --pmc passes (gfx950 correction per MI355X_MICROARCH.md)"
json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
print("wrote", os.path.join(out_dir, f"{tag}_summary.json"))
