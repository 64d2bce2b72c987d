#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace --stats CSV.  usage: step_breakdown.py <kernel_stats.csv> <steps incl. warm-up> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {tot / steps / 1e6:.3f} ms/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print(f'{float(r["TotalDurationNs"]) / steps / 1e6:8.3f} ms/step  x{int(r["Calls"]) / steps:5.1f}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:100]}')
