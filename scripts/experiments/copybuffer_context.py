#!/usr/bin/env python3
"""Which launches sit around the __amd_rocclr_copyBuffer / fillBuffer kernels of a step?  Reads a rocprofv3 kernel trace CSV (sorted by
start time) and prints, for the LAST step's worth of dispatches, each copy / fill kernel with its grid size and its two neighbours.
usage: copybuffer_context.py <kernel_trace.csv> [max lines]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:60] for r in rows]
# the last occurrence of the Adam kernel ends a step; take the dispatches between the last two
ends = [i for i, n in enumerate(names) if "vnp_adam" in n]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
step = list(range(lo, hi))
print(f"{len(step)} dispatches in the last step")
from collections import Counter
cnt = Counter()
shown = 0
for i in step:
    if "rocclr" in names[i]:
        g = rows[i].get("Grid_Size") or rows[i].get("Grid_Size_X")
        key = (names[i], names[i - 1], names[i + 1] if i + 1 < len(names) else "")
        cnt[key] += 1
for (n, before, after), c in cnt.most_common(lim):
    print(f"{c:4d} x {n:34s} after {before:60s} before {after}")
