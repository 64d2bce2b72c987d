#!/usr/bin/env python3
"""Markdown kernel table of a profiled workload, generated from profiles/<tag>_<name>_summary.json (scripts/summarize_profile.py) —
DESIGN.md §3 embeds the output between its `<!-- kernel-table:... -->` markers instead of a hand-written table.

usage: kernel_table.py <tag> <name> [--min-ms 0.05] [--update DESIGN.md]
"""
import json
import os
import re
import sys

argv = sys.argv[1:]
min_ms, update = 0.05, None
if "--min-ms" in argv:
    i = argv.index("--min-ms"); min_ms = float(argv[i + 1]); del argv[i:i + 2]
if "--update" in argv:
    i = argv.index("--update"); update = argv[i + 1]; del argv[i:i + 2]
tag, name = argv[0], argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{name}_summary.json")))
steps = S.get("steps_profiled_incl_warmup") or 1.0
rows = []
for k, v in S["kernels"].items():
    ms = v.get("ms_per_step", v["total_ms"] / steps)
    if ms < min_ms:
        continue
    t = S.get("traffic", {}).get(k)
    p = S.get("pipes", {}).get(k)
    rows.append((ms, k, v["calls"] / steps, v["avg_us"],
                 (t["hbm_bytes_per_launch"] / 1e9 if t else None), (t.get("TBps_at_avg_duration") if t else None),
                 (p.get("mfma_pipe_busy_frac") if p else None), (p.get("clock_GHz") if p else None)))
rows.sort(reverse=True)
out = [f"| kernel (`profiles/{tag}_{name}_kernel_stats.csv`) | launches / step | avg µs | ms / step | fabric GB / launch | TB/s | MFMA busy | clock GHz |",
       "|---|---|---|---|---|---|---|---|"]
fmt = lambda x, f: "" if x is None else format(x, f)
for ms, k, n, us, gb, tbs, busy, clk in rows:
    k = k.replace("void ", "").replace("|", "\\|")
    out.append(f"| `{k}` | {n:.0f} | {us:.0f} | {ms:.2f} | {fmt(gb, '.2f')} | {fmt(tbs, '.2f')} | {fmt(busy if busy and busy > 0.005 else None, '.2f')} | {fmt(clk, '.2f')} |")
vnp = sum(r[0] for r in rows)
out.append(f"| all kernels of the step ({S.get('all_kernels_ms_per_step', 0):.2f} ms / step; listed: {vnp:.2f}) | | | | | | | |")
text = "\n".join(out)
if update:
    path = os.path.join(ROOT, update)
    doc = open(path).read()
    a, b = f"<!-- kernel-table:{tag}_{name} -->", f"<!-- /kernel-table:{tag}_{name} -->"
    if a not in doc or b not in doc:
        raise SystemExit(f"markers {a} ... {b} not found in {update}")
    doc = re.sub(re.escape(a) + r".*?" + re.escape(b), a + "\n" + text + "\n" + b, doc, flags=re.S)
    open(path, "w").write(doc)
    print(f"updated {update}: {len(rows)} kernels")
else:
    print(text)
