#!/usr/bin/env python3
"""Chamfer entries of one bench line: python bench.py ... | python scripts/experiments/chamfer_line.py"""
import json, sys
d = json.loads(sys.stdin.read())
print("ms_per_step", d["ms_per_step"])
for v in d["roofline_other"]:
    if "hamfer" in v["kernel"]:
        print(v["kernel"][:70], {k: v[k] for k in v if k in ("ms", "frac", "achieved", "visited_fraction", "pairs_evaluated")})
