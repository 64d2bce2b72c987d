#!/bin/bash
# kernel-time table of the bench step (config 2; run on the GPU box from the repo root): scripts/experiments/prof_bench.sh [steps]
steps=${1:-5}
root=$(pwd); out=$root/gpurun_out/prof_bench; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline --no-fp32-mode --no-extras --steps $steps --warmup 3 > $out/stats.log 2>&1
cd $root
python3 - "$out" "$steps" <<'PY'
import csv, glob, sys
out, steps = sys.argv[1], int(sys.argv[2]) + 3
f = glob.glob(out + "/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {tot / 1e6 / steps:.2f} ms / step over {steps} steps (3 warm-up)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"]) / steps:6.1f} x {float(r["AverageNs"]) / 1e3:8.1f} us = {float(r["TotalDurationNs"]) / 1e6 / steps:6.2f} ms')
PY
find $out -name "*kernel_trace.csv" -size +20M -delete
