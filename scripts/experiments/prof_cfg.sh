#!/bin/bash
# kernel-time table of one config (run on the GPU box from the repo root): scripts/experiments/prof_cfg.sh <cfg> [steps]
cfg=${1:-4}; steps=${2:-3}
root=$(pwd); out=$root/gpurun_out/prof_cfg$cfg; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/scripts/bench_configs.py $cfg --steps $steps > $out/stats.log 2>&1
cd $root
python3 - "$out" "$steps" <<'PY'
import csv, glob, sys
out, steps = sys.argv[1], int(sys.argv[2]) + 5
f = glob.glob(out + "/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {tot / 1e6 / steps:.2f} ms / step over {steps} steps (5 warm-up)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f'{r["Name"][:100]:100s} {int(r["Calls"]) / steps:6.1f} x {float(r["AverageNs"]) / 1e3:8.1f} us = {float(r["TotalDurationNs"]) / 1e6 / steps:6.2f} ms')
PY
