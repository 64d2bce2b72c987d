#!/bin/bash
# Dev helper (GPU box, repo root): kernel-trace stats of a short bench run -> gpurun_out/<tag>/top.txt
tag=${1:-q}
root=$(pwd)
mkdir -p $root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/stats -- python3 $root/bench.py --no-cpu-baseline --no-fp32-mode --no-extras --steps 10 --warmup 3 > $root/gpurun_out/$tag/stats.log 2>&1
cd $root
f=$(find gpurun_out/$tag/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/$tag/top.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels ms/step", tot / 1e6 / 13)
for r in rows[:40]:
    print(f'{r["Name"].split("(")[0][:90]:90s} calls/step {int(r["Calls"])/13:6.1f} avg_us {float(r["AverageNs"])/1e3:9.1f} ms/step {float(r["TotalDurationNs"])/1e6/13:7.3f}')
PY
find gpurun_out/$tag -name "*kernel_trace.csv" -delete
head -30 gpurun_out/$tag/top.txt
