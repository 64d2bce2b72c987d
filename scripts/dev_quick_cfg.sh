#!/bin/bash
# Dev helper (GPU box, repo root): kernel-trace stats of a short run of a BASELINE config -> gpurun_out/<tag>/top.txt
cfg=${1:-3}; tag=${2:-qc}
root=$(pwd)
mkdir -p $root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/stats -- python3 $root/scripts/bench_configs.py $cfg --steps 10 > $root/gpurun_out/$tag/stats.log 2>&1
cd $root
f=$(find gpurun_out/$tag/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/$tag/top.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels ms (whole run)", tot / 1e6)
for r in rows[:45]:
    print(f'{r["Name"].split("(")[0][:90]:90s} calls {int(r["Calls"]):6d} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):6.2f}')
PY
find gpurun_out/$tag -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/$tag/stats.log
head -32 gpurun_out/$tag/top.txt
