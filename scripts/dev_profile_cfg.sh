#!/bin/bash
# dev: kernel stats of scripts/bench_configs.py <cfg>
export TMPDIR=/tmp
cfg=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg$cfg -- python3 scripts/bench_configs.py $cfg > gpurun_out/prof_cfg$cfg.log 2>&1
python3 - <<PY
import glob, csv
f = glob.glob("gpurun_out/prof_cfg$cfg/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 7
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print("config $cfg: sum of kernel time per step: %.2f ms" % tot)
for r in rows[:16]:
    print("%-64s calls/step %7.1f  ms/step %8.3f  avg %8.1f us  %5s%%" % (r["Name"][:64], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
