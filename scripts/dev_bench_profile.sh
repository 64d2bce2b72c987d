#!/bin/bash
# dev: default bench, then the same under rocprofv3 kernel trace; args: <tag>
set -e
tag=${1:-x}
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b_$tag.log 2> gpurun_out/b_$tag.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --no-fp32-mode --steps 5 > gpurun_out/bp_$tag.log 2>&1
python3 - <<PY
import json, glob, csv
d = json.loads(open("gpurun_out/b_$tag.log").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), "points/s", round(d["value"]), "gemm ms", round(d["roofline"]["gemm_ms_per_step"], 2),
      "fp32 mode ms", round(d.get("fp32_mfma_mode", {}).get("ms_per_step", 0), 2))
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 5 + 3   # timed + warmup
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print("sum of kernel time per step: %.3f ms" % tot)
for r in rows[:30]:
    print("%-70s calls/step %6.1f  ms/step %7.3f  avg %8.1f us  %5s%%" % (r["Name"][:70], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
