#!/bin/bash
# PMC passes on config 3's step for the EdgeConv kernels (run on the GPU box from the repo root): L2 hit / miss, L1 -> L2 requests and their
# latency, texture-addresser and L2 busy.  Counters only (no trace domains), the program directly after `--`.
set -e
root=$(pwd); out=$root/gpurun_out/pmc_edge
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/a -- python3 $root/scripts/bench_configs.py 3 --steps 1 > $out/a.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/b -- python3 $root/scripts/bench_configs.py 3 --steps 1 > $out/b.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TCC_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $out/c -- python3 $root/scripts/bench_configs.py 3 --steps 1 > $out/c.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
for tag in "abc":
    f = glob.glob(f"gpurun_out/pmc_edge/{tag}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(tag, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "edge_fused" in k or "edge_gather_u" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k in sorted(acc):
        print(tag, k[:60], {n: f"{v:.3g}" for n, v in acc[k].items()})
PY
