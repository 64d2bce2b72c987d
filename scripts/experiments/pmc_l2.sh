#!/bin/bash
# L2 hit rate and request volume per kernel over a few bench steps (run on the GPU box from the repo root): scripts/experiments/pmc_l2.sh
root=$(pwd); out=$root/gpurun_out/pmc_l2; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $out/l2 -- python3 $root/bench.py --no-cpu-baseline --no-fp32-mode --no-extras --steps 2 --warmup 1 > $out/l2.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_l2/l2/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_REQ_sum": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:25]
for k, v in rows:
    req = v.get("TCC_REQ_sum", 0); hit = v.get("TCC_HIT_sum", 0); miss = v.get("TCC_MISS_sum", 0)
    print(f"{k:70s} launches {n[k]:4d}  req/launch {req / max(n[k], 1) / 1e6:9.2f} M  hit {hit / max(hit + miss, 1):6.3f}  miss/launch {miss / max(n[k], 1) / 1e6:8.2f} M")
PY
