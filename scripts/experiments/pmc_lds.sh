#!/bin/bash
# LDS bank conflicts per kernel over a few steps (run on the GPU box from the repo root): scripts/experiments/pmc_lds.sh [config id]
# (no argument: bench.py = config 2; 3 / 4 / 5: scripts/bench_configs.py)
root=$(pwd); out=$root/gpurun_out/pmc_lds; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ -n "$1" ]; then
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --output-format csv -d $out/lds -- python3 $root/scripts/bench_configs.py $1 --steps 1 > $out/lds.log 2>&1
else
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --output-format csv -d $out/lds -- python3 $root/bench.py --no-cpu-baseline --no-fp32-mode --no-extras --steps 2 --warmup 1 > $out/lds.log 2>&1
fi
cd $root
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_lds/lds/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:25]
for k, v in rows:
    act = v.get("SQ_LDS_IDX_ACTIVE", 0)
    print(f"{k:70s} launches {n[k]:4d}  active {act:14.0f}  bank-conflict {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(act, 1):6.3f}  addr-conflict {v.get('SQ_LDS_ADDR_CONFLICT', 0) / max(act, 1):6.3f}")
PY
