#!/bin/bash
# instruction mix per kernel over a few steps (run on the GPU box from the repo root): scripts/experiments/pmc_insts.sh [config id]
root=$(pwd); out=$root/gpurun_out/pmc_insts; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ -n "$1" ]; then CMD="$root/scripts/bench_configs.py $1 --steps 1"; else CMD="$root/bench.py --no-cpu-baseline --no-fp32-mode --no-extras --steps 2 --warmup 1"; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/i -- python3 $CMD > $out/i.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_insts/i/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:30]
print(f"{'kernel':64s} {'launches':>8s} {'VALU/launch':>12s} {'VALU/VMEM':>10s} {'SALU/VALU':>10s} {'LDS/VALU':>9s}")
for k, v in rows:
    valu = v.get("SQ_INSTS_VALU", 0); vm = v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)
    print(f"{k:64s} {n[k]:8d} {valu / max(n[k], 1) / 1e6:10.2f} M {valu / max(vm, 1):10.1f} {v.get('SQ_INSTS_SALU', 0) / max(valu, 1):10.2f} {v.get('SQ_INSTS_LDS', 0) / max(valu, 1):9.2f}")
PY
