#!/bin/bash
# SQ counters of the pruned Chamfer scan (run on the GPU box from the repo root)
root=$(pwd); out=$root/gpurun_out/pmc_chamfer; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/a -- python3 $root/scripts/experiments/chamfer_pmc.py 2 > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $out/b -- python3 $root/scripts/experiments/chamfer_pmc.py 2 > $out/b.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/pmc_chamfer/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if "chamfer" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:24s} {x / n[k][c] / 1e6:12.3f} M per launch ({n[k][c]} launches)")
PY
