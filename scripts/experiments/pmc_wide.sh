#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the split GEMM epilogues (run on the GPU box from the repo root): scripts/experiments/pmc_wide.sh
root=$(pwd); out=$root/gpurun_out/pmc_wide; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $root/scripts/experiments/gemm_wide_bench.bin pmc > $out/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $root/scripts/experiments/gemm_wide_bench.bin pmc > $out/fetch.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
for what in ("write", "fetch"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_wide/{what}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(what, k, ["%.3f GB" % (x * 1024 / 1e9) for x in v])
PY
