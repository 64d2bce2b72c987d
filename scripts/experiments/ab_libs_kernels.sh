#!/bin/bash
# Per-kernel times of one config under several builds of the library, on ONE box: scripts/experiments/ab_libs_kernels.sh <config> <pattern> lib1.so lib2.so ...
cfg=$1; pat=$2; shift 2
root=$(pwd); out=$root/gpurun_out/ab_libs; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for lib in "$@"; do
    name=$(basename $lib .so)_r$round
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 $root/scripts/experiments/with_lib.py $root/$lib $root/scripts/bench_configs.py $cfg --steps 5 > $out/$name.log 2>&1
    echo "== $name: $(grep -h 'ms/step\|ms per step' $out/$name.log | tail -1)"
    f=$(find $out/$name -name '*kernel_stats.csv' | head -1)
    python3 - "$f" "$pat" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(p in r["Name"] for p in sys.argv[2].split(","))]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    tot += float(r["TotalDurationNs"])
    print(f'   {r["Name"].split("(")[0][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
print(f"   total of the matching kernels: {tot / 1e6:.3f} ms over the run")
PY
done; done
