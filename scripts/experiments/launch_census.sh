#!/bin/bash
# Launch census of one config's training step (run on the GPU box from the repo root): scripts/experiments/launch_census.sh <config> <tag>
# -> gpurun_out/census_<tag>.txt: launches and kernel time per step, by kernel, non-library kernels listed separately.
set -e
cfg=${1:-5}; tag=${2:-x}
root=$(pwd); out=$root/gpurun_out/census_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/scripts/bench_configs.py $cfg --steps 3 > $out/run.log 2>&1
cd $root
python3 scripts/experiments/launch_census.py $out > gpurun_out/census_$tag.txt
find $out -name "*kernel_trace.csv" -delete
