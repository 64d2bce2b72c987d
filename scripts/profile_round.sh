#!/bin/bash
# Round profile of the default bench: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
# (never combined with other trace domains).  usage (on the GPU box): scripts/profile_round.sh <tag>
set -e
tag=${1:-r01}
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --no-cpu-baseline --no-fp32-mode --steps 3 --warmup 1 > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --no-cpu-baseline --no-fp32-mode --steps 3 --warmup 1 > gpurun_out/${tag}_write.log 2>&1
tail -1 gpurun_out/${tag}_bench.log | cut -c1-400
