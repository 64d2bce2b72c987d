cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/p4 -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py 4 --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/p4.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/p4 -name "*kernel_trace.csv" -size +20M -delete
find $GRAFT_REPO_ROOT/gpurun_out/p4 -name "*stats*" | head
