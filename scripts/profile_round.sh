#!/bin/bash
# Round profiles (run on the GPU box, from the repo root): scripts/profile_round.sh <tag>
#   bench (config 2): kernel trace + stats; FETCH_SIZE and WRITE_SIZE in separate PMC passes; one SQ/GRBM pass (MFMA busy, wave
#   states, active cycles -> in-kernel clock).  configs 3, 4, 5 (scripts/bench_configs.py): kernel stats + FETCH / WRITE passes.
# PMC passes are never combined with other trace domains; the program follows `--` directly (python3, no shell / env hop).
# Summaries for profiles/ are produced afterwards by scripts/summarize_profile.py (--steps = timed + warm-up steps of the run: 14 for the
# bench passes below, 3 + 5 for the config stats passes).
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/$tag
rm -rf $out          # (on the box; the LOCAL gpurun_out/<tag> of an earlier pass must be removed by hand before the call: gpurun merges into it and the summaries would mix the two)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py --no-cpu-baseline --no-fp32-mode --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 $B --steps 10 --warmup 3 > $out/bench_stats.log 2>&1
echo "bench stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/bench_fetch -- python3 $B --steps 3 --warmup 1 > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/bench_write -- python3 $B --steps 3 --warmup 1 > $out/bench_write.log 2>&1
echo "bench traffic done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d $out/bench_sq -- python3 $B --steps 3 --warmup 1 > $out/bench_sq.log 2>&1
echo "bench sq done"
for c in 3 4 5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg${c}_stats -- python3 $root/scripts/bench_configs.py $c --steps 3 > $out/cfg${c}_stats.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cfg${c}_fetch -- python3 $root/scripts/bench_configs.py $c --steps 2 > $out/cfg${c}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cfg${c}_write -- python3 $root/scripts/bench_configs.py $c --steps 2 > $out/cfg${c}_write.log 2>&1
    echo "config $c done"
done
cd $root
# keep what travels back small: the per-dispatch traces are large, the stats / counter tables are what the summaries need
find $out -name "*kernel_trace.csv" -size +20M -delete
ls -la $out | head -40
