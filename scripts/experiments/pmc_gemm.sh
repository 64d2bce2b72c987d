#!/bin/bash
# PMC passes over the GEMM harness (dev): LDS conflicts, wave-state and instruction-mix counters for the 256-thread and the double-buffered kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/p1 -o p1 --output-format csv -- $R/scripts/experiments/gemm_variants_bench.bin pmc > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/p2 -o p2 --output-format csv -- $R/scripts/experiments/gemm_variants_bench.bin pmc > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d $OUT/p3 -o p3 --output-format csv -- $R/scripts/experiments/gemm_variants_bench.bin pmc > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY -d $OUT/p4 -o p4 --output-format csv -- $R/scripts/experiments/gemm_variants_bench.bin pmc > $OUT/p4.log 2>&1
ls -R $OUT | head -40
