#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun).  Outputs under gpurun_out/prof_r1/.
# 1) kernel trace + stats of the benchmark command, 2) separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r1
mkdir -p $OUT
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-burst"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o r1 -- $CMD > $OUT/trace_bench.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "rollout_kernel|socm_|colsum|weights_stats" -f csv -d $OUT/pmc_fetch -o r1 -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "rollout_kernel|socm_|colsum|weights_stats" -f csv -d $OUT/pmc_write -o r1 -- $CMD > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-include-regex "rollout_kernel" -f csv -d $OUT/pmc_sq -o r1 -- $CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
ls -la $OUT/*
