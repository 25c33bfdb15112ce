#!/bin/bash
# Round-end refresh on the GPU box: bench line, rocprofv3 kernel stats + PMC passes (condensed on the box: the raw
# traces exceed what gpurun copies back), d = 64 kernel averages.  Outputs: gpurun_out/refresh/.
set -u
export TMPDIR=/tmp
R=gpurun_out/refresh
rm -rf gpurun_out/prof_r1 gpurun_out/trace_cfg5r $R; mkdir -p $R
python3 bench.py > $R/bench.json 2> $R/bench.err
bash tools/profile_r1.sh > $R/profile_r1.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_r1 $R/r1 > $R/summarize.log 2>&1
du -sh gpurun_out/prof_r1/* > $R/du.log 2>&1
rm -rf gpurun_out/prof_r1
bash tools/trace_cfg.sh cfg5r > $R/cfg5_kernels.txt 2>&1
rm -rf gpurun_out/trace_cfg5r
python3 tools/quick_bench.py ouq20 cfg2 --loss 2>&1 | tail -6 > $R/quick.log
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "specialised" 2>&1 | tail -3 > $R/four_waves_test.log
du -sh gpurun_out >> $R/du.log
