#!/bin/bash
# Profiling recipe of a round (run on the GPU box through gpurun): bash tools/profile_round.sh r2
# 1) rocprofv3 kernel trace + stats of the benchmark command, 2) separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters:
# never combined with the trace domains), condensed by tools/summarize_profile.py into gpurun_out/refresh/<round>/.
set -u
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
RND=${1:-r2}
OUT=gpurun_out/prof_$RND
R=gpurun_out/refresh/$RND
rm -rf $OUT $R; mkdir -p $OUT $R
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-burst"
# the counter passes see the headline configuration only (the secondary configurations launch the same kernel template with
# other shapes; their dispatches would be averaged into the per-launch traffic that bench.py reads back)
PMC_CMD="$CMD --no-secondary"
KERNELS="rollout_kernel|rollout4_kernel|rollout1_kernel|rollout_ctrl|socm_|stopping_|colsum|weights_stats|unet_bwd|unet_wgrad|mnet_"
# kernel_stats.csv: the headline configuration alone, so that the rollout kernel's AVERAGE is the number bench.py reports as
# roofline.kernel_ms; kernel_stats_full.csv: the whole default command (secondary configurations included)
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o $RND -- $PMC_CMD > $OUT/trace_bench.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_full -o $RND -- $CMD > $OUT/trace_full.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KERNELS" -f csv -d $OUT/pmc_fetch -o $RND -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KERNELS" -f csv -d $OUT/pmc_write -o $RND -- $PMC_CMD > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-include-regex "rollout_kernel|rollout4_kernel|rollout1_kernel|unet_bwd|unet_wgrad_kernel" -f csv -d $OUT/pmc_sq -o $RND -- $PMC_CMD > $OUT/pmc_sq.log 2>&1
python3 tools/summarize_profile.py $OUT $R > $R/summarize.log 2>&1
f=$(find $OUT/trace_full -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/kernel_stats_full.csv
python3 bench.py > $R/bench.json 2> $R/bench.err
# the fused control-network backward alone (cfg3 and the cfg5 slice), the iteration in both modes, the stage-chain floor
for c in cfg3 cfg5r; do bash tools/k2_prof.sh $c > $R/k2_$c.txt 2>&1; done
for c in cfg2 cfg3; do for m in eager graph; do bash tools/iter_prof.sh $c $m > $R/iter_${c}_$m.txt 2>&1; done; done
for m in eager graph; do bash tools/iter_prof.sh cfg5r $m > $R/iter_cfg5r_$m.txt 2>&1; done
bash tools/iter_prof.sh md graph > $R/iter_md_graph.txt 2>&1
# the pair-grid network's kernels alone at d = 64 (wide form) and d = 10, kernel averages + SQ counters
bash tools/prof_any.sh k3_cfg5r tools/k3_bench.py cfg5r > $R/k3_cfg5r.txt 2>&1
python3 tools/k3_bench.py cfg3 >> $R/k3_cfg5r.txt 2>&1
bash tools/pmc_any.sh "mnet_" k3 python3 tools/k3_bench.py cfg5r > $R/k3_cfg5r_pmc.txt 2>&1
# non-default architectures: variant library vs descriptor-driven kernels (the variants are prebuilt in-tree)
# (the 512_256_128 variant is not shipped: SOCMX_SPECIALIZE=1 compiles it here, as backend.specialize_arch does at first use)
for h in "128 64 32" "512 256 128"; do SOCMX_SPECIALIZE=1 python3 tools/arch_bench.py $h 2>&1 | grep hdims >> $R/arch_variants.txt; done
# the sharded code path on one GPU: launcher + RCCL at world size 1 (collectives inside the captured graph)
python3 bench.py --gpus 1 --spawn --defer-graph --steps 20 --warmup 5 --no-cpu-baseline --no-burst > $R/bench_sharded_world1.json 2> $R/bench_sharded_world1.err
# the d = 64 contraction kernels alone (with the forward kernel's in-kernel cycle counters) and the two issue micro-benchmarks
# their schedule is built on
(hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -DSOCMX_CONTRACTION_PROF -o /tmp/cb tools/ubench/contraction_bench.hip 2>/dev/null && /tmp/cb 64 400 512 5) > $R/contraction_cfg5r.txt 2>&1
(hipcc -O3 --offload-arch=gfx950 -o /tmp/mi tools/ubench/mfma_issue.hip 2>/dev/null && /tmp/mi) > $R/mfma_issue.txt 2>&1
(hipcc -O3 --offload-arch=gfx950 -o /tmp/ov tools/ubench/mfma_valu_overlap.hip 2>/dev/null && /tmp/ov) > $R/mfma_valu_overlap.txt 2>&1
# the 4-row tile's building blocks: lane mapping + k-group sum against the CPU, the 4x4x1 issue rate, the per-CU weight stream
(hipcc -O3 --offload-arch=gfx950 -o /tmp/r4c tools/ubench/r4_check.hip 2>/dev/null && timeout 60 /tmp/r4c) > $R/r4_check.txt 2>&1
(hipcc -O3 --offload-arch=gfx950 -o /tmp/m44 tools/ubench/mfma4x4.hip 2>/dev/null && timeout 60 /tmp/m44) > $R/mfma4x4.txt 2>&1
(hipcc -O3 --offload-arch=gfx950 -o /tmp/l2r tools/ubench/l2ring.hip 2>/dev/null && timeout 60 /tmp/l2r) > $R/l2ring.txt 2>&1
# SQ counters of the one-row rollout kernel alone (VALU issue, waits, LDS conflicts)
bash tools/pmc_any.sh "rollout1_kernel" r1 python3 tools/quick_bench.py cfg3 > $R/rollout1_pmc.txt 2>&1
# the one-row design's building blocks: DPP-broadcast fmac rate, L2 stream / LDS weight reads beside it
(hipcc -O3 --offload-arch=gfx950 -o /tmp/vr1 tools/ubench/valu_row1.hip 2>/dev/null && timeout 120 /tmp/vr1 2.4) > $R/valu_row1_ubench.txt 2>&1
python3 tools/iter_bench.py md eager > $R/iter_md.txt 2>&1; python3 tools/iter_bench.py md graph >> $R/iter_md.txt 2>&1
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o stage_chain stage_chain.hip 2>/dev/null && ./stage_chain) > $R/stage_chain.txt 2>&1
python3 tools/quick_bench.py cfg3 cfg2 ouq20 ouh20 oul10 cfg5r cfg4r burst 2>&1 | grep -E "parity|rollout|iteration" > $R/quick.txt
# the two-tile burst kernel (csrc/socmx_rollout32.hip): rocprof average and SQ counters of 65,536-row launches, its phase table, the
# A/B against the 16-row form (bit-identity included) and the issue micro-benchmark its design rests on
bash tools/prof_any.sh burst32 tools/quick_bench.py burst --phases > $R/burst32.txt 2>&1
bash tools/pmc_any.sh "rollout32_kernel" b32 python3 tools/quick_bench.py burst > $R/burst32_pmc.txt 2>&1
python3 tools/burst_ab.py 2>&1 | grep -v amdgpu.ids > $R/burst_ab.txt
python3 tools/burst_settings.py 2>&1 | grep -v amdgpu.ids > $R/burst_settings.txt
(hipcc -O3 --offload-arch=gfx950 -o /tmp/mvm tools/ubench/mfma_valu_mix.hip 2>/dev/null && timeout 120 /tmp/mvm) > $R/mfma_valu_mix.txt 2>&1
rm -rf gpurun_out/pmc_b32
rm -rf $OUT gpurun_out/k2prof gpurun_out/iterprof gpurun_out/prof gpurun_out/pmc_k3
ls -la $R
# round 5: what a multiply-add costs on this chip's VALU form by form (explicit registers: VGPR banks, accumulator counts, waves
# per SIMD), and one unit of a packed-fma matrix-vector kernel with its activations by v_readlane_b32 / broadcast LDS reads
(hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o /tmp/vb tools/ubench/valu_banks.hip 2>/dev/null && timeout 120 /tmp/vb) > $R/valu_banks.txt 2>&1
(hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o /tmp/pku tools/ubench/pk_unit.hip 2>/dev/null && timeout 60 /tmp/pku) > $R/pk_unit.txt 2>&1
# the sharded default (several ranks: the autograd-free body eagerly, no RCCL call inside a graph) timed at world size 1
python3 bench.py --gpus 1 --spawn --no-dist-graph --steps 20 --warmup 5 --no-cpu-baseline --no-burst > $R/bench_sharded_world1_eager.json 2> $R/bench_sharded_world1_eager.err
# the one-row kernel's phase tables (needs the instrumented library: make -C soc-matching_amd/csrc PROF=1) at configs[2], soc.yaml's default
# d = 20 and the README's Linear OU (dense sigma), and the run-to-run determinism check
if [ -f soc-matching_amd/socmx/libsocmx_prof.so ]; then
  SOCMX_LIB=soc-matching_amd/socmx/libsocmx_prof.so python3 tools/r1_phases.py 128 > $R/r1_phases.txt 2>&1
  SOCMX_LIB=soc-matching_amd/socmx/libsocmx_prof.so python3 tools/r1_phases.py 128 OU_quadratic_easy 20 50 > $R/r1_phases_d20.txt 2>&1
  SOCMX_LIB=soc-matching_amd/socmx/libsocmx_prof.so python3 tools/r1_phases.py 64 OU_linear 10 100 > $R/r1_phases_oul10.txt 2>&1
fi
python3 tools/determinism_check.py cfg3_double_well_d10_K200 > $R/determinism_check.txt 2>&1
DET_SHARD=1 python3 tools/determinism_check.py tiny_double_well_d10 2>&1 | grep -v -E "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" >> $R/determinism_check.txt
# round 6: the captured sharded iteration (own RCCL communicators) soaked at world size 1, eager process-group collectives in front of every capture
python3 tools/soak_capture.py 120 2>&1 | grep -v -E "initialize_models|^RCCL|^HIP|^ROCm|^Hostname|^Librccl" > $R/soak_capture.txt
# round 6: the SAVED control-network backward -- the rollout's activation slabs / sign records against kernel A's own, the gradients of the two
# backward entries, the rollout with and without the export (stand-alone)
(python3 tools/dbg_saved.py double_well 10 200 128; python3 tools/dbg_saved.py OU_quadratic_easy 2 50 128; python3 tools/dbg_saved.py OU_linear 10 100 64) 2>&1 | grep -v amdgpu.ids > $R/saved_backward.txt
