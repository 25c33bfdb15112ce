#!/bin/bash
# SQ counters of the fused control-network backward kernels: bash tools/k2_pmc.sh cfg3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CFG=${1:-cfg3}
OUT=gpurun_out/k2pmc
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-include-regex "unet_bwd_tile|unet_wgrad_kernel" -f csv -d $OUT -o pmc_$CFG -- python3 tools/k2_bench.py $CFG > $OUT/$CFG.log 2>&1
f=$(find $OUT -name "pmc_${CFG}_counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    m = {c: sum(x) / len(x) for c, x in v.items()}
    for c, val in sorted(m.items()):
        print(f"   {c:28s} {val:16.0f}")
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print(f"   MFMA busy / (4 SIMD x busy cycles) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CYCLES']):.3f}")
PY
