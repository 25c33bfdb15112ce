#!/bin/bash
# SQ counters of kernels matching a regex for any command: bash tools/pmc_any.sh "<regex>" <tag> python3 tools/kernel_times.py cfg5r
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
RE=$1; TAG=$2; shift 2
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-include-regex "$RE" -f csv -d $OUT -o pmc -- "$@" > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --kernel-include-regex "$RE" -f csv -d $OUT/b -o pmc -- "$@" > $OUT/run2.log 2>&1
tail -4 $OUT/run.log
python3 - $OUT <<'PY'
import csv, sys, collections, glob
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k)
        for c, d in sorted(v.items()):
            print(f"   {c:30s} {sum(d.values()) / len(d):16.0f}   ({len(d)} dispatches)")
PY
