#!/bin/bash
# (at most 4 TCC counters per pass: a fifth aborts rocprofv3, which then hangs in its signal handler)
# L2 / fabric counters of kernels matching a regex: bash tools/pmc_l2.sh "<regex>" <tag> python3 tools/k2_bench.py cfg3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
RE=$1; TAG=$2; shift 2
OUT=gpurun_out/pmcl2_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-include-regex "$RE" -f csv -d $OUT/a -o pmc -- "$@" > $OUT/run.log 2>&1
timeout 150 rocprofv3 --pmc TCC_REQ_sum TCC_EA0_WRREQ_sum --kernel-include-regex "$RE" -f csv -d $OUT/b -o pmc -- "$@" > $OUT/run2.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-include-regex "$RE" -f csv -d $OUT/c -o pmc -- "$@" > $OUT/run3.log 2>&1
tail -2 $OUT/run.log
python3 - $OUT <<'PY'
import csv, sys, collections, glob
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k)
        for c, d in sorted(v.items()):
            print(f"   {c:30s} {sum(d.values()) / len(d):16.0f}   ({len(d)} dispatches)")
PY
