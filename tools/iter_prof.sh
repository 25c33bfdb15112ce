#!/bin/bash
# per-kernel breakdown of the SOCM iteration: bash tools/iter_prof.sh cfg2 graph
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CFG=${1:-cfg3}; MODE=${2:-graph}
mkdir -p gpurun_out/iterprof
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/iterprof -o it_${CFG}_$MODE -- python3 tools/iter_bench.py $CFG $MODE 20 > gpurun_out/iterprof/${CFG}_$MODE.log 2>&1
tail -1 gpurun_out/iterprof/${CFG}_$MODE.log
f=$(find gpurun_out/iterprof -name "it_${CFG}_${MODE}_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time per iteration (24 iterations incl. warm-up): {tot/24/1e3:.1f} us, {sum(int(r['Calls']) for r in rows)/24:.0f} launches")
for r in rows[:28]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls'])/24:6.1f}/it {float(r['TotalDurationNs'])/24/1e3:8.1f} us/it  {r['Name'][:100]}")
PY
