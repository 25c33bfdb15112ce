#!/bin/bash
# kernel breakdown of the fused control-network backward: bash tools/k2_prof.sh [cfg3]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CFG=${1:-cfg3}
mkdir -p gpurun_out/k2prof
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/k2prof -o k2_$CFG -- python3 tools/k2_bench.py $CFG > gpurun_out/k2prof/$CFG.log 2>&1
tail -1 gpurun_out/k2prof/$CFG.log
f=$(find gpurun_out/k2prof -name "k2_${CFG}_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:10.1f} us  x{r['Calls']:>5}  {r['Percentage']:>6}%  {r['Name'][:110]}")
PY
