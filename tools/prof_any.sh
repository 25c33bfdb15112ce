#!/bin/bash
# rocprofv3 kernel-trace stats of any python tool: bash tools/prof_any.sh <tag> tools/x.py args...   (summary on stdout)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof -o $TAG -- python3 "$@" > gpurun_out/prof/$TAG.log 2>&1
tail -3 gpurun_out/prof/$TAG.log
f=$(find gpurun_out/prof -name "${TAG}_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls']):5d}  {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Name'][:110]}")
PY
