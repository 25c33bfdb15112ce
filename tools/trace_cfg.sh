#!/bin/bash
# Kernel trace of tools/quick_bench.py for one config (default cfg5r --loss), aggregated per kernel.
set -u
export TMPDIR=/tmp
CFG=${1:-cfg5r}
OUT=gpurun_out/trace_$CFG
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/t -o it -- python3 tools/quick_bench.py $CFG --loss > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
    print(f'{float(r["TotalDurationNs"])/1e6:10.2f}ms  n={r["Calls"]:>5}  avg={float(r["AverageNs"])/1e3:10.1f}us  {r["Name"][:120]}')
PY
tail -3 $OUT/bench.log
