#!/bin/bash
# Kernel trace of the SOCM iteration (bench.py), aggregated per kernel.  Run on the GPU box through gpurun.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/trace_iter
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/t -o it -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-burst > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_iter/t/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ns", tot)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f'{float(r["TotalDurationNs"])/1e3:10.1f}us  n={r["Calls"]:>5}  avg={float(r["AverageNs"])/1e3:8.1f}us  {r["Name"][:110]}')
PY
