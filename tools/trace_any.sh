#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script, top kernels by total time:  bash tools/trace_any.sh tools/kernel_times.py cfg5r
set -u
export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o kt -- python3 "$@" > /tmp/kt.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%10.1fus n=%4s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
