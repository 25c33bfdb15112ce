#!/bin/bash
# per-kernel breakdown of the SOCM iteration, STEADY STATE ONLY: bash tools/iter_prof.sh cfg2 graph
# (the kernel trace is cut at the start of the N-th last rollout launch: warm-up iterations, the capture and -- in graph mode --
#  the eager iterations in front of it are not averaged in)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CFG=${1:-cfg3}; MODE=${2:-graph}; N=${3:-20}
mkdir -p gpurun_out/iterprof
rm -rf gpurun_out/iterprof/it_${CFG}_${MODE}*
rocprofv3 --kernel-trace -f csv -d gpurun_out/iterprof -o it_${CFG}_$MODE -- python3 tools/iter_bench.py $CFG $MODE $N > gpurun_out/iterprof/${CFG}_$MODE.log 2>&1
tail -1 gpurun_out/iterprof/${CFG}_$MODE.log
f=$(find gpurun_out/iterprof -name "it_${CFG}_${MODE}_kernel_trace.csv" | head -1)
python3 - "$f" "$N" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
roll = [r for r in rows if "rollout" in r["Kernel_Name"] and "ctrl" not in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
assert len(roll) >= n, (len(roll), n)
t0 = roll[-n]["s"]
steady = [r for r in rows if r["s"] >= t0]
agg = collections.OrderedDict()
for r in steady:
    a = agg.setdefault(r["Kernel_Name"], [0, 0])
    a[0] += 1
    a[1] += r["e"] - r["s"]
tot = sum(a[1] for a in agg.values())
span = max(r["e"] for r in steady) - t0
print(f"steady state, {n} iterations: {span / n / 1e3:.1f} us wall per iteration, kernel time {tot / n / 1e3:.1f} us per iteration, "
      f"{len(steady) / n:.1f} launches per iteration")
for name, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{ns / c / 1e3:9.1f} us x{c / n:6.2f}/it {ns / n / 1e3:8.1f} us/it  {name[:110]}")
lib = [(k, v) for k, v in agg.items() if k.startswith("Cijk") or "gemm" in k.lower()]
print("library GEMM kernels in the steady state:", "none" if not lib else "; ".join(f"{k[:60]} x{v[0] / n:.2f}/it" for k, v in lib))
PY
