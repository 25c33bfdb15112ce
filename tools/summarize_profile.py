#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_r1.sh (gpurun_out/prof_r1/) into profiles/r1/.

    python tools/summarize_profile.py [gpurun_out/prof_r1] [profiles/r1]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r1"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r1"
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    f = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return f[0] if f else None


ks = find("trace", "*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r.get("MinNs", ""), r.get("MaxNs", "")])
ds = find("trace", "*domain_stats.csv")
if ds:
    shutil.copy(ds, os.path.join(dst, "domain_stats.csv"))
log = os.path.join(src, "trace_bench.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{\"metric\"")]
    if lines:
        open(os.path.join(dst, "bench_under_rocprof.json"), "w").write(lines[-1])

summary = collections.OrderedDict()
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    disp = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
        disp[k] = dict(grid=r.get("Grid_Size", ""), wg=r.get("Workgroup_Size", ""), vgpr=r.get("VGPR_Count", ""),
                       accum_vgpr=r.get("Accum_VGPR_Count", ""), sgpr=r.get("SGPR_Count", ""),
                       lds=r.get("LDS_Block_Size", ""), scratch=r.get("Scratch_Size", ""))
    for k, counters in acc.items():
        e = summary.setdefault(k, collections.OrderedDict(dispatch=disp[k]))
        for c, vals in sorted(counters.items()):
            per = collections.defaultdict(float)      # a counter may be reported per XCD/SE: add up per dispatch
            for d, v in vals:
                per[d] += v
            e[c] = dict(mean_per_dispatch=sum(per.values()) / len(per), dispatches=len(per))
# which library the counters were taken on: bench.py refuses a summary of another version (roofline.traffic stays null)
try:
    sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "soc-matching_amd")]
    from socmx import _lib
    summary["_meta"] = dict(socmx_version=int(_lib.lib().socmx_version()))
except Exception as e:  # noqa: BLE001
    summary["_meta"] = dict(socmx_version=None, error=str(e))
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
