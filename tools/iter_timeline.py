#!/usr/bin/env python3
"""One steady-state iteration as a timeline (start -> end per kernel, queue), from the kernel trace tools/iter_prof.sh leaves in
gpurun_out/iterprof: python3 tools/iter_timeline.py cfg3 graph"""
import csv, glob, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, mode = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(ROOT, "gpurun_out", "iterprof", f"it_{cfg}_{mode}_kernel_trace.csv"))
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
roll = [i for i, r in enumerate(rows) if "rollout" in r["Kernel_Name"] and "ctrl" not in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
i0, i1 = roll[-3], roll[-2]
t0 = rows[i0]["s"]
for r in rows[i0:i1 + 1]:
    print(f"{(r['s'] - t0) / 1e3:8.1f} -> {(r['e'] - t0) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:100]}")
