#!/usr/bin/env python3
"""Print the kernel timeline of one SOCM iteration from a rocprofv3 kernel trace (gpurun_out/trace_iter)."""
import csv, glob, sys
f = glob.glob("gpurun_out/trace_iter/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rollout_kernel" in r["Kernel_Name"]]
# the last two rollouts that have loss kernels between them
pick = None
for a, b in zip(idx[:-1], idx[1:]):
    if b - a > 50:
        pick = (a, b)
a, b = pick
t0 = int(rows[a]["Start_Timestamp"])
busy = 0
print("queue  start_us   dur_us  name")
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Queue_Id"]:>5} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r["Kernel_Name"][:90]}')
print("iteration span us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
