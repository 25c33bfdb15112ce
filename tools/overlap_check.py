#!/usr/bin/env python3
"""Which kernels ran WHILE a rollout kernel was running?  python tools/overlap_check.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
roll = [r for r in rows if "rollout_kernel" in r["Kernel_Name"]]
r0 = roll[len(roll) // 2]
a, b = int(r0["Start_Timestamp"]), int(r0["End_Timestamp"])
print(f"rollout {(b - a) / 1e3:.1f} us")
inside = [r for r in rows if a < int(r["Start_Timestamp"]) < b]
print(f"{len(inside)} kernels started inside it:")
for r in inside[:40]:
    print(f"  +{(int(r['Start_Timestamp']) - a) / 1e3:8.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
nxt = [r for r in rows if int(r["Start_Timestamp"]) >= b][:45]
print("after it:")
for r in nxt:
    print(f"  +{(int(r['Start_Timestamp']) - b) / 1e3:8.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
