#!/usr/bin/env python3
"""Last `n` launches of a rocprofv3 kernel_trace.csv in start order with durations and gaps: tools/trace_order.py <csv glob> [n]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1], recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev_end = None
tot = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].replace("void camal::", "").split("(")[0][:70]
    print(f"{name:70s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', '')}")
    prev_end = e
    tot += e - s
print("sum of durations", tot / 1e3, "us; span", (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3, "us")
