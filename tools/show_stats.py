#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv (tools/profile.sh output)."""
import csv, glob, sys
pat = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
f = sorted(glob.glob(pat, recursive=True))[0]
print(f)
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.2f} {r['Percentage']}%")
