#!/usr/bin/env python3
"""Average the counters of a rocprofv3 counter_collection.csv per kernel name."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1], recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
