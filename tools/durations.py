#!/usr/bin/env python3
"""Fold a rocprofv3 --kernel-trace --stats kernel_stats csv of one bench.py workload into profiles/kernel_durations.json:
average launch duration per kernel class (what bench.py prints beside its HIP-event figures).

    tools/durations.py <workload> <kernel_stats.csv>
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from traffic import kernel_class  # noqa: E402


def main():
    wl, path = sys.argv[1:3]
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        c = kernel_class(r["Name"])
        if c:
            tot[c] += float(r["TotalDurationNs"])
            n[c] += int(r["Calls"])
    out = os.path.join(ROOT, "profiles", "kernel_durations.json")
    db = json.load(open(out)) if os.path.exists(out) else {}
    db[wl] = {c: {"avg_ms": tot[c] / n[c] * 1e-6, "launches": n[c], "source": os.path.basename(path)} for c in tot}
    json.dump(db, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(db[wl], indent=1))


if __name__ == "__main__":
    main()
