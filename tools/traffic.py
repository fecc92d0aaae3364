#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/pmc.sh) of one bench.py workload into
profiles/traffic.json: HBM-side bytes per launch per kernel class.

    tools/traffic.py <workload> <fetch counter csv glob> <write counter csv glob>

FETCH_SIZE and WRITE_SIZE count kilobytes at the L2's memory side; on gfx950 FETCH_SIZE tallies 128-byte
read requests as 64 bytes, so it is doubled (MI355X_MICROARCH.md, "HBM").  Infinity-Cache hits are
included in both, so this is an upper bound on DRAM traffic.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_class(name):
    if "post_kernel" in name or "post_stream_kernel" in name:
        return "post"
    if "conv_kernel" in name or "block_kernel" in name or "block1_kernel" in name:
        return "conv"
    if "stem_kernel" in name:
        return "stem"
    return None


def per_class(pattern, counter):
    f = sorted(glob.glob(pattern, recursive=True))[0]
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        c = kernel_class(r["Kernel_Name"])
        if c and r["Counter_Name"] == counter:
            tot[c] += float(r["Counter_Value"])
            n[c] += 1
    return tot, n


def main():
    wl, fetch_glob, write_glob = sys.argv[1:4]
    ft, fn = per_class(fetch_glob, "FETCH_SIZE")
    wt, wn = per_class(write_glob, "WRITE_SIZE")
    path = os.path.join(ROOT, "profiles", "traffic.json")
    db = json.load(open(path)) if os.path.exists(path) else {}
    db[wl] = {}
    for c in sorted(set(ft) | set(wt)):
        rd = 2.0 * 1024.0 * ft[c] / max(fn[c], 1)
        wr = 1024.0 * wt[c] / max(wn[c], 1)
        db[wl][c] = {"bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                     "launches_sampled": int(fn[c]),
                     "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1"}
    json.dump(db, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(db[wl], indent=1))


if __name__ == "__main__":
    main()
