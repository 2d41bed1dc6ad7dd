#!/usr/bin/env python3
"""Summarises a tools/profile.sh output directory: per-kernel average of every counter."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for f in sorted(glob.glob(d + "/trace/*/*_kernel_stats.csv")):
    print(open(f).read().strip())
for f in sorted(glob.glob(d + "/pmc*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rc_draw" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:40s} {c:28s} n={len(v):3d} avg={sum(v) / len(v):.1f}")
