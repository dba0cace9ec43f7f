#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel name: mean per dispatch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    n = max(len(v) for v in d.values())
    if n < 3: continue
    print(f"{k:60s} n={n}")
    for c, v in sorted(d.items()):
        print(f"    {c:32s} mean {sum(v)/len(v):16.1f}")
