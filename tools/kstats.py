#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [n_iterations]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot/1e6:.2f} ms  ({tot/1e6/nit:.3f} ms per iteration over {nit})")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:7.1f} us {float(r['Percentage']):5.1f}%")
