#!/usr/bin/env python3
"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) into the per-kernel
traffic JSON under profiles/.  usage: pmc_traffic.py fetch.csv write.csv out.json
hbm_MB = (2*FETCH_SIZE + WRITE_SIZE) KB / 1024: on gfx950 FETCH_SIZE reports half the bytes of 16-B/lane reads
(MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are counted, so this is traffic beyond the per-XCD L2,
an upper bound on HBM bytes."""
import collections, csv, json, sys


def means(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/pmc_round.sh) over "
               "tools/bench_ops.py --size 64 --batch 256; mean per launch. hbm_MB = (2*FETCH_SIZE + WRITE_SIZE) KB / 1024 "
               "(gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md); Infinity-Cache hits are counted: traffic beyond "
               "the per-XCD L2, an upper bound on HBM bytes. igemm_kernel<0,2,2,32,true,0> averages the three 17.18-GFLOP "
               "stride-2 forward layers and the 100-channel head (algorithmic mean 49.8 MB/launch).",
       "kernels": {}}
for k in sorted(set(fetch) & set(write)):
    f, n = fetch[k]
    w, _ = write[k]
    if n < 3:
        continue
    out["kernels"][k] = {"launches": n, "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                         "hbm_MB_per_launch": round((2 * f + w) / 1024, 2)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k[:70]:70s} {v['hbm_MB_per_launch']:9.2f} MB/launch  (n={v['launches']})")
