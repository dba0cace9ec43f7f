#!/usr/bin/env python3
"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) into the per-kernel
traffic JSON under profiles/.  usage: pmc_traffic.py fetch.csv write.csv out.json [size batch]
hbm_MB = (2*FETCH_SIZE + WRITE_SIZE) KB / 1024: on gfx950 FETCH_SIZE reports half the bytes of 16-B/lane reads
(MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are counted, so this is traffic beyond the per-XCD L2,
an upper bound on HBM bytes."""
import collections, csv, json, sys


def means(path, counter, by_grid=False):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            key = r["Kernel_Name"] + (f" grid={r.get('Grid_Size', '?')}" if by_grid else "")
            acc[key].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
size, batch = (sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else ("64", "256")
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/pmc_round.sh) over "
               f"tools/bench_ops.py --size {size} --batch {batch}; mean per launch. hbm_MB = (2*FETCH_SIZE + WRITE_SIZE) KB / 1024 "
               "(gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md); Infinity-Cache hits are counted: traffic beyond "
               "the per-XCD L2, an upper bound on HBM bytes. An igemm_kernel<0,...> entry averages the stride-2 forward "
               "layers and the 100-channel head bench_ops.py runs (their algorithmic MB are printed by bench_ops.py).",
       "kernels": {}}
for k in sorted(set(fetch) & set(write)):
    f, n = fetch[k]
    w, _ = write[k]
    if n < 3:
        continue
    out["kernels"][k] = {"launches": n, "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                         "hbm_MB_per_launch": round((2 * f + w) / 1024, 2)}
# per-layer view of the implicit-GEMM family: the same kernel at different grid sizes = different layers of bench_ops.py
fg, wg = means(sys.argv[1], "FETCH_SIZE", True), means(sys.argv[2], "WRITE_SIZE", True)
out["igemm_by_grid"] = {}
for k in sorted(set(fg) & set(wg)):
    if any(t in k for t in ("igemm_kernel", "igemm_dma_kernel", "igemm_dma_x3_kernel", "igemm_x3_dgw_kernel", "igemm_bf16_dgw_kernel")) and fg[k][1] >= 3:
        out["igemm_by_grid"][k] = {"launches": fg[k][1], "hbm_MB_per_launch": round((2 * fg[k][0] + wg[k][0]) / 1024, 2)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k[:70]:70s} {v['hbm_MB_per_launch']:9.2f} MB/launch  (n={v['launches']})")
