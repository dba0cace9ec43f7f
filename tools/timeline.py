#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel_trace.csv: how much wall time has an MFMA conv kernel in flight,
what runs in the gaps.  usage: timeline.py kernel_trace.csv [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows]
ev.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = ev[int(len(ev) * skip)][0]          # analyse the steady-state tail
ev = [e for e in ev if e[0] >= t0]
span = max(e[1] for e in ev) - ev[0][0]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
mf = [(s, e) for s, e, n, q in ev if "igemm_kernel" in n or "c3_" in n and "mfma" in n]
allk = [(s, e) for s, e, n, q in ev]
print(f"span {span/1e6:.2f} ms; any kernel busy {union(allk)/span*100:.1f}%; MFMA-kernel in flight {union(mf)/span*100:.1f}%; "
      f"sum of MFMA kernel durations {sum(e-s for s,e in mf)/span*100:.1f}% of span")
# time with exactly 0 MFMA kernels: what else runs?
pts = []
for s, e in mf: pts += [(s, 1), (e, -1)]
pts.sort(); gaps = []; depth = 0; last = ev[0][0]
for t, d in pts:
    if depth == 0 and t > last: gaps.append((last, t))
    depth += d
    if depth == 0: last = t
acc = collections.Counter()
for s, e, n, q in ev:
    if "igemm_kernel" in n: continue
    for gs, ge in gaps:
        if ge <= s: continue
        if gs >= e: break
        acc[n[:48]] += max(0, min(e, ge) - max(s, gs))
print("gap time total %.2f ms; kernels running inside MFMA-free gaps (ms):" % (sum(ge - gs for gs, ge in gaps) / 1e6))
for n, v in acc.most_common(12): print(f"   {n:48s} {v/1e6:8.2f}")
qs = collections.Counter(q for _, _, _, q in ev); print("queues:", dict(qs))
