#!/usr/bin/env python3
"""How much does the REFERENCE's own fp32 gradient error move with the summation order?  (CPU only)

The per-tensor bound of the parity tests is ``4 x the reference's own fp32 error`` against an fp64 run on the same activation
pattern.  At 512 px / batch 2 the HIP path sits at 2-3.5x of that error (tools/err_ratio_512.py), with the SAME ratio on every
tensor upstream of a small-batch BatchNorm.  This tool measures the denominator's own spread: the oracle (= the reference's
arithmetic, oneDNN) is run in fp32 with 1, 2, 4 and 8 threads -- oneDNN blocks and orders its reductions by thread count -- and
each run is compared with the fp64 oracle on that run's own activation pattern.  If the reference's error moves by a factor F
between two of ITS OWN summation orders, an independent implementation cannot be expected inside a ratio much tighter than F.

    python tools/ref_noise_spread.py --size 512 --batch 2 --out profiles/r03_reference_noise_spread_512px.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import discogan_ref as O  # noqa: E402
from tests import kink_probe as KP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--threads", default="1,2,4,8")
    ap.add_argument("--iters", default="0,1")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    st = O.build_state(image_size=a.size, seed=1234)
    A, B = O.synthetic_batch(a.batch, a.size, seed=0)
    doc = {}
    for it in [int(x) for x in a.iters.split(",")]:
        dstep = O.is_dis_step(it, st.args)
        live = ("dis_A", "dis_B") if dstep else ("gen_A", "gen_B")
        per_thread = {}
        for nt in [int(x) for x in a.threads.split(",")]:
            torch.set_num_threads(nt)
            t0 = time.time()
            for net in st.nets.values():
                net.zero_grad()
            with KP.record_masks_oracle(st.nets) as m32:
                O.train_iteration(st, A, B, it, do_step=False)
            g32 = {f"{n}.{pn}": p.grad.detach().clone() for n in live for pn, p in st.nets[n].named_parameters()}
            # reset BatchNorm buffers? train_iteration with do_step=False still updates running stats: irrelevant for gradients (train mode)
            torch.set_num_threads(max(8, nt))
            s64 = KP.run_masked64(O, st, m32, A, B, it)
            rows = {}
            for n in live:
                t64 = dict(s64.nets[n].named_parameters())
                for pn, _ in st.nets[n].named_parameters():
                    rows[f"{n}.{pn}"] = KP.rel_err(g32[f"{n}.{pn}"], t64[pn].grad)
            per_thread[nt] = rows
            print(f"iter {it} threads {nt}: worst {max(rows.values()):.2e}, median {sorted(rows.values())[len(rows) // 2]:.2e} ({time.time() - t0:.0f} s)", flush=True)
            del s64
        names = list(next(iter(per_thread.values())))
        spread = {k: max(per_thread[t][k] for t in per_thread) / max(min(per_thread[t][k] for t in per_thread), 1e-30) for k in names}
        worst = sorted(spread.items(), key=lambda kv: -kv[1])[:8]
        print(f"iter {it} ({'D' if dstep else 'G'}-step): max/min of the reference's own error over thread counts: "
              f"worst tensor {worst[0][0]} x{worst[0][1]:.2f}, median x{sorted(spread.values())[len(spread) // 2]:.2f}", flush=True)
        doc[str(it)] = dict(step="D" if dstep else "G", err_by_threads={str(t): per_thread[t] for t in per_thread},
                            spread_max_over_min=spread, worst=worst)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(dict(image_size=a.size, batch=a.batch, note="relative L2 of the oracle's fp32 gradients vs the fp64 oracle on the same "
                       "activation pattern, per thread count", results=doc), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
