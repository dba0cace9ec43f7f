#!/usr/bin/env python3
"""Which gradient tensors carry the 3.3-3.5x error ratio at 512 px, and what in the kernel causes it?  

tests/test_model_gpu.py::test_masked_fp64_gradient_parity_512 compares every gradient tensor of a D-step and a G-step from the
seeded init (512 px, batch 2) with an fp64 run of the oracle on the implementation's own activation pattern, and the reference's
fp32 gradients with fp64 on ITS pattern; the ratio of the two errors is 0.8-1.4 up to 128 px and 3.3 / 3.5 at 512 px.  This tool
prints the ratio PER TENSOR and repeats the run with library variants that change one suspect at a time:

    fp32            the exact-fp32 MFMA path as shipped (32x32x2: one sequential fp32 chain per output element, up to 32768 long)
    fp32_splitkN    the same kernels with every conv reduction cut into N slabs summed in a fixed order (option "splitk" N): other
                    summation orders of the SAME arithmetic -- if the ratios scatter without a trend in N, they are draws of the
                    rounding noise through the small-batch BatchNorms, not a property of one order
    f32x3           plane path (exact bf16 products, the same sequential fp32 accumulation order per tile)

    python tools/err_ratio_512.py --out gpurun_out/err_ratio_512.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from discogan_modernized_amd import _lib  # noqa: E402
from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args  # noqa: E402
from oracle import discogan_ref as O  # noqa: E402  (checker)
from tests import test_model_gpu as T  # noqa: E402  (the comparison itself lives in the test module)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--variants", default="fp32,fp32_splitk4,f32x3")
    ap.add_argument("--teacher_forced_steps", type=int, default=0,
                    help="instead of the D-step / G-step from the seeded init: teacher-forced iterations 0..N-1 WITH the Adam steps "
                         "(every iteration starts from the oracle's weights), i.e. iterations 2-3 run in the post-update, saturated-"
                         "discriminator regime (one variant only)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    st = O.build_state(image_size=a.size, seed=1234)
    cache, doc = {}, {}
    for v in a.variants.split(","):
        mf = "f32x3" if v == "f32x3" else "f32"
        tr = DiscoGANTrainer(default_args(), device="cuda", image_size=a.size, seed=1234, mfma_dtype=mf)
        if v.startswith("fp32_splitk"):            # fp32_splitkN: every conv reduction in N slabs, summed in a fixed order
            _lib.set_option("splitk", int(v[len("fp32_splitk"):]))
        rows = []
        try:
            if a.teacher_forced_steps:
                T._teacher_forced(a.size, a.batch, a.teacher_forced_steps, tr=tr, st=st, step=True, rows_out=rows, mfma_dtype=mf)
            else:
                T._teacher_forced(a.size, a.batch, 0, tr=tr, st=st, iter_list=[0, 1], step=False, rows_out=rows, noise_cache=cache, mfma_dtype=mf)
        except AssertionError as e:
            print(f"[{v}] bound exceeded: {e}", flush=True)
        finally:
            _lib.set_option("splitk", 0)
        tr.close()
        del tr
        torch.cuda.empty_cache()
        doc[v] = rows
        for it in sorted({x["iter"] for x in rows}):
            r = sorted([x for x in rows if x["iter"] == it], key=lambda x: -x["ratio"])
            big = [x for x in r if x["ratio"] > 2]
            print(f"[{v}] iter {it}: worst ratio {r[0]['ratio']:.2f} ({r[0]['tensor']}), {len(big)} of {len(r)} tensors above 2; "
                  f"worst err {max(x['err_hip'] for x in r):.2e} (reference fp32 {max(x['err_reference_fp32'] for x in r):.2e})", flush=True)
            for x in r[:10]:
                print(f"      {x['tensor']:34s} n={x['numel']:>9d} hip {x['err_hip']:.2e}  ref {x['err_reference_fp32']:.2e}  ratio {x['ratio']:.2f}", flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
