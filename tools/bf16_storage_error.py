#!/usr/bin/env python3
"""Gradient error of the two bf16 configurations against the exact-fp32 path (same weights, same batch):
   mfma_dtype=bf16 with fp32-stored feature maps, and with bf16-stored feature maps (act_dtype=bf16).
   python tools/bf16_storage_error.py [--size 64 --batch 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    A, B = synthetic_batch(a.batch, a.size, 0, "cuda")
    res = {}
    for name, kw in (("fp32", {}), ("bf16/f32 maps", dict(mfma_dtype="bf16")), ("bf16/bf16 maps", dict(mfma_dtype="bf16", act_dtype="bf16"))):
        tr = DiscoGANTrainer(default_args(), device="cuda", image_size=a.size, seed=1234, **kw)
        l0 = tr.losses_to_floats(tr.train_iteration(A, B, 0, do_step=False))
        gd = tr.optim_dis.flat_g.clone()
        l1 = tr.losses_to_floats(tr.train_iteration(A, B, 1, do_step=False))
        gg = tr.optim_gen.flat_g.clone()
        res[name] = (l0, gd, gg, tr)
    ref = res["fp32"]
    for name in ("bf16/f32 maps", "bf16/bf16 maps"):
        r = res[name]
        print(f"== {name}")
        print("   losses it0:", {k: f"{(r[0][k] - ref[0][k]) / (abs(ref[0][k]) + 1e-12):+.2e}" for k in ref[0]})
        for i, what in ((1, "D grads"), (2, "G grads")):
            rel = ((r[i] - ref[i]).norm() / ref[i].norm()).item()
            opt = ref[3].optim_dis if i == 1 else ref[3].optim_gen
            worst = 0.0
            for p, off in zip(opt.params, opt.offsets):
                n = p.numel()
                d = (r[i][off:off + n] - ref[i][off:off + n]).norm() / (ref[i][off:off + n].norm() + 1e-30)
                worst = max(worst, d.item())
            print(f"   {what}: relative L2 {rel:.3e}, worst tensor {worst:.3e}")
    a_, b_ = res["bf16/f32 maps"], res["bf16/bf16 maps"]
    for i, what in ((1, "D grads"), (2, "G grads")):
        print(f"   between the two bf16 configurations, {what}: {((a_[i] - b_[i]).norm() / a_[i].norm()).item():.3e}")


if __name__ == "__main__":
    main()
