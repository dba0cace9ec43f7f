#!/usr/bin/env python3
"""Does training on the bf16 matrix path track exact fp32 beyond the first iterations?  (missing #6)

Runs the same seeded DiscoGAN (64 px, batch 64 -- BASELINE configs[0]'s shape) for N iterations on the exact-fp32 path, on
bf16 operands with fp32-stored feature maps and on configs[4]'s full arithmetic (bf16 operands + bf16-stored feature maps), with
a FRESH synthetic batch per iteration drawn from one seeded generator (identical across the runs).  Single iterations of a GAN
are chaotic after the first discriminator update, so the comparison is over windowed means of the logged losses
(image_translation.py:394-398): reconstruction (RECON A+B), feature matching (FM A+B), discriminator (DIS A+B).

    python tools/bf16_trajectory.py --iters 300 --out gpurun_out/bf16_trajectory.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args  # noqa: E402

CONFIGS = {"fp32": {}, "bf16_f32maps": dict(mfma_dtype="bf16"), "bf16_bf16maps": dict(mfma_dtype="bf16", act_dtype="bf16")}
# windowed-mean ratio bf16 / fp32 that tests/test_model_gpu.py::test_bf16_training_trajectory_tracks_fp32_short enforces
BANDS = {"recon": (0.90, 1.10), "fm": (0.60, 1.60)}


def _grad_cosines(tr, probe, A, B, it):
    """Per-tensor cosine between the gradient the run's arithmetic computes and the exact-fp32 gradient AT THE SAME WEIGHTS, BatchNorm
    buffers and batch (no optimiser step; the run's buffers are restored afterwards).  Returns {tensor name: cosine} of the stepped side."""
    bufs = {n: {k: b.detach().clone() for k, b in net.named_buffers()} for n, net in tr.nets.items()}
    for n, net in tr.nets.items():
        probe.nets[n].load_state_dict(net.state_dict())
    tr.train_iteration(A, B, it, do_step=False)
    probe.train_iteration(A, B, it, do_step=False)
    torch.cuda.synchronize()
    live = ("dis_A", "dis_B") if tr.is_dis_step(it) else ("gen_A", "gen_B")
    out = {}
    for n in live:
        for (pn, p), (_, q) in zip(tr.nets[n].named_parameters(), probe.nets[n].named_parameters()):
            a, b = p.grad.detach().double().reshape(-1), q.grad.detach().double().reshape(-1)
            out[f"{n}.{pn}"] = round(float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300)), 5)
    with torch.no_grad():
        for n, net in tr.nets.items():
            for k, b in net.named_buffers():
                b.copy_(bufs[n][k])
    return out


def run(iters=300, size=64, batch=64, configs=tuple(CONFIGS), dev="cuda", data_pool=16, probe_iters=()):
    g = torch.Generator().manual_seed(4321)
    # a pool of smooth synthetic "image" batches (low-frequency patterns, not white noise, so that reconstruction can improve)
    pool = []
    for _ in range(data_pool):
        lo = torch.rand(2, batch, 3, size // 8, size // 8, generator=g)
        up = torch.nn.functional.interpolate(lo.reshape(2 * batch, 3, size // 8, size // 8), size=(size, size), mode="bilinear", align_corners=False)
        pool.append(up.reshape(2, batch, 3, size, size).clamp(0, 1).to(dev))
    out = {}
    cosines = {}
    probe = DiscoGANTrainer(default_args(), device=dev, image_size=size, seed=1234) if probe_iters else None
    for name in configs:
        tr = DiscoGANTrainer(default_args(), device=dev, image_size=size, seed=1234, use_graph=True, **CONFIGS[name])
        rows = []
        for it in range(iters):
            A, B = pool[it % data_pool]
            if probe is not None and it in probe_iters and name != "fp32":
                cosines.setdefault(name, {})[str(it)] = _grad_cosines(tr, probe, A, B, it)
            f = tr.losses_to_floats(tr.train_iteration(A, B, it))
            rows.append([f["recon_loss_A"] + f["recon_loss_B"], f["fm_loss_A"] + f["fm_loss_B"], f["dis_loss_A"] + f["dis_loss_B"],
                         f["gen_loss_A"] + f["gen_loss_B"]])
        tr.finish()
        torch.cuda.synchronize()
        tr.close()
        out[name] = rows
    if probe is not None:
        probe.close()
        out["_grad_cosine_vs_fp32"] = cosines
    return out


def compare(res, window=30):
    ref = torch.tensor(res["fp32"], dtype=torch.float64)
    nwin = ref.shape[0] // window
    bands = {}
    for name, rows in res.items():
        t = torch.tensor(rows, dtype=torch.float64)
        assert torch.isfinite(t).all(), f"{name}: non-finite loss"
        r = {}
        for col, key in enumerate(("recon", "fm", "dis", "gen")):
            a = t[:nwin * window, col].reshape(nwin, window).mean(1)
            b = ref[:nwin * window, col].reshape(nwin, window).mean(1)
            r[key] = [round(float(x), 4) for x in (a / b.clamp_min(1e-12))]
            r[key + "_abs"] = [round(float(x), 5) for x in a]
        bands[name] = r
    return bands


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--window", type=int, default=30)
    ap.add_argument("--out", default=None)
    ap.add_argument("--data_pool", type=int, default=16)
    ap.add_argument("--probe_iters", default="", help="comma list: at these iterations also log the per-tensor cosine between the run's gradient "
                    "and the exact-fp32 gradient at the same weights / batch (stepped side)")
    a = ap.parse_args()
    probes = tuple(int(x) for x in a.probe_iters.split(",") if x)
    res = run(a.iters, a.size, a.batch, data_pool=a.data_pool, probe_iters=probes)
    cos = res.pop("_grad_cosine_vs_fp32", None)
    bands = compare(res, a.window)
    doc = dict(iters=a.iters, image_size=a.size, batch=a.batch, window=a.window,
               note="ratios = windowed mean of the run / windowed mean of the fp32 run, per window; *_abs = the run's own windowed means",
               bands=bands)
    if cos:
        summ = {}
        for name, per_it in cos.items():
            summ[name] = {it: dict(min=min(v.values()), median=sorted(v.values())[len(v) // 2], worst=min(v, key=v.get), tensors=len(v))
                          for it, v in per_it.items()}
        doc["grad_cosine_vs_fp32"] = dict(note="cosine(gradient of the run's arithmetic, exact-fp32 gradient) at the SAME weights, BatchNorm "
                                               "buffers and batch, per parameter tensor of the stepped side", summary=summ, per_tensor=cos)
    print(json.dumps({k: v for k, v in doc.items() if k != "grad_cosine_vs_fp32"}, indent=1))
    if cos:
        print(json.dumps(doc["grad_cosine_vs_fp32"]["summary"], indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(dict(doc, raw=res), open(a.out, "w"))


if __name__ == "__main__":
    main()
