#!/usr/bin/env python3
"""Generate the golden fixtures that pin the oracle (and through it the HIP path).

Run in the AUTHORING container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--skip-ref] [--skip-oracle]

Outputs (committed, small JSON -- data only, no reference source):
  tests/golden/ref_s512_n2.json     outputs of the TRUE reference: /root/reference/model.py nets +
                                    image_translation.get_gan_loss/get_fm_loss, driven through the
                                    loop body image_translation.py:336-390 for iterations 0,1,2
                                    (D,G,G) on synthetic tensors (seed 1234 models, seed 0 data, N=2).
  tests/golden/ref_s512_n2_gstep.json   TRUE reference, ONE G-step taken from the seeded init (loop body driven
                                    with iters=1 on fresh seed-1234 weights, N=2): generator gradient norms /
                                    samples at the reference's only size without a preceding D update.
  tests/golden/ref_s512_n32_dstep.json / ref_s512_n32_gstep.json
                                    TRUE reference at BASELINE configs[3]'s own batch (N=32): iteration 0 (D-step)
                                    and a G-step from the seeded init; ~35 GB RSS, minutes of CPU (--n32).
  tests/golden/oracle_s64_n4.json   same capture from oracle/discogan_ref.py for the derived 64 px
                                    network (the reference cannot run at 64 px).
  tests/golden/oracle_s16_n4.json   tiny 16 px variant (2 stride-2 stages) used by fast GPU tests.

The reference needs cv2/torchvision only for image-file code (dataset.py:2,11); two empty stub
modules make ``image_translation`` importable.
"""
import argparse
import json
import os
import sys
import time
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def sample_idx(numel, k=8):
    return [int((i * 2654435761) % numel) for i in range(1, k + 1)]


def tensor_digest(t):
    f = t.detach().reshape(-1)
    d = f.double()
    return dict(shape=list(t.shape), sum=float(d.sum()), abssum=float(d.abs().sum()),
                samples=[float(f[i]) for i in sample_idx(f.numel())])


def capture_run(nets, optim_gen, optim_dis, crit, gan_fn, fm_fn, A, B, n_iters, args, iter_list=None, lean=False):
    """Drives image_translation.py:336-390 and records everything the parity tests compare.
    iter_list: the iteration indices to run (default 0..n_iters-1); lean: skip the per-feature digests."""
    rec = dict(init={}, iters=[])
    for name, net in nets.items():
        rec["init"][name] = {k: tensor_digest(v) for k, v in net.state_dict().items()
                             if v.dtype.is_floating_point}
    gA, gB, dA, dB = nets["gen_A"], nets["gen_B"], nets["dis_A"], nets["dis_B"]
    for iters in (iter_list if iter_list is not None else range(n_iters)):
        t0 = time.time()
        for net in nets.values():
            net.zero_grad()
        AB = gB(A)
        BA = gA(B)
        ABA = gA(AB)
        BAB = gB(BA)
        recon_loss_A = crit["recon"](ABA, A)
        recon_loss_B = crit["recon"](BAB, B)
        A_dis_real, A_feats_real = dA(A)
        A_dis_fake, A_feats_fake = dA(BA)
        dis_loss_A, gen_loss_A = gan_fn(A_dis_real, A_dis_fake, crit["gan"], "cpu")
        fm_loss_A = fm_fn(A_feats_real, A_feats_fake, crit["feat"], "cpu")
        B_dis_real, B_feats_real = dB(B)
        B_dis_fake, B_feats_fake = dB(AB)
        dis_loss_B, gen_loss_B = gan_fn(B_dis_real, B_dis_fake, crit["gan"], "cpu")
        fm_loss_B = fm_fn(B_feats_real, B_feats_fake, crit["feat"], "cpu")
        rate = args["starting_rate"] if iters < args["gan_curriculum"] else args["default_rate"]
        gen_loss_A_total = (fm_loss_B * 0.9 + gen_loss_B * 0.1) * (1 - rate) + recon_loss_A * rate
        gen_loss_B_total = (fm_loss_A * 0.9 + gen_loss_A * 0.1) * (1 - rate) + recon_loss_B * rate
        gen_loss = gen_loss_A_total + gen_loss_B_total
        dis_loss = dis_loss_A + dis_loss_B
        t_fwd = time.time() - t0
        dstep = iters % args["update_interval"] == 0
        if dstep:
            dis_loss.backward()
        else:
            gen_loss.backward()
        t_bwd = time.time() - t0 - t_fwd
        it = dict(iter=iters, step="D" if dstep else "G",
                  losses=dict(gen_loss_A=float(gen_loss_A), gen_loss_B=float(gen_loss_B),
                              fm_loss_A=float(fm_loss_A), fm_loss_B=float(fm_loss_B),
                              recon_loss_A=float(recon_loss_A), recon_loss_B=float(recon_loss_B),
                              dis_loss_A=float(dis_loss_A), dis_loss_B=float(dis_loss_B),
                              gen_loss=float(gen_loss), dis_loss=float(dis_loss)),
                  dis_out=dict(A_real=A_dis_real.detach().reshape(-1).tolist(),
                               A_fake=A_dis_fake.detach().reshape(-1).tolist(),
                               B_real=B_dis_real.detach().reshape(-1).tolist(),
                               B_fake=B_dis_fake.detach().reshape(-1).tolist()),
                  outputs=dict(AB=tensor_digest(AB), BA=tensor_digest(BA),
                               ABA=tensor_digest(ABA), BAB=tensor_digest(BAB)),
                  feats=({} if lean else dict(A_real=[tensor_digest(f) for f in A_feats_real],
                                              B_fake=[tensor_digest(f) for f in B_feats_fake])),
                  grad_norms={}, grad_samples={}, after_step={}, buffers={})
        live = ("dis_A", "dis_B") if dstep else ("gen_A", "gen_B")
        for name in live:                      # only the stepped side's grads are results (F5)
            net = nets[name]
            it["grad_norms"][name] = {k: float(p.grad.double().norm()) for k, p in net.named_parameters()}
            it["grad_samples"][name] = {k: [float(p.grad.reshape(-1)[i]) for i in sample_idx(p.numel())]
                                        for k, p in net.named_parameters()}
        (optim_dis if dstep else optim_gen).step()
        for name in live:
            net = nets[name]
            it["after_step"][name] = {k: [float(p.detach().reshape(-1)[i]) for i in sample_idx(p.numel())]
                                      for k, p in net.named_parameters()}
        for name, net in nets.items():
            it["buffers"][name] = {k: (int(b) if b.dtype == torch.int64 else tensor_digest(b))
                                   for k, b in net.named_buffers()}
        it["time_s"] = dict(fwd=t_fwd, bwd=t_bwd, total=time.time() - t0)
        print(f"  iter {iters} {it['step']}-step  {it['time_s']['total']:.1f}s  "
              f"GEN {it['losses']['gen_loss_A']:.6f}/{it['losses']['gen_loss_B']:.6f} "
              f"DIS {it['losses']['dis_loss_A']:.6f}/{it['losses']['dis_loss_B']:.6f}", flush=True)
        rec["iters"].append(it)
    return rec


ARGS = dict(learning_rate=2e-4, beta1=0.5, beta2=0.999, weight_decay=0.00001, gan_curriculum=10000,
            starting_rate=0.01, default_rate=0.5, update_interval=3)


def run_reference(n=2, n_iters=3, iter_list=None):
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].Compose = object
    sys.path.insert(0, "/root/reference")
    import model as ref_model                       # /root/reference/model.py
    import image_translation as ref_it              # /root/reference/image_translation.py
    import torch.nn as nn
    import torch.optim as optim
    from itertools import chain
    torch.manual_seed(1234)                          # distributed_image_translation.py:372
    nets = dict(gen_A=ref_model.Generator(extra_layers=True), gen_B=ref_model.Generator(extra_layers=True),
                dis_A=ref_model.Discriminator(), dis_B=ref_model.Discriminator())
    crit = dict(recon=nn.MSELoss(), gan=nn.BCELoss(), feat=nn.HingeEmbeddingLoss())
    og = optim.Adam(chain(nets["gen_A"].parameters(), nets["gen_B"].parameters()), lr=ARGS["learning_rate"],
                    betas=(ARGS["beta1"], ARGS["beta2"]), weight_decay=ARGS["weight_decay"])
    od = optim.Adam(chain(nets["dis_A"].parameters(), nets["dis_B"].parameters()), lr=ARGS["learning_rate"],
                    betas=(ARGS["beta1"], ARGS["beta2"]), weight_decay=ARGS["weight_decay"])
    g = torch.Generator().manual_seed(0)
    A = torch.rand(n, 3, 512, 512, generator=g)
    B = torch.rand(n, 3, 512, 512, generator=g)
    rec = capture_run(nets, og, od, crit, ref_it.get_gan_loss, ref_it.get_fm_loss, A, B, n_iters, ARGS,
                      iter_list=iter_list, lean=n > 2)
    if n > 2:                      # the init digests are those of ref_s512_n2.json (same seed): keep the file small
        rec["init"] = {}
    rec["meta"] = dict(source="reference /root/reference model.py + image_translation.py loop body :336-390",
                       iter_list=list(iter_list) if iter_list is not None else list(range(n_iters)),
                       image_size=512, n=n, model_seed=1234, data_seed=0, torch=torch.__version__,
                       threads=torch.get_num_threads(), args=ARGS,
                       state_dict_keys={k: list(v.state_dict().keys()) for k, v in nets.items()})
    return rec


def run_oracle(image_size, n, n_iters=3):
    from oracle import discogan_ref as O
    st = O.build_state(image_size=image_size, seed=1234)
    A, B = O.synthetic_batch(n, image_size, seed=0)
    crit = dict(recon=st.recon_criterion, gan=st.gan_criterion, feat=st.feat_criterion)
    rec = capture_run(st.nets, st.optim_gen, st.optim_dis, crit, O.get_gan_loss, O.get_fm_loss,
                      A, B, n_iters, ARGS)
    rec["meta"] = dict(source="oracle/discogan_ref.py (depth-rule net; no reference execution exists)",
                       image_size=image_size, n=n, model_seed=1234, data_seed=0, torch=torch.__version__,
                       threads=torch.get_num_threads(), args=ARGS,
                       state_dict_keys={k: list(v.state_dict().keys()) for k, v in st.nets.items()})
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-ref", action="store_true")
    ap.add_argument("--skip-oracle", action="store_true")
    ap.add_argument("--gstep", action="store_true", help="only: reference G-step from the seeded init, N=2")
    ap.add_argument("--n32", choices=["dstep", "gstep"], default=None,
                    help="only: reference at N=32 (BASELINE configs[3] batch), iteration 0 or a G-step from init")
    a = ap.parse_args()
    if a.gstep or a.n32:
        n = 32 if a.n32 else 2
        kind = a.n32 or "gstep"
        print(f"reference @512 N={n} {kind} from the seeded init ...", flush=True)
        rec = run_reference(n=n, iter_list=[0] if kind == "dstep" else [1])
        name = f"ref_s512_n{n}_{kind}.json"
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(rec, f)
        print("wrote", name, flush=True)
        return
    if not a.skip_ref:
        print("reference @512 N=2 ...", flush=True)
        rec = run_reference()
        with open(os.path.join(HERE, "ref_s512_n2.json"), "w") as f:
            json.dump(rec, f)
    if not a.skip_oracle:
        for s, n in ((64, 4), (16, 4)):
            print(f"oracle @{s} N={n} ...", flush=True)
            rec = run_oracle(s, n)
            with open(os.path.join(HERE, f"oracle_s{s}_n{n}.json"), "w") as f:
                json.dump(rec, f)


if __name__ == "__main__":
    main()
