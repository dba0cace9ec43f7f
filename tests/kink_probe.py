"""Activation-kink bookkeeping for the gradient parity tests (test infrastructure, imports the oracle).

LeakyReLU / ReLU have a discontinuous derivative at 0.  Two correct fp32 implementations of the same BatchNorm
(PyTorch: x*alpha + (beta - mean*alpha); HIP: fma(x - mean, alpha, beta)) round the pre-activation differently by
~1e-7, so an element within rounding of 0 gets slope 1.0 in one and 0.2 (or 0) in the other.  At the tiny batches
the oracle can run, ONE such element moves whole gradient tensors by 1e-3..1e-2: that is a property of the
function being differentiated, not an arithmetic error of either side.

This module removes the ambiguity instead of widening tolerances:
  * ``record_masks_hip`` hooks the HIP modules and records the sign pattern (out > 0) of every activation the
    HIP path actually used, in call order per network;
  * ``record_masks_oracle`` does the same for an oracle (PyTorch) network;
  * ``masked_copy`` builds an fp64 copy of an oracle network whose LeakyReLU / ReLU take their derivative
    pattern from a recorded mask list, i.e. the SAME piecewise-linear function the implementation under test
    differentiated.  Gradients of that copy are the ground truth for that implementation with no kink term left.
``python -m tests.kink_probe S N`` prints the per-tensor table (GPU box).
"""
from __future__ import annotations

import copy
import sys

import torch
import torch.nn as nn


class MaskedAct(nn.Module):
    """LeakyReLU(slope) / ReLU (slope 0) whose branch per element comes from a recorded mask queue."""

    def __init__(self, slope, queue, counter):
        super().__init__()
        self.slope, self.queue, self.counter = float(slope), queue, counter

    def forward(self, u):
        m = self.queue[self.counter[0]]
        self.counter[0] += 1
        assert m.shape == u.shape, (tuple(m.shape), tuple(u.shape))
        self.last_flips = int(((u > 0) != m).sum())
        return torch.where(m, u, u * self.slope)


def _is_kink_act(m):
    return isinstance(m, (nn.LeakyReLU, nn.ReLU))


def masked_copy(net, masks, dtype=torch.float64):
    """Deep copy of an oracle net in ``dtype`` with every LeakyReLU/ReLU replaced by MaskedAct reading ``masks``
    (list of bool tensors in activation-call order over ALL forward calls of this net in the iteration)."""
    n2 = copy.deepcopy(net).to(dtype)
    counter = [0]
    for parent in list(n2.modules()):
        for name, child in list(parent.named_children()):
            if _is_kink_act(child):
                slope = child.negative_slope if isinstance(child, nn.LeakyReLU) else 0.0
                setattr(parent, name, MaskedAct(slope, masks, counter))
    n2._mask_counter = counter
    return n2


def flips_of(net):
    return sum(getattr(m, "last_flips", 0) for m in net.modules() if isinstance(m, MaskedAct))


class record_masks_oracle:
    """Context manager: masks[name] = [out > 0 for every LeakyReLU/ReLU call of nets[name], in call order]."""

    def __init__(self, nets):
        self.nets, self.masks, self.handles = nets, {k: [] for k in nets}, []

    def __enter__(self):
        for k, net in self.nets.items():
            for m in net.modules():
                if _is_kink_act(m):
                    self.handles.append(m.register_forward_hook(
                        lambda mod, inp, out, k=k: self.masks[k].append((out.detach() > 0).clone())))
        return self.masks

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        return False


class record_masks_hip:
    """Same for the HIP modules: the fused activations are outputs of the first Conv2d (3 input channels, fused
    LeakyReLU) and of every BatchNorm2d (fused LeakyReLU / ReLU) -- model._run_fused_steps / Discriminator.forward_steps."""

    def __init__(self, nets):
        self.nets, self.masks, self.handles = nets, {k: [] for k in nets}, []

    def __enter__(self):
        from discogan_modernized_amd import model as M
        from discogan_modernized_amd import ops
        # the hooks read the fp32 outputs of the BatchNorm groups: keep them written while recording (f32x3 plane path)
        self._po, ops.X3_PLANES_ONLY = ops.X3_PLANES_ONLY, False
        for k, net in self.nets.items():
            for m in net.modules():
                if isinstance(m, M.BatchNorm2d) or (isinstance(m, M.Conv2d) and m.in_channels == 3):
                    self.handles.append(m.register_forward_hook(
                        lambda mod, inp, out, k=k: self.masks[k].append((out.detach() > 0).cpu().contiguous())))
        return self.masks

    def __exit__(self, *exc):
        from discogan_modernized_amd import ops
        ops.X3_PLANES_ONLY = self._po
        for h in self.handles:
            h.remove()
        return False


def masked_state(st, masks, dtype=torch.float64):
    """fp64 copy of an oracle training state whose nets differentiate the recorded activation pattern."""
    from types import SimpleNamespace
    s = SimpleNamespace(**{k: v for k, v in vars(st).items() if k not in ("nets", "optim_gen", "optim_dis")})
    s.nets = {k: masked_copy(st.nets[k], masks[k], dtype) for k in st.nets}
    s.generator_A, s.generator_B = s.nets["gen_A"], s.nets["gen_B"]
    s.discriminator_A, s.discriminator_B = s.nets["dis_A"], s.nets["dis_B"]
    s.optim_gen = s.optim_dis = None
    return s


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def run_masked64(O, st, masks, A, B, it):
    """One oracle iteration (no optimiser step) in fp64 on the masked copy; returns the masked state."""
    s = masked_state(st, masks)
    torch.set_default_dtype(torch.float64)
    try:
        O.train_iteration(s, A.double(), B.double(), it, do_step=False)
    finally:
        torch.set_default_dtype(torch.float32)
    for k, n in s.nets.items():
        assert n._mask_counter[0] == len(masks[k]), f"{k}: used {n._mask_counter[0]} of {len(masks[k])} recorded masks"
    return s


def main(S, N, iters=3):
    from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args
    from oracle import discogan_ref as O
    st = O.build_state(image_size=S, seed=1234)
    tr = DiscoGANTrainer(default_args(), device="cuda", image_size=S, seed=1234)
    A, B = O.synthetic_batch(N, S, seed=0)
    Ag, Bg = A.cuda(), B.cuda()
    for it in range(iters):
        for k in st.nets:
            tr.nets[k].load_state_dict(st.nets[k].state_dict())
        # plain fp64 oracle (its own natural masks)
        s64 = copy.deepcopy(st)
        for n in s64.nets.values():
            n.double()
        torch.set_default_dtype(torch.float64)
        try:
            with record_masks_oracle(s64.nets) as m64:
                O.train_iteration(s64, A.double(), B.double(), it, do_step=False)
        finally:
            torch.set_default_dtype(torch.float32)
        with record_masks_oracle(st.nets) as m32:
            O.train_iteration(st, A, B, it, do_step=False)
        with record_masks_hip(tr.nets) as mh:
            tr.train_iteration(Ag, Bg, it, do_step=False)
        torch.cuda.synchronize()
        s_h = run_masked64(O, st, mh, A, B, it)       # truth for the HIP path's piecewise-linear function
        s_o = run_masked64(O, st, m32, A, B, it)      # truth for the fp32 oracle's
        dstep = O.is_dis_step(it, st.args)
        live = ("dis_A", "dis_B") if dstep else ("gen_A", "gen_B")
        for k in st.nets:
            fl_h = sum(int((a != b).sum()) for a, b in zip(mh[k], m64[k]))
            fl_o = sum(int((a != b).sum()) for a, b in zip(m32[k], m64[k]))
            tot = sum(a.numel() for a in m64[k])
            print(f"iter {it} {k}: activation sign flips vs fp64: hip {fl_h}, oracle-fp32 {fl_o} of {tot}")
        print(f"iter {it} ({'D' if dstep else 'G'}-step)  {'tensor':34s} {'hip/plain64':>11s} {'hip/masked':>11s} {'ref/plain64':>11s} {'ref/masked':>11s}")
        worst = [0.0, 0.0, 0.0, 0.0]
        for name in live:
            P = [dict(n.named_parameters()) for n in (tr.nets[name], st.nets[name], s64.nets[name], s_h.nets[name], s_o.nets[name])]
            for pn in P[1]:
                gh, go, g64, gmh, gmo = (p[pn].grad for p in P)
                row = (rel_err(gh, g64), rel_err(gh, gmh), rel_err(go, g64), rel_err(go, gmo))
                worst = [max(a, b) for a, b in zip(worst, row)]
                print(f"   {name + '.' + pn:40s} {row[0]:11.2e} {row[1]:11.2e} {row[2]:11.2e} {row[3]:11.2e}")
        print(f"   {'WORST':40s} {worst[0]:11.2e} {worst[1]:11.2e} {worst[2]:11.2e} {worst[3]:11.2e}", flush=True)
        (st.optim_dis if dstep else st.optim_gen).step()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3)
