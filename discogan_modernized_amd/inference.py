"""Inference entry (reference inference.py:1-198): load ``gen_B_final.pth`` (A -> B) or ``gen_A_final.pth`` (B -> A),
run the generator in eval mode, optionally reconstruct with the reverse generator (:171-187).

The hot part is the eval-mode generator forward.  ``FoldedGenerator`` runs it with every BatchNorm folded into the
convolution before it: in eval mode BN is the per-channel affine map ``y*s + t`` with ``s = gamma/sqrt(running_var+eps)``,
``t = beta - running_mean*s`` (model.py:84-139 under ``.eval()``), so ``act(BN(conv(x, w))) = act(conv(x, w*s) + t)``:
the scale goes into the weights once at load time and the shift + LeakyReLU/ReLU ride in the conv kernel's epilogue
(dg_conv_fwd_bias_act / dg_conv_dgrad_bias_act).  16 kernels per pass, no normalisation pass over any activation.

Image decoding is outside the hot path: inputs are tensor files (``torch.save``d float [n,3,S,S] in [0,1] or uint8
[n,S,S,3]); image files are read through PIL when it is importable (bilinear resize; the reference uses cv2.resize,
inference.py:62, which is not installed here).  Results are written as ``<stem>_result.pt`` (dict of input / generated /
reconstructed tensors) and, when PIL is available, the reference's side-by-side PNG (:76-111).

    python -m discogan_modernized_amd.inference --model_path models/... --input_path batch.pt --image_size 64 --direction AtoB
"""
from __future__ import annotations

import argparse
from pathlib import Path

import torch

from . import model as M
from . import ops


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="HIP/MI355X implementation of DiscoGAN inference")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--model_path", type=str, required=True, help="directory with gen_A_final.pth / gen_B_final.pth")
    p.add_argument("--input_path", type=str, required=True, help="tensor file (.pt), image file, or directory of images")
    p.add_argument("--output_dir", type=str, default="./inference_results")
    p.add_argument("--image_size", type=int, default=64)
    p.add_argument("--direction", type=str, default="AtoB", choices=["AtoB", "BtoA"])
    p.add_argument("--use_extra_layers", action="store_true")
    p.add_argument("--dataset_type", type=str, default=None,
                   choices=["edges2handbags", "edges2shoes", "handbags2shoes", "celebA", None])
    p.add_argument("--domain", type=str, default=None, choices=["A", "B", None])
    p.add_argument("--no_fold", action="store_true", help="run the training modules in eval() mode instead of the folded form")
    return p.parse_args(argv)


class FoldedGenerator:
    """Eval-mode ``Generator`` with BatchNorm folded into the convolutions (forward only, no autograd)."""

    def __init__(self, gen: M.Generator):
        self.image_size = gen.image_size
        self.layers = []
        for seq in (gen.encoder, gen.decoder):
            mods = list(seq)
            i = 0
            while i < len(mods):
                conv = mods[i]
                bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], M.BatchNorm2d) else None
                j = i + (2 if bn is not None else 1)
                actm = mods[j] if j < len(mods) and isinstance(mods[j], M._Act) else None
                act = actm.act if actm is not None else ops.ACT_NONE
                slope = actm.negative_slope if actm is not None else 0.0
                self.layers.append(self._fold(conv, bn, act, slope))
                i = j + (1 if actm is not None else 0)

    @staticmethod
    @torch.no_grad()
    def _fold(conv, bn, act, slope):
        w = conv.weight.detach()
        transposed = isinstance(conv, M.ConvTranspose2d)
        edge = (conv.out_channels == 3) if transposed else (conv.in_channels == 3)
        bias = None
        if bn is not None:
            s = bn.weight.detach() * torch.rsqrt(bn.running_var.detach() + bn.eps)
            bias = (bn.bias.detach() - bn.running_mean.detach() * s).contiguous()
            # output channels: dim 0 of a Conv2d weight [K,C,4,4], dim 1 of a ConvTranspose2d weight [Cin,Cout,4,4]
            w = w * (s.view(1, -1, 1, 1) if transposed else s.view(-1, 1, 1, 1))
            w = ops.krsc_param(w.contiguous()) if not edge else w.contiguous()
        return dict(kind=("convT" if transposed else "conv"), edge=edge, w=w, bias=bias, act=act, slope=float(slope),
                    stride=conv.stride, pad=conv.padding)

    @torch.no_grad()
    def __call__(self, x):
        h = x
        for L in self.layers:
            if L["edge"] and L["kind"] == "conv":
                h = ops.c3_fwd(h, L["w"], L["act"], L["slope"])                 # conv1 + LeakyReLU (no BatchNorm)
            elif L["edge"]:
                h = ops.c3_dgrad(h, L["w"], L["act"])                             # last convT + Sigmoid
            elif L["kind"] == "conv":
                h = ops.conv_fwd_bias_act(h, L["w"], L["bias"], L["stride"], L["pad"], L["act"], L["slope"])
            else:
                hin, win = h.shape[2], h.shape[3]
                hw = ((hin - 1) * L["stride"] - 2 * L["pad"] + 4, (win - 1) * L["stride"] - 2 * L["pad"] + 4)
                h = ops.conv_dgrad_bias_act(h, L["w"], L["bias"], hw, L["stride"], L["pad"], L["act"], L["slope"])
        return h


def load_generator(model_dir, direction, image_size, device, use_extra_layers=False, fold=True, reverse=False):
    """inference.py:127-136: AtoB uses gen_B_final.pth, BtoA gen_A_final.pth (the naming trap of SURVEY Appendix B)."""
    fwd = "gen_B_final.pth" if direction == "AtoB" else "gen_A_final.pth"
    rev = "gen_A_final.pth" if direction == "AtoB" else "gen_B_final.pth"
    path = Path(model_dir) / (rev if reverse else fwd)
    if not path.exists():
        return None, path
    g = M.Generator(extra_layers=use_extra_layers, image_size=image_size).to(device)
    g.load_state_dict(torch.load(path, map_location="cpu"))
    g.eval()
    return (FoldedGenerator(g) if fold else g), path


def _load_inputs(path: Path, image_size, device):
    """[(stem, float tensor [1..n,3,S,S] on the device)]"""
    if path.suffix == ".pt":
        t = torch.load(path, map_location="cpu")
        t = ops.u8hwc_to_f32chw(t.to(device)) if t.dtype == torch.uint8 else t.float().to(device)
        return [(path.stem, t)]
    files = (sorted(path.glob("*.jpg")) + sorted(path.glob("*.png"))) if path.is_dir() else [path]
    try:
        from PIL import Image
    except ImportError as e:
        raise RuntimeError("image files need PIL; pass a tensor file (.pt) instead") from e
    import numpy as np
    out = []
    for f in files:
        img = Image.open(f).convert("RGB").resize((image_size, image_size), Image.BILINEAR)
        u8 = torch.from_numpy(np.asarray(img).copy()).unsqueeze(0)
        out.append((f.stem, ops.u8hwc_to_f32chw(u8.to(device))))
    return out


def _save(out_dir: Path, stem, inp, gen, rec):
    torch.save(dict(input=inp.cpu(), generated=gen.cpu(), reconstructed=None if rec is None else rec.cpu()), out_dir / f"{stem}_result.pt")
    try:
        from PIL import Image
    except ImportError:
        return
    panels = [inp[0], gen[0]] + ([rec[0]] if rec is not None else [])
    row = torch.cat([p.clamp(0, 1) for p in panels], dim=2)                       # side by side, like :76-111
    Image.fromarray((row.permute(1, 2, 0).cpu() * 255).round().to(torch.uint8).numpy()).save(out_dir / f"{stem}_result.png")


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: this implementation has no CPU path (use the reference for CPU runs)")
    device = torch.device("cuda", torch.cuda.current_device())
    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    gen, path = load_generator(args.model_path, args.direction, args.image_size, device, args.use_extra_layers, fold=not args.no_fold)
    if gen is None:
        print(f"model load failed: {path} not found; available:", [p.name for p in Path(args.model_path).glob('*.pth')])
        return None
    print(f"model loaded: {path}")
    rev, _ = load_generator(args.model_path, args.direction, args.image_size, device, args.use_extra_layers, fold=not args.no_fold,
                            reverse=True)
    results = []
    for stem, x in _load_inputs(Path(args.input_path), args.image_size, device):
        with torch.no_grad():
            generated = gen(x)
            reconstructed = rev(generated) if rev is not None else None           # A->B->A / B->A->B, :171-187
        _save(out_dir, stem, x, generated, reconstructed)
        results.append((stem, generated, reconstructed))
        print(f"saved: {out_dir / (stem + '_result.pt')}")
    return results


if __name__ == "__main__":
    main()
