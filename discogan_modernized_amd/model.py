"""Drop-in ``Generator`` / ``Discriminator`` (reference model.py:5-225) on the HIP kernels.

Same constructor signatures, attribute names (``conv1..conv8``, ``bn2..bn7``, ``encoder``,
``decoder``, ``main``), parameter order, state_dict keys / logical shapes and return types as the
reference, so checkpoints (``gen_A_*.pth`` ...) interoperate.  The optional
``image_size`` argument applies the depth rule: the default 512 IS the
reference network; 64 gives the original DiscoGAN 64 px network the reference CLI defaults to
but cannot build (model.py:8-35 is hard-wired to 512).

Layer modules hold the parameters; ``forward`` fuses each [conv, BatchNorm, activation] group into
kernel calls.  Interior feature maps are logical NCHW tensors with NHWC memory; the image side is
plain NCHW as in the reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F
from . import ops


# BatchNorm statistics can come out of the producing conv kernel's epilogue (dg_conv_*_bnstats) instead of a
# separate read pass.  Measured on MI355X: with the two-stream schedule the separate HBM-bound statistics pass overlaps the
# other chain's MFMA kernels almost for free, while the fused form lengthens the MFMA kernels' tails and needs its partial rows
# merged: 15.21 vs 14.95 ms/step at 64 px / batch 256 (round 1, exact fp32: 140 us kernels) -- but 164.9 vs 165.6 ms at 512 px /
# batch 32 (round 3, same-box A/B, alternating: 194.0 / 194.0 against 193.2 / 193.1 images/s; 800 us kernels).
#   "auto"  : exact-fp32 path: fused where the conv launch is at least 40 GFLOP (every interior layer at 512 px / batch 32 is 69 or
#             137; the 64 px / batch 256 layers are 14) -- the default
#   False   : always the separate statistics pass
#   "split" : fused only for layers whose conv plan uses split-K -- the statistics then come out of the
#             split-K reduction kernel (measured 15.14 ms/step at 64 px: still slower than the separate pass)
#   True    : fused everywhere
FUSE_BN_STATS = {"auto": "auto", "0": False, "1": True, "split": "split"}[__import__("os").environ.get("DG_FUSE_BN", "auto")]
FUSE_BN_MIN_FLOP = 4e10


def _fuse_stats(conv=None, x=None):
    """Statistics from the conv kernels on the paths other than f32x3 planes?  bf16 matrix path (shadow operands / bf16-stored
    feature maps): yes since round 3 (ops.FUSE_STATS16, DG_FUSE_BN16=0 switches it off) -- the bf16 kernels run 100-200 us per
    launch at 512 px and the statistics pass is 3.7 % of that step; exact-fp32 path: DG_FUSE_BN, by default by the size of the launch."""
    if ops.SHADOW or ops.ACT16:
        return True if ops.FUSE_STATS16 else False
    if FUSE_BN_STATS != "auto":
        return FUSE_BN_STATS
    if conv is None or x is None:
        return False
    n, _, h, w = x.shape
    pixels = h * w if isinstance(conv, ConvTranspose2d) else (h * w) // 4        # GEMM rows per image (x 4 parity classes folded in)
    return 2.0 * n * pixels * 16 * conv.in_channels * conv.out_channels >= FUSE_BN_MIN_FLOP


def stage_channels(image_size: int):
    n = int(round(math.log2(image_size))) - 2
    if n < 1 or 2 ** (n + 2) != image_size:
        raise ValueError(f"image_size must be a power of two >= 8, got {image_size}")
    return [min(64 * 2 ** i, 2048) for i in range(n)]


def _init_weight(shape):
    """Same RNG consumption and values as nn.Conv2d / nn.ConvTranspose2d.reset_parameters
    (kaiming_uniform_(a=sqrt(5)) on the logical [d0,d1,4,4] tensor)."""
    w = torch.empty(shape)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    return w


class _FlatGradMixin:
    """Keeps ``.grad`` views of a flat gradient buffer alive across ``zero_grad`` (see optim.Adam)."""

    def zero_grad(self, set_to_none: bool = True):  # noqa: D401
        for p in self.parameters():
            flat = getattr(p, "_dg_flat_grad", None)
            if flat is not None:
                flat.zero_()
                p.grad = flat
            elif p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.zero_()


class Conv2d(nn.Module):
    """nn.Conv2d(in,out,4,stride,pad,bias=False) (model.py:8,11,...,35)."""

    def __init__(self, in_channels, out_channels, kernel_size=4, stride=2, padding=1, bias=False):
        super().__init__()
        if kernel_size != 4 or bias or (stride, padding) not in ((2, 1), (1, 0)):
            raise ValueError("only Conv2d(k=4, (s,p) in {(2,1),(1,0)}, bias=False) exists in the DiscoGAN path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.stride, self.padding = stride, padding
        w = _init_weight((out_channels, in_channels, 4, 4))
        if in_channels != 3:
            w = ops.krsc_param(w)  # memory [K][4][4][C], logical shape unchanged
        self.weight = nn.Parameter(w)

    def forward(self, x, fused_act=ops.ACT_NONE, slope=0.2, want_stats=False, want_planes=False):
        """want_stats (interior stride-2 layers): returns (y, BatchNorm partial statistics of y).
        want_planes (3-channel first layer, f32x3 plane path): also write the plane triple of y for the next layer's weight-grad."""
        if self.in_channels == 3:
            return F.ConvC3Fn.apply(x, self.weight, fused_act, slope, want_planes)
        if want_stats:
            return F.ConvFn.apply(x, self.weight, self.stride, self.padding, want_stats)
        return F.ConvFn.apply(x, self.weight, self.stride, self.padding)

    @property
    def emits_bn_stats(self):
        return self.in_channels != 3 and self.stride == 2

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size=(4, 4), stride={self.stride}, padding={self.padding}, bias=False"


class ConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d(in,out,4,stride,pad,bias=False) (model.py:114,118,...,142)."""

    def __init__(self, in_channels, out_channels, kernel_size=4, stride=2, padding=1, bias=False):
        super().__init__()
        if kernel_size != 4 or bias or (stride, padding) not in ((2, 1), (1, 0)):
            raise ValueError("only ConvTranspose2d(k=4, (s,p) in {(2,1),(1,0)}, bias=False) exists in the DiscoGAN path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.stride, self.padding = stride, padding
        w = _init_weight((in_channels, out_channels, 4, 4))
        if out_channels != 3:
            w = ops.krsc_param(w)  # memory [Cin][4][4][Cout]
        self.weight = nn.Parameter(w)

    def forward(self, x, fused_act=ops.ACT_NONE, want_stats=False):
        if self.out_channels == 3:
            return F.ConvTransposeC3Fn.apply(x, self.weight, fused_act)
        if want_stats:
            return F.ConvTransposeFn.apply(x, self.weight, self.stride, self.padding, want_stats)
        return F.ConvTransposeFn.apply(x, self.weight, self.stride, self.padding)

    @property
    def emits_bn_stats(self):
        return self.out_channels != 3 and self.stride == 2

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size=(4, 4), stride={self.stride}, padding={self.padding}, bias=False"


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d(C): eps 1e-5, momentum 0.1, affine, track_running_stats (model.py:12,...)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, y, fused_act=ops.ACT_NONE, slope=0.2, partials=None, z_cm=False, dy_cm=False, z_po=False, dy_po=False):
        if self.training and y.shape[0] * y.shape[2] * y.shape[3] <= 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(y.shape)}")
        return F.BatchNormActFn.apply(y, self.weight, self.bias, self.running_mean, self.running_var,
                                      self.num_batches_tracked, self.training, self.eps, self.momentum, fused_act, slope,
                                      partials, z_cm, dy_cm, z_po, dy_po)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, affine=True, track_running_stats=True"


class _Act(nn.Module):
    act = ops.ACT_NONE

    def __init__(self, slope=0.0, inplace=False):
        super().__init__()
        self.negative_slope, self.inplace = slope, inplace

    def forward(self, x):
        return F.ActFn.apply(x, self.act, self.negative_slope)


class LeakyReLU(_Act):
    act = ops.ACT_LEAKY

    def __init__(self, negative_slope=0.01, inplace=False):
        super().__init__(negative_slope, inplace)


class ReLU(_Act):
    act = ops.ACT_RELU

    def __init__(self, inplace=False):
        super().__init__(0.0, inplace)


class Sigmoid(_Act):
    act = ops.ACT_SIGMOID

    def __init__(self):
        super().__init__(0.0, False)


def _plane_hints(conv, x, nxt, z_is_output=False):
    """Hints for the BatchNorm behind ``conv`` on the f32x3 plane path, from the convolutions on either side of it:
      z_cm / dy_cm (ops.X3_CM): a window input-grad kernel reads z (the next layer ``nxt`` is a stride-2 ConvTranspose2d with
          <= 128 output channels) / dy (``conv`` is a stride-2 Conv2d with <= 128 input channels) -> quad-chunk planes;
      z_po / dy_po (ops.X3_PLANES_ONLY): EVERY reader of z (the next conv's forward and weight-grad) / of dy (this conv's
          input-grad and weight-grad) is a plane kernel -> no fp32 copy.  ``z_is_output``: z is handed to the caller (a
          discriminator feature map, the end of a Sequential) -> it keeps its fp32 copy."""
    if not ops.X3:
        return False, False, False, False
    n, _, h, w = x.shape
    dy_cm = z_cm = dy_po = z_po = False
    s2conv = isinstance(conv, Conv2d) and conv.stride == 2 and conv.in_channels != 3
    s2convT = isinstance(conv, ConvTranspose2d) and conv.stride == 2 and conv.out_channels != 3
    if s2conv:
        dy_cm = ops.x3_window_dgrad(n, h, w, conv.in_channels, conv.out_channels)
        dy_po = ops.x3_all_plane_readers(n, h, w, conv.in_channels, conv.out_channels, forward_is_dgrad=True)
        ho, wo = h // 2, w // 2
    elif s2convT:
        # backward of the transposed conv: input-grad = a FORWARD-form conv of dy, weight-grad with dy in the x role
        dy_po = ops.x3_all_plane_readers(n, 2 * h, 2 * w, conv.out_channels, conv.in_channels, forward_is_dgrad=False)
        ho, wo = 2 * h, 2 * w
    else:
        ho, wo = (4, 4) if isinstance(conv, ConvTranspose2d) else (1, 1)
    if not z_is_output and isinstance(nxt, ConvTranspose2d) and nxt.stride == 2 and nxt.out_channels != 3:
        # z = this group's output [n, nxt.in_channels, ho, wo]; the transposed conv's forward is the input-grad of a
        # Conv2d(nxt.out_channels -> nxt.in_channels) on a [2 ho, 2 wo] input
        z_cm = ops.x3_window_dgrad(n, 2 * ho, 2 * wo, nxt.out_channels, nxt.in_channels)
        z_po = ops.x3_all_plane_readers(n, 2 * ho, 2 * wo, nxt.out_channels, nxt.in_channels, forward_is_dgrad=True)
    elif not z_is_output and isinstance(nxt, Conv2d) and nxt.stride == 2 and nxt.in_channels != 3:
        z_po = ops.x3_all_plane_readers(n, ho, wo, nxt.in_channels, nxt.out_channels, forward_is_dgrad=False)
    return z_cm and ops.X3_CM, dy_cm and ops.X3_CM, z_po, dy_po


C3_PLANES = __import__("os").environ.get("DG_X3_C3_PLANES", "1") != "0"      # A/B switch of _first_layer_planes


def _first_layer_planes(x, nxt):
    """f32x3 plane path: will the layer behind the 3-channel first conv read the first conv's output as a plane triple -- in its
    FORWARD (the window forward kernel, any pass) or in its weight-gradient (only when its weights take gradients in this pass)?
    Then the first conv writes the triple itself instead of leaving a separate split pass over the network's largest activation to
    ``ops.planes_of``."""
    if not (ops.X3 and C3_PLANES and isinstance(nxt, Conv2d) and nxt.stride == 2):
        return False
    n, _, h, w = x.shape
    if ops._x3_ok(0, n, h // 2, w // 2, nxt.in_channels, nxt.out_channels, 2, 1) and \
            (ops.X3_RSP or getattr(nxt.weight, "_dg_x3", (None, 0, None))[2] is not None):
        return True
    return torch.is_grad_enabled() and nxt.weight.requires_grad and ops._x3_ok(2, n, h // 2, w // 2, nxt.in_channels, nxt.out_channels, 2, 1)


def drain(gen):
    """Run a ``*_steps`` generator to completion and return its value."""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


def _run_fused(layers, x):
    return drain(_run_fused_steps(layers, x))


def _run_fused_steps(layers, x):
    """Walk a list of layer modules fusing [conv][BatchNorm][activation] groups into kernel calls.
    A generator: yields after every group so that a caller can issue two networks layer by layer in lock
    step (trainer.py); the final activation is the generator's return value."""
    i, n = 0, len(layers)
    while i < n:
        conv = layers[i]
        bn = layers[i + 1] if i + 1 < n and isinstance(layers[i + 1], BatchNorm2d) else None
        j = i + (2 if bn is not None else 1)
        act_mod = layers[j] if j < n and isinstance(layers[j], _Act) else None
        act = act_mod.act if act_mod is not None else ops.ACT_NONE
        slope = act_mod.negative_slope if act_mod is not None else 0.0
        if bn is not None and bn.training and not ops.X3 and _fuse_stats(conv, x) and conv.emits_bn_stats:
            y, st = conv(x, want_stats=_fuse_stats(conv, x))       # BN statistics from the conv / split-K reduce kernel
            x = bn(y, act, slope, st)
        elif bn is not None:
            nxt = layers[j + (1 if act_mod is not None else 0)] if j + (1 if act_mod is not None else 0) < n else None
            z_cm, dy_cm, z_po, dy_po = _plane_hints(conv, x, nxt, z_is_output=nxt is None)
            st = None
            if ops.X3 and ops.X3_FUSE_STATS and bn.training and conv.emits_bn_stats:
                y, st = conv(x, want_stats=True)            # plane path: statistics from the plane kernel's epilogue (None: no such kernel)
                st = st if st.numel() > 0 else None
            else:
                y = conv(x)
            yield
            x = bn(y, act, slope, st, z_cm, dy_cm, z_po, dy_po)
        elif isinstance(conv, Conv2d) and conv.in_channels == 3 and act in (ops.ACT_LEAKY, ops.ACT_RELU, ops.ACT_NONE):
            nxt = layers[j + (1 if act_mod is not None else 0)] if j + (1 if act_mod is not None else 0) < n else None
            x = conv(x, act, slope, want_planes=_first_layer_planes(x, nxt))      # conv1 + LeakyReLU in one kernel
        elif isinstance(conv, ConvTranspose2d) and conv.out_channels == 3 and act in (ops.ACT_SIGMOID, ops.ACT_NONE):
            x = conv(x, act)                                # last convT + Sigmoid in one kernel
        else:
            x = conv(x)
            if act_mod is not None:
                x = act_mod(x)
        i = j + (1 if act_mod is not None else 0)
        if i < n:
            yield
    return x


# ---- grouped passes (round 4) --------------------------------------------------------------------------------------------
# The reference runs the passes of an iteration in pairs of identical shape (image_translation.py:342-361).  group_generators /
# group_discriminators run ``g`` such passes -- different networks of the same architecture, or the same network on different inputs
# -- layer by layer with ONE launch per kernel for all of them (functional.*GroupFn -> ops.*_g -> dg_*_g).  Training mode, fp32
# tensors, exact-fp32 or register-staged f32x3 arithmetic (ops.group_ok()); results per pass are bitwise those of net(x).
def _fire_forward_hooks(records):
    """Forward hooks registered on the fused modules (BatchNorm2d, the first Conv2d: tests/kink_probe.py records activation patterns
    through them) see the same calls as in the one-pass form: problem by problem, layer by layer -- the order net(x) would have fired them."""
    for rec in records:
        for mod, inp, out in rec:
            for hook in list(mod._forward_hooks.values()):
                hook(mod, (inp,), out)


def _group_layers(layer_lists, xs):
    g, n = len(xs), len(layer_lists[0])
    records = [[] for _ in range(g)]
    i = 0
    while i < n:
        convs = [L[i] for L in layer_lists]
        conv = convs[0]
        has_bn = i + 1 < n and isinstance(layer_lists[0][i + 1], BatchNorm2d)
        j = i + (2 if has_bn else 1)
        act_mod = layer_lists[0][j] if j < n and isinstance(layer_lists[0][j], _Act) else None
        act = act_mod.act if act_mod is not None else ops.ACT_NONE
        slope = act_mod.negative_slope if act_mod is not None else 0.0
        ws = [c.weight for c in convs]
        if has_bn:
            bns = [L[i + 1] for L in layer_lists]
            if not all(b.training for b in bns):
                raise RuntimeError("grouped passes run BatchNorm in training mode only")
            ys = F.ConvGroupFn.apply(g, conv.stride, conv.padding, isinstance(conv, ConvTranspose2d), *xs, *ws)
            if ys[0].shape[0] * ys[0].shape[2] * ys[0].shape[3] <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(ys[0].shape)}")
            xs = F.BatchNormActGroupFn.apply(g, bns[0].eps, bns[0].momentum, act, slope, *ys, *[b.weight for b in bns], *[b.bias for b in bns],
                                             *[b.running_mean for b in bns], *[b.running_var for b in bns], *[b.num_batches_tracked for b in bns])
            for k in range(g):
                records[k].append((bns[k], ys[k], xs[k]))
        elif isinstance(conv, Conv2d) and conv.in_channels == 3 and act in (ops.ACT_LEAKY, ops.ACT_RELU, ops.ACT_NONE):
            ins = xs
            xs = F.ConvC3GroupFn.apply(g, act, slope, *xs, *ws)                 # conv1 + LeakyReLU in one kernel
            for k in range(g):
                records[k].append((convs[k], ins[k], xs[k]))
        elif isinstance(conv, ConvTranspose2d) and conv.out_channels == 3 and act in (ops.ACT_SIGMOID, ops.ACT_NONE):
            xs = F.ConvTransposeC3GroupFn.apply(g, act, *xs, *ws)               # last convT + Sigmoid in one kernel
        else:
            xs = F.ConvGroupFn.apply(g, conv.stride, conv.padding, isinstance(conv, ConvTranspose2d), *xs, *ws)
            if act_mod is not None:
                xs = F.ActGroupFn.apply(g, act, slope, *xs)
        xs = list(xs)
        i = j + (1 if act_mod is not None else 0)
    _fire_forward_hooks(records)
    return xs


def group_generators(nets, xs, cut_at_bottleneck=False):
    """[net(x) for net, x in zip(nets, xs)] for generators of one architecture, every layer as one grouped launch per kernel.
    cut_at_bottleneck: the autograd graph is CUT at the encoders' outputs (the [N, 100, 1, 1] bottleneck activations): the decoders run
    on detached leaves; returns (outputs, encoder outputs, the leaves).  loss.backward() then stops at the leaves -- every decoder
    gradient is complete -- and ``torch.autograd.backward(encoder outputs, [leaf.grad ...])`` continues through the encoders: the point
    at which the trainer sends the decoder gradients off while the encoder half still runs (trainer.py, overlap_comm="graph")."""
    if any(n_.main is not None for n_ in nets):
        raise RuntimeError("grouped passes need the encoder / decoder form of the generators")
    if not cut_at_bottleneck:
        return _group_layers([list(n_.encoder) + list(n_.decoder) for n_ in nets], list(xs))
    hs = _group_layers([list(n_.encoder) for n_ in nets], list(xs))
    leaves = [h.detach().requires_grad_(True) for h in hs]
    return _group_layers([list(n_.decoder) for n_ in nets], leaves), hs, leaves


def group_discriminators(nets, xs):
    """[net(x) for ...] for discriminators of one architecture: list of (sigmoid output, feature maps).  A network may appear several
    times (its real and its fake pass): those problems must be consecutive."""
    g = len(xs)
    d0 = nets[0]
    hs = list(F.ConvC3GroupFn.apply(g, ops.ACT_LEAKY, d0.relu1.negative_slope, *xs, *[d.conv1.weight for d in nets]))
    feats = [[] for _ in range(g)]
    records = [[(nets[k].conv1, xs[k], hs[k])] for k in range(g)]
    for i in range(2, d0.n_stages + 1):
        convs = [getattr(d, f"conv{i}") for d in nets]
        bns = [getattr(d, f"bn{i}") for d in nets]
        if not all(b.training for b in bns):
            raise RuntimeError("grouped passes run BatchNorm in training mode only")
        ys = F.ConvGroupFn.apply(g, convs[0].stride, convs[0].padding, False, *hs, *[c.weight for c in convs])
        hs = list(F.BatchNormActGroupFn.apply(g, bns[0].eps, bns[0].momentum, ops.ACT_LEAKY, getattr(d0, f"relu{i}").negative_slope, *ys,
                                              *[b.weight for b in bns], *[b.bias for b in bns], *[b.running_mean for b in bns],
                                              *[b.running_var for b in bns], *[b.num_batches_tracked for b in bns]))
        for k in range(g):
            feats[k].append(hs[k])
            records[k].append((bns[k], ys[k], hs[k]))
    heads = [getattr(d, f"conv{d0.n_stages + 1}") for d in nets]
    ys = F.ConvGroupFn.apply(g, heads[0].stride, heads[0].padding, False, *hs, *[c.weight for c in heads])
    outs = F.ActGroupFn.apply(g, ops.ACT_SIGMOID, 0.0, *ys)
    _fire_forward_hooks(records)
    return [(outs[k], feats[k]) for k in range(g)]


class Discriminator(_FlatGradMixin, nn.Module):
    """Reference model.py:5-69.  Returns ``(sigmoid [N,1,1,1], [relu2, ..., relu_n])``."""

    def __init__(self, image_size: int = 512):
        super().__init__()
        ch = stage_channels(image_size)
        self.image_size, self.n_stages = image_size, len(ch)
        cin = 3
        for i, c in enumerate(ch, start=1):
            setattr(self, f"conv{i}", Conv2d(cin, c, 4, 2, 1, bias=False))
            if i >= 2:
                setattr(self, f"bn{i}", BatchNorm2d(c))
            setattr(self, f"relu{i}", LeakyReLU(0.2, inplace=True))
            cin = c
        setattr(self, f"conv{len(ch) + 1}", Conv2d(cin, 1, 4, 1, 0, bias=False))
        self.sigmoid = Sigmoid()

    def forward(self, input_tensor):
        return drain(self.forward_steps(input_tensor))

    def forward_steps(self, input_tensor):
        """forward() as a generator that yields after every conv(+BN+act) group."""
        feats = []
        h = self.conv1(input_tensor, ops.ACT_LEAKY, self.relu1.negative_slope,
                       want_planes=_first_layer_planes(input_tensor, getattr(self, "conv2", None)))
        yield
        for i in range(2, self.n_stages + 1):
            relu, bn, conv = getattr(self, f"relu{i}"), getattr(self, f"bn{i}"), getattr(self, f"conv{i}")
            if bn.training and not ops.X3 and _fuse_stats(conv, h):
                y, st = conv(h, want_stats=_fuse_stats(conv, h))   # BN statistics from the conv / split-K reduce kernel
                h = bn(y, ops.ACT_LEAKY, relu.negative_slope, st)
            else:
                _, dy_cm, _, dy_po = _plane_hints(conv, h, None, z_is_output=True)      # z is a feature map: it keeps its fp32 copy
                st = None
                if ops.X3 and ops.X3_FUSE_STATS and bn.training:
                    y, st = conv(h, want_stats=True)        # plane path: statistics from the plane kernel's epilogue
                    st = st if st.numel() > 0 else None
                else:
                    y = conv(h)
                yield
                h = bn(y, ops.ACT_LEAKY, relu.negative_slope, st, False, dy_cm, False, dy_po)
            feats.append(h)
            yield
        out = self.sigmoid(getattr(self, f"conv{self.n_stages + 1}")(h))
        return out, feats


class Generator(_FlatGradMixin, nn.Module):
    """Reference model.py:72-225 (``extra_layers`` True/False build the same network there too)."""

    def __init__(self, extra_layers: bool = False, image_size: int = 512):
        super().__init__()
        ch = stage_channels(image_size)
        self.image_size = image_size
        enc = []
        cin = 3
        for i, c in enumerate(ch):
            enc.append(Conv2d(cin, c, 4, 2, 1, bias=False))
            if i >= 1:
                enc.append(BatchNorm2d(c))
            enc.append(LeakyReLU(0.2, inplace=True))
            cin = c
        enc += [Conv2d(cin, 100, 4, 1, 0, bias=False), BatchNorm2d(100), LeakyReLU(0.2, inplace=True)]
        self.encoder = nn.Sequential(*enc)
        dec = [ConvTranspose2d(100, ch[-1], 4, 1, 0, bias=False), BatchNorm2d(ch[-1]), ReLU(True)]
        for i in range(len(ch) - 1, 0, -1):
            dec += [ConvTranspose2d(ch[i], ch[i - 1], 4, 2, 1, bias=False), BatchNorm2d(ch[i - 1]), ReLU(True)]
        dec += [ConvTranspose2d(ch[0], 3, 4, 2, 1, bias=False), Sigmoid()]
        self.decoder = nn.Sequential(*dec)
        self.main = None  # legacy hook, model.py:215-220

    def forward(self, input_tensor):
        if self.main is not None:
            return self.main(input_tensor)
        return drain(self.forward_steps(input_tensor))

    def forward_steps(self, input_tensor):
        """forward() as a generator that yields after every conv(+BN+act) group."""
        if self.main is not None:
            return self.main(input_tensor)
        h = yield from _run_fused_steps(list(self.encoder), input_tensor)
        yield
        return (yield from _run_fused_steps(list(self.decoder), h))
