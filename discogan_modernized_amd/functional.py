"""torch.autograd glue: each Function's forward/backward is one or two C-ABI kernel calls.

Autograd only sequences the calls (the reference relies on ``loss.backward()``,
image_translation.py:385-390); no arithmetic is done by ATen except gradient accumulation into
``.grad`` buffers.  Functions skip work autograd does not need (``ctx.needs_input_grad``), which is
how the dead backward work of the reference disappears when the caller freezes the
side that is not stepped.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops


def _flat_grad_of(param):
    """The flat-buffer gradient view of a parameter (optim.Adam installs it) when it is the live .grad:
    backward kernels then accumulate into it directly and autograd gets None for that input, instead of
    materialising a gradient tensor and running an ATen add per parameter per pass."""
    fg = getattr(param, "_dg_flat_grad", None)
    if fg is not None and param.grad is fg:
        return fg
    return None


# Bucketed gradient exchange (trainer._GradBuckets): FINAL_PASS is True while the trainer issues the forward calls
# whose backward is the LAST one to touch each parameter of the stepped networks; FINAL_HOOK(param) is called right
# after that backward has enqueued the kernel that completes the parameter's gradient in the flat buffer.
FINAL_PASS = False
FINAL_HOOK = None


def _final(ctx_flag, *params):
    if ctx_flag and FINAL_HOOK is not None:
        for p in params:
            FINAL_HOOK(p)


# Auxiliary HIP stream for weight-gradient kernels (set by the trainer).  dgrad feeds the next BN-backward
# on the chain's own stream; wgrad only feeds Adam at the end of the iteration, so it is launched on this
# stream behind an event and fills the matrix pipes while the chain runs its HBM-bound BN-backward kernels.
WGRAD_STREAM = None


def _launch_wgrad(fn, *tensors):
    """Run fn() (a wgrad that accumulates into the flat gradient buffer) on WGRAD_STREAM, after the work
    already queued on the current stream; keeps the tensors it reads alive for that stream."""
    aux = WGRAD_STREAM
    if aux is None:
        fn()
        return
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    aux.wait_event(ev)
    with torch.cuda.stream(aux):
        fn()
    for t in tensors:
        t.record_stream(aux)


def _release_unless_wgrad(ctx, x):
    """Forward of an interior conv: the layer input's bf16 shadow / plane triple is read again only by this layer's weight
    gradient.  No weight gradient coming (no_grad generator passes of a D-step, frozen discriminators of a G-step) -> the
    derived copy is dropped right here instead of at the end of the iteration."""
    if not ctx.needs_input_grad[1]:
        ops.derived_release(x)


def _release_after_backward(dy, x):
    """Backward of an interior conv: input-grad and weight-grad were the last readers of dy's and x's derived copies (with
    the weight gradient on a third stream the copies stay until the trainer clears the tables)."""
    if WGRAD_STREAM is None:
        ops.derived_release(dy, x)


class ConvFn(Function):
    """nn.Conv2d(C,K,4,stride,pad,bias=False), interior (C % 32 == 0)."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, want_stats=False):
        x = ops.as_nhwc(x)
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad)
        ctx.wref = w
        ctx.final = FINAL_PASS
        if not want_stats:
            y = ops.conv_fwd(x, w, stride, pad)
            _release_unless_wgrad(ctx, x)
            return y
        y, stat = ops.conv_fwd(x, w, stride, pad, want_stats=want_stats)
        _release_unless_wgrad(ctx, x)
        if stat is None:
            stat = torch.empty(0, device=y.device)
        ctx.mark_non_differentiable(stat)
        # without this autograd hands backward() a freshly zero-filled tensor of stat's shape for the non-differentiable output:
        # one fill launch per fused-statistics conv and backward (48 per iteration at 512 px, rocprof round 3)
        ctx.set_materialize_grads(False)
        return y, stat

    @staticmethod
    def backward(ctx, dy, dstat=None):
        if dy is None:                                 # (set_materialize_grads(False): an unused output arrives as None)
            return None, None, None, None, None
        x, w = ctx.saved_tensors
        w = ctx.wref                                   # the Parameter object itself (carries the bf16 shadow attribute)
        stride, pad = ctx.sp
        dy = ops.as_nhwc(dy)
        dx = ops.conv_dgrad(dy, w, (x.shape[2], x.shape[3]), stride, pad) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            fg = _flat_grad_of(ctx.wref)
            if fg is not None:
                _launch_wgrad(lambda: ops.conv_wgrad(dy, x, stride, pad, out=fg, accumulate=True), dy, x)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.conv_wgrad(dy, x, stride, pad)
        _release_after_backward(dy, x)
        return dx, dw, None, None, None


class ConvTransposeFn(Function):
    """nn.ConvTranspose2d(Cin,Cout,4,stride,pad,bias=False), interior: forward = conv dgrad with the
    same weight tensor."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, want_stats=False):
        x = ops.as_nhwc(x)
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad)
        ctx.wref = w
        ctx.final = FINAL_PASS
        hin, win = x.shape[2], x.shape[3]
        hout, wout = (hin - 1) * stride - 2 * pad + 4, (win - 1) * stride - 2 * pad + 4
        ctx.out_hw = (hout, wout)
        if not want_stats:
            y = ops.conv_dgrad(x, w, (hout, wout), stride, pad)
            _release_unless_wgrad(ctx, x)
            return y
        y, stat = ops.conv_dgrad(x, w, (hout, wout), stride, pad, want_stats=want_stats)
        _release_unless_wgrad(ctx, x)
        if stat is None:
            stat = torch.empty(0, device=y.device)
        ctx.mark_non_differentiable(stat)
        # without this autograd hands backward() a freshly zero-filled tensor of stat's shape for the non-differentiable output:
        # one fill launch per fused-statistics conv and backward (48 per iteration at 512 px, rocprof round 3)
        ctx.set_materialize_grads(False)
        return y, stat

    @staticmethod
    def backward(ctx, dy, dstat=None):
        if dy is None:
            return None, None, None, None, None
        x, w = ctx.saved_tensors
        w = ctx.wref
        stride, pad = ctx.sp
        dy = ops.as_nhwc(dy)
        dx = ops.conv_fwd(dy, w, stride, pad) if ctx.needs_input_grad[0] else None
        # dw[cin][r][s][cout] = sum x[..cin] * dy[..cout]: conv wgrad with roles (dy := x, x := dy)
        dw = None
        if ctx.needs_input_grad[1]:
            fg = _flat_grad_of(ctx.wref)
            if fg is not None:
                _launch_wgrad(lambda: ops.conv_wgrad(x, dy, stride, pad, out=fg, accumulate=True), dy, x)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.conv_wgrad(x, dy, stride, pad)
        _release_after_backward(dy, x)
        return dx, dw, None, None, None


class ConvC3Fn(Function):
    """First layer nn.Conv2d(3,K,4,2,1) on the NCHW image, with the following in-place
    LeakyReLU fused (model.py:8-9, 80-81)."""

    @staticmethod
    def forward(ctx, x, w, act, slope, want_planes=False):
        y = ops.c3_fwd(x, w, act, slope, want_planes=want_planes)
        ctx.save_for_backward(x, w, y)
        ctx.act = (act, slope)
        ctx.wref = w
        ctx.final = FINAL_PASS
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        act, slope = ctx.act
        if ctx.needs_input_grad[0] or act == ops.ACT_NONE:
            g = ops.act_bwd(dy, y, act, slope) if act != ops.ACT_NONE else ops.as_nhwc(dy)
            fuse = {}
        else:
            # weight gradient only (image input): the activation backward rides in the wgrad kernel's dy loads
            g = ops.as_nhwc(dy)
            fuse = dict(act_out=y, act=act, slope=slope)
        dx = ops.c3_dgrad(g, w, ops.ACT_NONE) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            fg = _flat_grad_of(ctx.wref)
            if fg is not None and fg.is_contiguous():
                _launch_wgrad(lambda: ops.c3_wgrad(g, x, out=fg, accumulate=True, **fuse), g, x)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.c3_wgrad(g, x, **fuse)
        return dx, dw, None, None, None


class ConvTransposeC3Fn(Function):
    """Last layer nn.ConvTranspose2d(K,3,4,2,1) + Sigmoid producing the NCHW image (model.py:142-143)."""

    @staticmethod
    def forward(ctx, x, w, act):
        x = ops.as_nhwc(x)
        out = ops.c3_dgrad(x, w, act)
        ctx.save_for_backward(x, w, out)
        ctx.act = act
        ctx.wref = w
        ctx.final = FINAL_PASS
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        g = ops.act_bwd(dout, out, ctx.act) if ctx.act != ops.ACT_NONE else dout.contiguous()
        dx = ops.c3_fwd(g, w, ops.ACT_NONE) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            fg = _flat_grad_of(ctx.wref)
            if fg is not None and fg.is_contiguous():
                _launch_wgrad(lambda: ops.c3_wgrad(x, g, out=fg, accumulate=True), g, x)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.c3_wgrad(x, g)
        return dx, dw, None


class BatchNormActFn(Function):
    """nn.BatchNorm2d (+ in-place LeakyReLU / ReLU).  Training mode updates the running buffers in
    the kernel exactly like PyTorch (momentum 0.1, unbiased running_var, num_batches_tracked += 1)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, nbt, training, eps, momentum, act, slope,
                partials=None, z_cm=False, dy_cm=False, z_po=False, dy_po=False):
        """z_cm / dy_cm (f32x3 plane path, ops.X3_CM): the plane triple of z / of this layer's dy will be read by a window
        input-grad kernel -> written in the quad-chunk layout; z_po / dy_po (ops.X3_PLANES_ONLY): every reader of z / dy is a plane
        kernel -> the fp32 copy is not written (model.py decides both from the neighbouring convolutions)."""
        y = ops.as_nhwc(y)
        if training and partials is not None and partials.numel() > 0:
            # statistics came out of the producing conv kernel's epilogue: no extra pass over y
            saved = ops.bn_stats_from_partials(partials, y, running_mean, running_var, nbt, eps, momentum)
        elif training:
            saved = ops.bn_train_stats(y, running_mean, running_var, nbt, eps, momentum)
        else:
            saved = torch.stack([running_mean, torch.rsqrt(running_var + eps)])
        z = ops.bn_act_fwd(y, saved, gamma, beta, act, slope, planes_cm=z_cm, planes_only=z_po and training)
        ctx.save_for_backward(y, saved, gamma, beta)
        ctx.cfg = (act, slope, training)
        ctx.dy_cm = bool(dy_cm)
        ctx.dy_po = bool(dy_po)
        ctx.prefs = (gamma, beta)
        ctx.final = FINAL_PASS
        return z

    @staticmethod
    def backward(ctx, dz):
        y, saved, gamma, beta = ctx.saved_tensors
        act, slope, training = ctx.cfg
        if not training:
            raise RuntimeError("BatchNormActFn: backward in eval mode is not supported")
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        fg, fb = _flat_grad_of(ctx.prefs[0]), _flat_grad_of(ctx.prefs[1])
        if need_p and fg is not None and fb is not None:
            dy, _, _ = ops.bn_act_bwd(dz, y, saved, gamma, beta, act, slope, out_grads=(fg, fb), planes_cm=ctx.dy_cm, planes_only=ctx.dy_po)
            _final(ctx.final, ctx.prefs[0], ctx.prefs[1])
            return (dy,) + (None,) * 15
        dy, dgamma, dbeta = ops.bn_act_bwd(dz, y, saved, gamma, beta, act, slope, need_param_grads=need_p, planes_cm=ctx.dy_cm,
                                           planes_only=ctx.dy_po)
        return (dy, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                None, None, None, None, None, None, None, None, None, None, None, None, None)


class ActFn(Function):
    """Stand-alone LeakyReLU / ReLU / Sigmoid (backward from the OUTPUT, in-place semantics)."""

    @staticmethod
    def forward(ctx, x, act, slope):
        out = ops.act_fwd(x, act, slope)
        ctx.save_for_backward(out)
        ctx.cfg = (act, slope)
        return out

    @staticmethod
    def backward(ctx, dy):
        (out,) = ctx.saved_tensors
        act, slope = ctx.cfg
        return ops.act_bwd(dy, out, act, slope), None, None


class MSELossFn(Function):
    @staticmethod
    def forward(ctx, x, t, out=None):
        loss, xd, td = ops.mse_fwd(x, t, out)
        ctx.save_for_backward(xd, td)
        ctx.in_strides = x.stride()
        return loss

    @staticmethod
    def backward(ctx, gout):
        xd, td = ctx.saved_tensors
        gout = gout.contiguous()
        dx = ops.mse_bwd(xd, td, gout) if ctx.needs_input_grad[0] else None
        dt = None
        if ctx.needs_input_grad[1]:
            dt = ops.mse_bwd(td, xd, gout)
        return dx, dt, None


class BCELossFn(Function):
    """nn.BCELoss against a constant label tensor (image_translation.py:157-166)."""

    @staticmethod
    def forward(ctx, p, label, out=None):
        shape = p.shape
        loss, pc = ops.bce_fwd(p.reshape(-1), label, out)
        ctx.save_for_backward(pc)
        ctx.cfg = (label, shape)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (pc,) = ctx.saved_tensors
        label, shape = ctx.cfg
        return ops.bce_bwd(pc, label, gout.contiguous()).reshape(shape), None, None


class BCETargetLossFn(Function):
    """nn.BCELoss against a target TENSOR (no gradient w.r.t. the target), image_translation.py:157-166."""

    @staticmethod
    def forward(ctx, p, target):
        shape = p.shape
        loss, pc, tc = ops.bce_target_fwd(p.reshape(-1), target.detach().reshape(-1))
        ctx.save_for_backward(pc, tc)
        ctx.shape = shape
        return loss

    @staticmethod
    def backward(ctx, gout):
        pc, tc = ctx.saved_tensors
        return ops.bce_target_bwd(pc, tc, gout.contiguous()).reshape(ctx.shape), None


class HingeEmbeddingLossFn(Function):
    """nn.HingeEmbeddingLoss(margin, mean) for targets in {+1, -1} (image_translation.py:141-142,269)."""

    @staticmethod
    def forward(ctx, x, y, margin):
        loss, xd, yd = ops.hinge_fwd(x, y.detach(), margin)
        ctx.save_for_backward(xd, yd)
        ctx.margin = margin
        ctx.in_shape = x.shape
        return loss

    @staticmethod
    def backward(ctx, gout):
        xd, yd = ctx.saved_tensors
        return ops.hinge_bwd(xd, yd, ctx.margin, gout.contiguous()), None, None


class FeatureMatchFn(Function):
    """One layer of get_fm_loss: mean((real.mean(0) - fake.mean(0))**2)."""

    @staticmethod
    def forward(ctx, real, fake, out=None):
        loss, diff, rd, fd = ops.fm_fwd(real, fake, out)
        ctx.save_for_backward(diff, rd, fd)
        return loss

    @staticmethod
    def backward(ctx, gout):
        diff, rd, fd = ctx.saved_tensors
        dreal, dfake = ops.fm_bwd(diff, rd, fd, gout.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dreal, dfake, None


class LossMixFn(Function):
    """The whole curriculum mix (get_gan_loss's 0.5*(real+fake), get_fm_loss's layer sum, and
    image_translation.py:367-382) in one launch over the loss vector, and all gradient seeds in one launch.
    ``slots`` are the views of ``lossvec`` that the differentiated loss (`which`: 6 gen_loss, 7 dis_loss)
    depends on, ``idx`` their slot numbers: autograd routes each seed to the loss op that wrote the slot and
    never visits the others.  Returns the 8 outputs of dg_loss_mix_fwd as 0-dim tensors."""

    @staticmethod
    def forward(ctx, lossvec, nfm, rate, arch, which, idx, *slots):
        out = ops.loss_mix_fwd(lossvec, nfm, rate, arch)
        ctx.cfg = (lossvec.numel(), nfm, rate, arch, which, idx)
        ctx.set_materialize_grads(False)
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gouts):
        nslots, nfm, rate, arch, which, idx = ctx.cfg
        gout = gouts[which]
        if gout is None:
            return (None,) * (6 + len(idx))
        gv = ops.loss_mix_bwd(gout.contiguous(), nslots, nfm, rate, arch, which)
        return (None, None, None, None, None, None) + tuple(gv[i] for i in idx)
