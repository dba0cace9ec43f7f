"""torch.autograd glue: each Function's forward/backward is one or two C-ABI kernel calls.

Autograd only sequences the calls (the reference relies on ``loss.backward()``,
image_translation.py:385-390); no arithmetic is done by ATen except gradient accumulation into
``.grad`` buffers.  Functions skip work autograd does not need (``ctx.needs_input_grad``), which is
how the dead backward work of the reference disappears when the caller freezes the
side that is not stepped.
"""
from __future__ import annotations

import functools

import torch
from torch.autograd import Function

from . import ops


def _fwd(f):
    """forward(): remember the ops.Context (arithmetic, operand forms, derived-copy tables) the call runs under."""
    @functools.wraps(f)
    def w(ctx, *a, **k):
        ctx.dgc = ops.current()
        return f(ctx, *a, **k)
    return staticmethod(w)


def _bwd(f):
    """backward(): run under the forward's context, whatever is ambient when autograd gets here -- two trainers with different
    arithmetic may be interleaved call by call."""
    @functools.wraps(f)
    def w(ctx, *g):
        with ops.use(ctx.dgc):
            return f(ctx, *g)
    return staticmethod(w)


# ConvC3Fn.backward with an input gradient: take the LeakyReLU backward inside the input-gradient and the weight-gradient kernels (bitwise
# the same results as the stand-alone act_bwd pass in front of them; False = that pass, for A/B)
FUSE_C3_DGRAD_ACT = __import__("os").environ.get("DG_FUSE_C3_DGRAD_ACT", "1") != "0"


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


def _release_unless_wgrad(ctx, x):
    """Forward of an interior conv: the layer input's bf16 shadow / plane triple is read again only by this layer's weight
    gradient.  No weight gradient coming (no_grad generator passes of a D-step, frozen discriminators of a G-step) -> the
    derived copy is dropped right here instead of at the end of the iteration."""
    if not ctx.needs_input_grad[1]:
        ops.derived_release(x)


def _release_after_backward(dy, x):
    """Backward of an interior conv: input-grad and weight-grad were the last readers of dy's and x's derived copies."""
    ops.derived_release(dy, x)


class ConvFn(Function):
    """nn.Conv2d(C,K,4,stride,pad,bias=False), interior (C % 32 == 0)."""

    @_fwd
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

    @_bwd
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
                ops.conv_wgrad(dy, x, stride, pad, out=fg, accumulate=True)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.conv_wgrad(dy, x, stride, pad)
        _release_after_backward(dy, x)
        return dx, dw, None, None, None


class ConvTransposeFn(Function):
    """nn.ConvTranspose2d(Cin,Cout,4,stride,pad,bias=False), interior: forward = conv dgrad with the
    same weight tensor."""

    @_fwd
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

    @_bwd
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
                ops.conv_wgrad(x, dy, stride, pad, out=fg, accumulate=True)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.conv_wgrad(x, dy, stride, pad)
        _release_after_backward(dy, x)
        return dx, dw, None, None, None


class ConvC3Fn(Function):
    """First layer nn.Conv2d(3,K,4,2,1) on the NCHW image, with the following in-place
    LeakyReLU fused (model.py:8-9, 80-81)."""

    @_fwd
    def forward(ctx, x, w, act, slope, want_planes=False):
        y = ops.c3_fwd(x, w, act, slope, want_planes=want_planes)
        ctx.save_for_backward(x, w, y)
        ctx.act = (act, slope)
        ctx.wref = w
        ctx.final = FINAL_PASS
        return y

    @_bwd
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        act, slope = ctx.act
        fused_dgrad = FUSE_C3_DGRAD_ACT and act == ops.ACT_LEAKY and ops.c3_dgrad_act_ok(w.shape[0])
        if act == ops.ACT_NONE or (ctx.needs_input_grad[0] and not fused_dgrad):
            g = ops.act_bwd(dy, y, act, slope) if act != ops.ACT_NONE else ops.as_nhwc(dy)
            fuse = {}
        else:
            # the activation backward rides in the dy loads of the weight-gradient kernel and (round 4) of the input-gradient kernel: no
            # stand-alone pass that reads dy and the saved output and writes a third tensor of that size
            g = ops.as_nhwc(dy)
            fuse = dict(act_out=y, act=act, slope=slope)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.c3_dgrad(g, w, ops.ACT_NONE, act_out=y, in_act=act, slope=slope) if fuse else ops.c3_dgrad(g, w, ops.ACT_NONE)
        dw = None
        if ctx.needs_input_grad[1]:
            fg = _flat_grad_of(ctx.wref)
            if fg is not None and fg.is_contiguous():
                ops.c3_wgrad(g, x, out=fg, accumulate=True, **fuse)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.c3_wgrad(g, x, **fuse)
        return dx, dw, None, None, None


class ConvTransposeC3Fn(Function):
    """Last layer nn.ConvTranspose2d(K,3,4,2,1) + Sigmoid producing the NCHW image (model.py:142-143)."""

    @_fwd
    def forward(ctx, x, w, act):
        x = ops.as_nhwc(x)
        out = ops.c3_dgrad(x, w, act)
        ctx.save_for_backward(x, w, out)
        ctx.act = act
        ctx.wref = w
        ctx.final = FINAL_PASS
        return out

    @_bwd
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        g = ops.act_bwd(dout, out, ctx.act) if ctx.act != ops.ACT_NONE else dout.contiguous()
        dx = ops.c3_fwd(g, w, ops.ACT_NONE) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            fg = _flat_grad_of(ctx.wref)
            if fg is not None and fg.is_contiguous():
                ops.c3_wgrad(x, g, out=fg, accumulate=True)
                _final(ctx.final, ctx.wref)
            else:
                dw = ops.c3_wgrad(x, g)
        return dx, dw, None


class BatchNormActFn(Function):
    """nn.BatchNorm2d (+ in-place LeakyReLU / ReLU).  Training mode updates the running buffers in
    the kernel exactly like PyTorch (momentum 0.1, unbiased running_var, num_batches_tracked += 1)."""

    @_fwd
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

    @_bwd
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

    @_fwd
    def forward(ctx, x, act, slope):
        out = ops.act_fwd(x, act, slope)
        ctx.save_for_backward(out)
        ctx.cfg = (act, slope)
        return out

    @_bwd
    def backward(ctx, dy):
        (out,) = ctx.saved_tensors
        act, slope = ctx.cfg
        return ops.act_bwd(dy, out, act, slope), None, None


class MSELossFn(Function):
    @_fwd
    def forward(ctx, x, t, out=None):
        loss, xd, td = ops.mse_fwd(x, t, out)
        ctx.save_for_backward(xd, td)
        ctx.in_strides = x.stride()
        return loss

    @_bwd
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

    @_fwd
    def forward(ctx, p, label, out=None):
        shape = p.shape
        loss, pc = ops.bce_fwd(p.reshape(-1), label, out)
        ctx.save_for_backward(pc)
        ctx.cfg = (label, shape)
        return loss

    @_bwd
    def backward(ctx, gout):
        (pc,) = ctx.saved_tensors
        label, shape = ctx.cfg
        return ops.bce_bwd(pc, label, gout.contiguous()).reshape(shape), None, None


class BCETargetLossFn(Function):
    """nn.BCELoss against a target TENSOR (no gradient w.r.t. the target), image_translation.py:157-166."""

    @_fwd
    def forward(ctx, p, target):
        shape = p.shape
        loss, pc, tc = ops.bce_target_fwd(p.reshape(-1), target.detach().reshape(-1))
        ctx.save_for_backward(pc, tc)
        ctx.shape = shape
        return loss

    @_bwd
    def backward(ctx, gout):
        pc, tc = ctx.saved_tensors
        return ops.bce_target_bwd(pc, tc, gout.contiguous()).reshape(ctx.shape), None


class HingeEmbeddingLossFn(Function):
    """nn.HingeEmbeddingLoss(margin, mean) for targets in {+1, -1} (image_translation.py:141-142,269)."""

    @_fwd
    def forward(ctx, x, y, margin):
        loss, xd, yd = ops.hinge_fwd(x, y.detach(), margin)
        ctx.save_for_backward(xd, yd)
        ctx.margin = margin
        ctx.in_shape = x.shape
        return loss

    @_bwd
    def backward(ctx, gout):
        xd, yd = ctx.saved_tensors
        return ops.hinge_bwd(xd, yd, ctx.margin, gout.contiguous()), None, None


class FeatureMatchFn(Function):
    """One layer of get_fm_loss: mean((real.mean(0) - fake.mean(0))**2)."""

    @_fwd
    def forward(ctx, real, fake, out=None):
        loss, diff, rd, fd = ops.fm_fwd(real, fake, out)
        ctx.save_for_backward(diff, rd, fd)
        return loss

    @_bwd
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

    @_fwd
    def forward(ctx, lossvec, nfm, rate, arch, which, idx, *slots):
        out = ops.loss_mix_fwd(lossvec, nfm, rate, arch)
        ctx.cfg = (lossvec.numel(), nfm, rate, arch, which, idx)
        ctx.set_materialize_grads(False)
        return tuple(out.unbind(0))

    @_bwd
    def backward(ctx, *gouts):
        nslots, nfm, rate, arch, which, idx = ctx.cfg
        gout = gouts[which]
        if gout is None:
            return (None,) * (6 + len(idx))
        gv = ops.loss_mix_bwd(gout.contiguous(), nslots, nfm, rate, arch, which)
        return (None, None, None, None, None, None) + tuple(gv[i] for i in idx)


# ---- grouped launches (round 4): one autograd node for the same layer of several passes -------------------------------------------
# ``g`` problems = the A-side and the B-side pass of a pair (image_translation.py:342-346: G_B(A) | G_A(B), G_A(AB) | G_B(BA);
# :353-361: D_A(.) | D_B(.)), or a discriminator layer's real and fake pass of both sides (g = 4, problems ordered
# D_A real, D_A fake, D_B real, D_B fake: the two passes through one module are consecutive, which is what lets the kernels add
# their parameter gradients / running-statistics updates in the order separate launches would).  Inputs and outputs are flat tuples,
# problem-major.  Each problem's numbers are bitwise those of the one-problem Function (tests/test_group_gpu.py).
def _all_or_none(grads, who):
    """Gradients of a grouped node arrive for all problems or for none (the trainer groups passes with the same liveness)."""
    live = [g_ is not None for g_ in grads]
    if any(live) and not all(live):
        raise RuntimeError(f"{who}: gradients arrived for some problems of a grouped node only")
    return all(live)


def _flat_grads_of(params):
    fgs = [_flat_grad_of(p) for p in params]
    if any(f is None for f in fgs):
        raise RuntimeError("grouped launches need parameters whose .grad is the optimiser's flat gradient view (optim.Adam)")
    return fgs


class ConvGroupFn(Function):
    """Interior Conv2d (transposed=False) or ConvTranspose2d (True) of ``g`` problems.  args: x_0..x_{g-1}, w_0..w_{g-1}."""

    @_fwd
    def forward(ctx, g, stride, pad, transposed, *xw):
        xs = [ops.as_nhwc(x) for x in xw[:g]]
        ws = list(xw[g:])
        ctx.g, ctx.sp, ctx.tr = g, (stride, pad), transposed
        ctx.wrefs = ws
        ctx.final = FINAL_PASS
        ctx.save_for_backward(*xs, *ws)
        if transposed:
            hin, win = xs[0].shape[2], xs[0].shape[3]
            hw = ((hin - 1) * stride - 2 * pad + 4, (win - 1) * stride - 2 * pad + 4)
            ys = ops.conv_dgrad_g(xs, ws, hw, stride, pad)
        else:
            ys = ops.conv_fwd_g(xs, ws, stride, pad)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @_bwd
    def backward(ctx, *dys):
        g = ctx.g
        if not _all_or_none(dys, "ConvGroupFn"):
            return (None,) * (4 + 2 * g)
        saved = ctx.saved_tensors
        xs, ws = list(saved[:g]), ctx.wrefs
        stride, pad = ctx.sp
        dys = [ops.as_nhwc(d) for d in dys]
        need_x = [ctx.needs_input_grad[4 + i] for i in range(g)]
        need_w = [ctx.needs_input_grad[4 + g + i] for i in range(g)]
        if any(need_x) != all(need_x) or any(need_w) != all(need_w):
            raise RuntimeError("ConvGroupFn: the problems of a grouped node must need the same gradients")
        dxs = [None] * g
        if need_x[0]:
            dxs = ops.conv_fwd_g(dys, ws, stride, pad) if ctx.tr else ops.conv_dgrad_g(dys, ws, (xs[0].shape[2], xs[0].shape[3]), stride, pad)
        if need_w[0]:
            fgs = _flat_grads_of(ws)
            share = ops._share_of(ws)
            if ctx.tr:      # dw[cin][r][s][cout] = sum x[..cin] * dy[..cout]: conv wgrad with roles (dy := x, x := dy)
                ops.conv_wgrad_g(xs, dys, stride, pad, fgs, True, share)
            else:
                ops.conv_wgrad_g(dys, xs, stride, pad, fgs, True, share)
            _final(ctx.final, *ws)
        return (None, None, None, None) + tuple(dxs) + (None,) * g


class ConvC3GroupFn(Function):
    """First layer Conv2d(3, K, 4, 2, 1) + fused LeakyReLU of ``g`` problems.  args: x_0.., w_0.."""

    @_fwd
    def forward(ctx, g, act, slope, *xw):
        xs, ws = list(xw[:g]), list(xw[g:])
        ys = ops.c3_fwd_g(xs, ws, act, slope)
        ctx.g, ctx.act = g, (act, slope)
        ctx.wrefs = ws
        ctx.final = FINAL_PASS
        ctx.save_for_backward(*xs, *ws, *ys)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @_bwd
    def backward(ctx, *dys):
        g = ctx.g
        if not _all_or_none(dys, "ConvC3GroupFn"):
            return (None,) * (3 + 2 * g)
        saved = ctx.saved_tensors
        xs, ws, ys = list(saved[:g]), ctx.wrefs, list(saved[2 * g:])
        act, slope = ctx.act
        need_x = [ctx.needs_input_grad[3 + i] for i in range(g)]
        need_w = [ctx.needs_input_grad[3 + g + i] for i in range(g)]
        if any(need_x) != all(need_x) or any(need_w) != all(need_w):
            raise RuntimeError("ConvC3GroupFn: the problems of a grouped node must need the same gradients")
        dxs = [None] * g
        if need_x[0] or act == ops.ACT_NONE:
            gs = ops.act_bwd_g(dys, ys, act, slope) if act != ops.ACT_NONE else [ops.as_nhwc(d) for d in dys]
            fuse = {}
        else:       # weight gradient only (image input): the activation backward rides in the wgrad kernel's dy loads
            gs = [ops.as_nhwc(d) for d in dys]
            fuse = dict(act_outs=ys, act=act, slope=slope)
        if need_x[0]:
            dxs = ops.c3_dgrad_g(gs, ws, ops.ACT_NONE)
        if need_w[0]:
            fgs = _flat_grads_of(ws)
            ops.c3_wgrad_g(gs, xs, fgs, True, share=ops._share_of(ws), **fuse)
            _final(ctx.final, *ws)
        return (None, None, None) + tuple(dxs) + (None,) * g


class ConvTransposeC3GroupFn(Function):
    """Last layer ConvTranspose2d(K, 3, 4, 2, 1) + Sigmoid of ``g`` problems.  args: x_0.., w_0.."""

    @_fwd
    def forward(ctx, g, act, *xw):
        xs = [ops.as_nhwc(x) for x in xw[:g]]
        ws = list(xw[g:])
        outs = ops.c3_dgrad_g(xs, ws, act)
        ctx.g, ctx.act = g, act
        ctx.wrefs = ws
        ctx.final = FINAL_PASS
        ctx.save_for_backward(*xs, *ws, *outs)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @_bwd
    def backward(ctx, *douts):
        g = ctx.g
        if not _all_or_none(douts, "ConvTransposeC3GroupFn"):
            return (None,) * (2 + 2 * g)
        saved = ctx.saved_tensors
        xs, ws, outs = list(saved[:g]), ctx.wrefs, list(saved[2 * g:])
        gs = ops.act_bwd_g(douts, outs, ctx.act) if ctx.act != ops.ACT_NONE else [d.contiguous() for d in douts]
        need_x = [ctx.needs_input_grad[2 + i] for i in range(g)]
        need_w = [ctx.needs_input_grad[2 + g + i] for i in range(g)]
        if any(need_x) != all(need_x) or any(need_w) != all(need_w):
            raise RuntimeError("ConvTransposeC3GroupFn: the problems of a grouped node must need the same gradients")
        dxs = [None] * g
        if need_x[0]:
            dxs = ops.c3_fwd_g(gs, ws, ops.ACT_NONE)
        if need_w[0]:
            fgs = _flat_grads_of(ws)
            ops.c3_wgrad_g(xs, gs, fgs, True, share=ops._share_of(ws))
            _final(ctx.final, *ws)
        return (None, None) + tuple(dxs) + (None,) * g


class BatchNormActGroupFn(Function):
    """Training-mode BatchNorm2d + LeakyReLU / ReLU of ``g`` problems.  args: y_0.., gamma_0.., beta_0.., running_mean_0..,
    running_var_0.., num_batches_tracked_0..  Problems through the SAME module (consecutive) update its running statistics one after
    the other and add their parameter gradients in problem order (share)."""

    @_fwd
    def forward(ctx, g, eps, momentum, act, slope, *t):
        ys = [ops.as_nhwc(y) for y in t[:g]]
        gammas, betas = list(t[g:2 * g]), list(t[2 * g:3 * g])
        rms, rvs, nbts = list(t[3 * g:4 * g]), list(t[4 * g:5 * g]), list(t[5 * g:6 * g])
        share = ops._share_of(gammas)
        saved = ops.bn_train_stats_g(ys, rms, rvs, nbts, eps, momentum, share)
        zs = ops.bn_act_fwd_g(ys, saved, gammas, betas, act, slope)
        ctx.g, ctx.cfg, ctx.share = g, (act, slope), share
        ctx.prefs = (gammas, betas)
        ctx.final = FINAL_PASS
        ctx.save_for_backward(*ys, *saved, *gammas, *betas)
        ctx.set_materialize_grads(False)
        return tuple(zs)

    @_bwd
    def backward(ctx, *dzs):
        g = ctx.g
        if not _all_or_none(dzs, "BatchNormActGroupFn"):
            return (None,) * (5 + 6 * g)
        sv = ctx.saved_tensors
        ys, saved = list(sv[:g]), list(sv[g:2 * g])
        gammas, betas = ctx.prefs
        act, slope = ctx.cfg
        need_p = [ctx.needs_input_grad[5 + g + i] or ctx.needs_input_grad[5 + 2 * g + i] for i in range(g)]
        if any(need_p) != all(need_p):
            raise RuntimeError("BatchNormActGroupFn: the problems of a grouped node must need the same gradients")
        dg = db = None
        if need_p[0]:
            dg, db = _flat_grads_of(gammas), _flat_grads_of(betas)
        dys = ops.bn_act_bwd_g(dzs, ys, saved, gammas, betas, act, slope, dg, db, True, ctx.share if need_p[0] else 1)
        if need_p[0]:
            _final(ctx.final, *gammas, *betas)
        return (None,) * 5 + tuple(dys) + (None,) * (5 * g)


class ActGroupFn(Function):
    """Stand-alone activation (the discriminators' Sigmoid) of ``g`` problems."""

    @_fwd
    def forward(ctx, g, act, slope, *xs):
        outs = ops.act_fwd_g(list(xs), act, slope)
        ctx.g, ctx.cfg = g, (act, slope)
        ctx.save_for_backward(*outs)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @_bwd
    def backward(ctx, *dys):
        if not _all_or_none(dys, "ActGroupFn"):
            return (None,) * (3 + ctx.g)
        act, slope = ctx.cfg
        return (None, None, None) + tuple(ops.act_bwd_g(dys, list(ctx.saved_tensors), act, slope))


class MSELossGroupFn(Function):
    """nn.MSELoss of ``g`` (x, target) pairs; the losses land in ``outs`` (slots of the trainer's loss vector).  args: x_0.., t_0.., out_0.."""

    @_fwd
    def forward(ctx, g, *t):
        xs, ts, outs = list(t[:g]), list(t[g:2 * g]), list(t[2 * g:3 * g])
        xd, td = ops.mse_fwd_g(xs, ts, outs)
        ctx.g = g
        ctx.save_for_backward(*xd, *td)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @_bwd
    def backward(ctx, *gouts):
        g = ctx.g
        if not _all_or_none(gouts, "MSELossGroupFn"):
            return (None,) * (1 + 3 * g)
        sv = ctx.saved_tensors
        dxs = ops.mse_bwd_g(list(sv[:g]), list(sv[g:]), [go.contiguous() for go in gouts])
        return (None,) + tuple(dxs) + (None,) * (2 * g)


class BCELossGroupFn(Function):
    """nn.BCELoss of ``g`` probability vectors against constant labels (image_translation.py:157-166).  args: p_0.., out_0.."""

    @_fwd
    def forward(ctx, g, labels, *t):
        ps, outs = list(t[:g]), list(t[g:2 * g])
        ctx.shapes = [p.shape for p in ps]
        pcs = ops.bce_fwd_g([p.reshape(-1) for p in ps], labels, outs)
        ctx.g, ctx.labels = g, tuple(labels)
        ctx.save_for_backward(*pcs)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @_bwd
    def backward(ctx, *gouts):
        g = ctx.g
        live = [i for i in range(g) if gouts[i] is not None]
        if not live:
            return (None,) * (2 + 2 * g)
        pcs = ctx.saved_tensors
        dps = ops.bce_bwd_g([pcs[i] for i in live], [ctx.labels[i] for i in live], [gouts[i].contiguous() for i in live])
        out = [None] * g
        for i, d in zip(live, dps):
            out[i] = d.reshape(ctx.shapes[i])
        return (None, None) + tuple(out) + (None,) * g


class FeatureMatchGroupFn(Function):
    """One layer of get_fm_loss for ``g`` discriminators.  args: real_0.., fake_0.., out_0.."""

    @_fwd
    def forward(ctx, g, *t):
        reals, fakes, outs = list(t[:g]), list(t[g:2 * g]), list(t[2 * g:3 * g])
        diffs, rd, fd = ops.fm_fwd_g(reals, fakes, outs)
        ctx.g = g
        ctx.save_for_backward(*diffs, *rd, *fd)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @_bwd
    def backward(ctx, *gouts):
        g = ctx.g
        if not _all_or_none(gouts, "FeatureMatchGroupFn"):
            return (None,) * (1 + 3 * g)
        sv = ctx.saved_tensors
        need_r = any(ctx.needs_input_grad[1 + i] for i in range(g))
        need_f = any(ctx.needs_input_grad[1 + g + i] for i in range(g))
        dr, df = ops.fm_bwd_g(list(sv[:g]), list(sv[g:2 * g]), list(sv[2 * g:]), [go.contiguous() for go in gouts], need_r, need_f)
        return (None,) + tuple(dr or [None] * g) + tuple(df or [None] * g) + (None,) * g
