"""Grouped launches (round 4): several problems of identical geometry in ONE launch per kernel.

The reference issues the passes of an iteration in independent pairs of identical shape -- G_B(A) | G_A(B), G_A(AB) | G_B(BA),
D_A(A) | D_B(B), D_A(BA) | D_B(AB) (image_translation.py:342-361) -- and every discriminator sees real and fake images with the same
weights (:353-354,360-361).  The grouped C-ABI entry points (dg_*_g) must give every problem BITWISE the result of its own one-problem
call, including the cases where consecutive problems accumulate into one tensor (a discriminator's real + fake pass: weight / BatchNorm
parameter gradients, running statistics).  Also here: the conv arithmetic as a per-call argument (two trainers of different arithmetic
interleaved call by call).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from discogan_modernized_amd import _lib, ops  # noqa: E402
from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args, synthetic_batch  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def nhwc(t):
    return t.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def same(a, b, what):
    assert a.shape == b.shape, what
    assert torch.equal(a, b), f"{what}: grouped result differs from the one-problem call (max |diff| {(a - b).abs().max().item():.3e})"


PRECS = [ops.PREC_F32, ops.PREC_F32X3, ops.PREC_BF16]
# (N, C, K, H): a split-K forward, an unsplit one, a 64-column input-grad tile, the deep 4x4 -> 2x2 layer
SHAPES = [(4, 64, 128, 16), (16, 64, 128, 32), (4, 128, 256, 8), (2, 256, 512, 4), (3, 128, 64, 16)]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("g", [2, 4])
@pytest.mark.parametrize("N,C,K,H", SHAPES)
def test_conv_group_is_bitwise_the_single_problem_calls(N, C, K, H, g, prec):
    xs = [nhwc(rnd(N, C, H, H, seed=10 + i)) for i in range(g)]
    ws = [ops.krsc_param(rnd(K, C, 4, 4, seed=20 + i, scale=1.0 / math.sqrt(16 * C)).to(DEV)) for i in range(g)]
    dys = [nhwc(rnd(N, K, H // 2, H // 2, seed=30 + i)) for i in range(g)]
    with ops.use(ops.Context(prec=prec, group_plan="single")):
        y1 = [ops.conv_fwd(x, w, 2, 1) for x, w in zip(xs, ws)]
        yg = ops.conv_fwd_g(xs, ws, 2, 1)
        d1 = [ops.conv_dgrad(d, w, (H, H), 2, 1) for d, w in zip(dys, ws)]
        dg = ops.conv_dgrad_g(dys, ws, (H, H), 2, 1)
        base = [rnd(K, C, 4, 4, seed=40 + i).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for i in range(g)]
        w1 = [ops.conv_wgrad(d, x, 2, 1, out=b.clone(), accumulate=True) for d, x, b in zip(dys, xs, base)]
        wg = [b.clone() for b in base]
        ops.conv_wgrad_g(dys, xs, 2, 1, wg, True)
    for i in range(g):
        same(yg[i], y1[i], f"forward, problem {i}")
        same(dg[i], d1[i], f"input-grad, problem {i}")
        same(wg[i], w1[i], f"weight-grad, problem {i}")


@pytest.mark.parametrize("prec", [ops.PREC_F32, ops.PREC_F32X3])
@pytest.mark.parametrize("N,C,K,H", [(4, 64, 128, 16), (2, 256, 512, 4), (64, 256, 512, 8)])
def test_conv_wgrad_shared_output_adds_in_problem_order(N, C, K, H, prec):
    """A discriminator's real and fake pass accumulate into ONE weight gradient (image_translation.py:353-361): four problems, two
    outputs; equal to the real call followed by the accumulating fake call."""
    g = 4
    xs = [nhwc(rnd(N, C, H, H, seed=50 + i)) for i in range(g)]
    dys = [nhwc(rnd(N, K, H // 2, H // 2, seed=60 + i)) for i in range(g)]
    base = [rnd(K, C, 4, 4, seed=70 + i).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for i in range(2)]
    with ops.use(ops.Context(prec=prec, group_plan="single")):
        ref = [b.clone() for b in base]
        for z in range(2):
            for j in range(2):
                ops.conv_wgrad(dys[2 * z + j], xs[2 * z + j], 2, 1, out=ref[z], accumulate=True)
        got = [b.clone() for b in base]
        ops.conv_wgrad_g(dys, xs, 2, 1, [got[0], got[0], got[1], got[1]], True, share=2)
        got0 = [torch.empty_like(b) for b in base]                      # accumulate = 0: the first member initialises
        ops.conv_wgrad_g(dys, xs, 2, 1, [got0[0], got0[0], got0[1], got0[1]], False, share=2)
        ref0 = [torch.empty_like(b) for b in base]
        for z in range(2):
            ops.conv_wgrad(dys[2 * z], xs[2 * z], 2, 1, out=ref0[z], accumulate=False)
            ops.conv_wgrad(dys[2 * z + 1], xs[2 * z + 1], 2, 1, out=ref0[z], accumulate=True)
    for z in range(2):
        same(got[z], ref[z], f"shared weight gradient {z}")
        same(got0[z], ref0[z], f"shared weight gradient {z}, not accumulating")


@pytest.mark.parametrize("g", [2, 4])
def test_heads_group(g):
    """The 4x4 valid heads: K = 1 (discriminator, plain reductions) and K = 100 (generator bottleneck), and the 1x1 -> 4x4 transposed head."""
    N, C = 8, 512
    xs = [nhwc(rnd(N, C, 4, 4, seed=80 + i)) for i in range(g)]
    ops.current().group_plan = "single"
    for K in (1, 100):
        ws = [ops.krsc_param(rnd(K, C, 4, 4, seed=90 + i, scale=0.02).to(DEV)) for i in range(g)]
        dys = [nhwc(rnd(N, K, 1, 1, seed=95 + i)) for i in range(g)]
        y1 = [ops.conv_fwd(x, w, 1, 0) for x, w in zip(xs, ws)]
        yg = ops.conv_fwd_g(xs, ws, 1, 0)
        d1 = [ops.conv_dgrad(d, w, (4, 4), 1, 0) for d, w in zip(dys, ws)]
        dg = ops.conv_dgrad_g(dys, ws, (4, 4), 1, 0)
        share = 2 if g == 4 else 1
        nout = g // share
        base = [torch.zeros(K, C, 4, 4, device=DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for _ in range(nout)]
        ref = [b.clone() for b in base]
        for i in range(g):
            ops.conv_wgrad(dys[i], xs[i], 1, 0, out=ref[i // share], accumulate=True)
        got = [b.clone() for b in base]
        ops.conv_wgrad_g(dys, xs, 1, 0, [got[i // share] for i in range(g)], True, share=share)
        for i in range(g):
            same(yg[i], y1[i], f"head K={K} forward {i}")
            same(dg[i], d1[i], f"head K={K} input-grad {i}")
        for z in range(nout):
            same(got[z], ref[z], f"head K={K} weight-grad {z}")
    ops.current().group_plan = "launch"


@pytest.mark.parametrize("prec", [ops.PREC_F32, ops.PREC_F32X3])
@pytest.mark.parametrize("g,share", [(2, 1), (4, 2)])
def test_edge_kernels_group(g, share, prec):
    N, S, K = 4, 32, 64
    imgs = [torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(100 + i)).to(DEV) for i in range(g)]
    ws = [rnd(K, 3, 4, 4, seed=110 + i, scale=0.1).to(DEV) for i in range(g)]
    dys = [nhwc(rnd(N, K, S // 2, S // 2, seed=120 + i)) for i in range(g)]
    with ops.use(ops.Context(prec=prec)):
        y1 = [ops.c3_fwd(x, w, ops.ACT_LEAKY, 0.2) for x, w in zip(imgs, ws)]
        yg = ops.c3_fwd_g(imgs, ws, ops.ACT_LEAKY, 0.2)
        d1 = [ops.c3_dgrad(d, w, ops.ACT_SIGMOID) for d, w in zip(dys, ws)]
        dg = ops.c3_dgrad_g(dys, ws, ops.ACT_SIGMOID)
        nout = g // share
        for fuse in (False, True):
            kw1 = [dict(act_out=y1[i], act=ops.ACT_LEAKY, slope=0.2) if fuse else {} for i in range(g)]
            ref = [torch.zeros(K, 3, 4, 4, device=DEV) for _ in range(nout)]
            for i in range(g):
                ops.c3_wgrad(dys[i], imgs[i], out=ref[i // share], accumulate=True, **kw1[i])
            got = [torch.zeros(K, 3, 4, 4, device=DEV) for _ in range(nout)]
            kwg = dict(act_outs=y1, act=ops.ACT_LEAKY, slope=0.2) if fuse else {}
            ops.c3_wgrad_g(dys, imgs, [got[i // share] for i in range(g)], True, share=share, **kwg)
            for z in range(nout):
                same(got[z], ref[z], f"c3 weight-grad {z} (fused activation backward: {fuse})")
    for i in range(g):
        same(yg[i], y1[i], f"c3 forward {i}")
        same(dg[i], d1[i], f"c3 input-grad {i}")


@pytest.mark.parametrize("g,share", [(2, 1), (4, 2), (4, 1)])
@pytest.mark.parametrize("N,C,H", [(4, 128, 16), (64, 512, 4), (8, 100, 1)])
def test_batchnorm_group_with_shared_module(N, C, H, g, share):
    """Statistics + apply + backward of g problems; with share = 2 problems 2z, 2z + 1 go through ONE module: its running statistics
    see two updates in order, its parameter gradients two additions in order."""
    nmod = g // share
    ys = [nhwc(rnd(N, C, H, H, seed=130 + i)) for i in range(g)]
    dzs = [nhwc(rnd(N, C, H, H, seed=140 + i)) for i in range(g)]
    gam = [(1.0 + 0.1 * rnd(C, seed=150 + z)).to(DEV) for z in range(nmod)]
    bet = [(0.1 * rnd(C, seed=160 + z)).to(DEV) for z in range(nmod)]

    def fresh():
        return ([torch.zeros(C, device=DEV) for _ in range(nmod)], [torch.ones(C, device=DEV) for _ in range(nmod)],
                [torch.zeros((), dtype=torch.long, device=DEV) for _ in range(nmod)],
                [0.01 * torch.ones(C, device=DEV) for _ in range(nmod)], [0.02 * torch.ones(C, device=DEV) for _ in range(nmod)])

    rm1, rv1, nb1, dg1, db1 = fresh()
    s1, z1, dy1 = [], [], []
    for i in range(g):
        m = i // share
        s1.append(ops.bn_train_stats(ys[i], rm1[m], rv1[m], nb1[m], 1e-5, 0.1))
        z1.append(ops.bn_act_fwd(ys[i], s1[i], gam[m], bet[m], ops.ACT_LEAKY, 0.2))
    for i in range(g):
        m = i // share
        dy1.append(ops.bn_act_bwd(dzs[i], ys[i], s1[i], gam[m], bet[m], ops.ACT_LEAKY, 0.2, out_grads=(dg1[m], db1[m]))[0])
    rm2, rv2, nb2, dg2, db2 = fresh()
    pick = lambda lst: [lst[i // share] for i in range(g)]       # noqa: E731
    s2 = ops.bn_train_stats_g(ys, pick(rm2), pick(rv2), pick(nb2), 1e-5, 0.1, share)
    z2 = ops.bn_act_fwd_g(ys, s2, pick(gam), pick(bet), ops.ACT_LEAKY, 0.2)
    dy2 = ops.bn_act_bwd_g(dzs, ys, s2, pick(gam), pick(bet), ops.ACT_LEAKY, 0.2, pick(dg2), pick(db2), True, share)
    for i in range(g):
        same(s2[i], s1[i], f"saved statistics {i}")
        same(z2[i], z1[i], f"apply {i}")
        same(dy2[i], dy1[i], f"backward {i}")
    for m in range(nmod):
        same(rm2[m], rm1[m], f"running_mean {m}")
        same(rv2[m], rv1[m], f"running_var {m}")
        assert int(nb2[m]) == int(nb1[m]) == share
        same(dg2[m], dg1[m], f"dgamma {m}")
        same(db2[m], db1[m], f"dbeta {m}")


def test_losses_and_activations_group():
    N, S = 8, 32
    xs = [torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(170 + i)).to(DEV) for i in range(2)]
    ts = [torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(180 + i)).to(DEV) for i in range(2)]
    o1 = [ops.mse_fwd(x, t)[0] for x, t in zip(xs, ts)]
    o2 = [torch.empty((), device=DEV) for _ in range(2)]
    ops.mse_fwd_g(xs, ts, o2)
    gout = [torch.full((), 0.3 + i, device=DEV) for i in range(2)]
    b1 = [ops.mse_bwd(x, t, go) for x, t, go in zip(xs, ts, gout)]
    b2 = ops.mse_bwd_g(xs, ts, gout)
    for i in range(2):
        same(o2[i], o1[i], f"mse {i}")
        same(b2[i], b1[i], f"mse backward {i}")
    ps = [torch.rand(N, generator=torch.Generator().manual_seed(190 + i)).to(DEV) for i in range(4)]
    ps[1][0], ps[2][1] = 0.0, 1.0                         # saturation: the -100 clamp and the 1e-12 guard
    labels = (1.0, 0.0, 1.0, 0.0)
    l1 = [ops.bce_fwd(p, lab)[0] for p, lab in zip(ps, labels)]
    l2 = [torch.empty((), device=DEV) for _ in range(4)]
    ops.bce_fwd_g(ps, labels, l2)
    go4 = [torch.full((), 0.5 + 0.1 * i, device=DEV) for i in range(4)]
    g1 = [ops.bce_bwd(p, lab, go) for p, lab, go in zip(ps, labels, go4)]
    g2 = ops.bce_bwd_g(ps, labels, go4)
    for i in range(4):
        same(l2[i], l1[i], f"bce {i}")
        same(g2[i], g1[i], f"bce backward {i}")
    reals = [nhwc(rnd(N, 128, 8, 8, seed=200 + i)) for i in range(2)]
    fakes = [nhwc(rnd(N, 128, 8, 8, seed=210 + i)) for i in range(2)]
    f1 = [ops.fm_fwd(r, f) for r, f in zip(reals, fakes)]
    fo = [torch.empty((), device=DEV) for _ in range(2)]
    diffs, rd, fd = ops.fm_fwd_g(reals, fakes, fo)
    fb1 = [ops.fm_bwd(f1[i][1], f1[i][2], f1[i][3], gout[i], False, True)[1] for i in range(2)]
    fb2 = ops.fm_bwd_g(diffs, rd, fd, gout, False, True)[1]
    for i in range(2):
        same(fo[i], f1[i][0], f"feature matching {i}")
        same(diffs[i], f1[i][1], f"feature matching diff {i}")
        same(fb2[i], fb1[i], f"feature matching backward {i}")
    zs = [rnd(N, 1, 1, 1, seed=220 + i).to(DEV) for i in range(4)]
    a1 = [ops.act_fwd(z, ops.ACT_SIGMOID) for z in zs]
    a2 = ops.act_fwd_g(zs, ops.ACT_SIGMOID)
    d1 = [ops.act_bwd(z, a, ops.ACT_SIGMOID) for z, a in zip(zs, a1)]
    d2 = ops.act_bwd_g(zs, a2, ops.ACT_SIGMOID)
    for i in range(4):
        same(a2[i], a1[i], f"sigmoid {i}")
        same(d2[i], d1[i], f"sigmoid backward {i}")


def _run(tr, A, B, iters):
    vals = [tr.losses_to_floats(tr.train_iteration(A, B, it)) for it in range(iters)]
    tr.finish()
    torch.cuda.synchronize()
    state = {f"{n}.{k}": v.detach().clone() for n, net in tr.nets.items() for k, v in net.state_dict().items()}
    return vals, state


@pytest.mark.parametrize("S,N,mfma,graph,streams", [(16, 4, "f32", False, True), (64, 8, "f32", True, True), (64, 8, "f32x3", True, True),
                                                    (64, 8, "f32x3", False, False), (128, 2, "f32", False, True)])
def test_grouped_training_step_is_bitwise_the_ungrouped_one(S, N, mfma, graph, streams):
    """The whole D,G,G,D,G,G,D schedule with every pair of passes (and, in the D-steps, every discriminator layer's real + fake pass of
    both sides) as grouped launches: logged losses, every parameter, every BatchNorm buffer bitwise equal to the two-chain schedule."""
    A, B = synthetic_batch(N, S, 5, DEV)
    ref = DiscoGANTrainer(default_args(), device=DEV, image_size=S, seed=1234, mfma_dtype=mfma, use_graph=graph, two_streams=streams,
                          group_launch=False)
    v1, s1 = _run(ref, A, B, 7)
    del ref
    tr = DiscoGANTrainer(default_args(), device=DEV, image_size=S, seed=1234, mfma_dtype=mfma, use_graph=graph, two_streams=streams,
                         group_launch=True, group_plan="single")
    assert tr.group_launch
    v2, s2 = _run(tr, A, B, 7)
    for it, (a, b) in enumerate(zip(v1, v2)):
        assert a == b, f"iteration {it}: {a} vs {b}"
    for k in s1:
        assert torch.equal(s1[k], s2[k]), f"{k} differs after 7 iterations"


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("g", [2, 4])
@pytest.mark.parametrize("N,C,K,H", SHAPES)
def test_conv_group_planned_for_the_whole_launch(N, C, K, H, g, prec):
    """Default plan of a grouped launch: split-K sized for all problems together (fewer slabs each) -- the same products in another fixed
    summation order: equal to the one-problem results at fp32 rounding, and deterministic."""
    xs = [nhwc(rnd(N, C, H, H, seed=10 + i)) for i in range(g)]
    ws = [ops.krsc_param(rnd(K, C, 4, 4, seed=20 + i, scale=1.0 / math.sqrt(16 * C)).to(DEV)) for i in range(g)]
    dys = [nhwc(rnd(N, K, H // 2, H // 2, seed=30 + i)) for i in range(g)]
    tol = 2e-2 if prec == ops.PREC_BF16 else 2e-5
    with ops.use(ops.Context(prec=prec)):
        assert ops.current().group_plan == "launch"
        y1 = [ops.conv_fwd(x, w, 2, 1) for x, w in zip(xs, ws)]
        d1 = [ops.conv_dgrad(d, w, (H, H), 2, 1) for d, w in zip(dys, ws)]
        w1 = [ops.conv_wgrad(d, x, 2, 1) for d, x in zip(dys, xs)]
        for rep in range(2):
            yg = ops.conv_fwd_g(xs, ws, 2, 1)
            dg = ops.conv_dgrad_g(dys, ws, (H, H), 2, 1)
            wg = [torch.zeros_like(w) for w in w1]
            ops.conv_wgrad_g(dys, xs, 2, 1, wg, True)
            if rep == 0:
                first = (yg, dg, wg)
    for i in range(g):
        for got, ref, again, what in ((yg[i], y1[i], first[0][i], "forward"), (dg[i], d1[i], first[1][i], "input-grad"), (wg[i], w1[i], first[2][i], "weight-grad")):
            assert torch.equal(got, again), f"{what} {i}: not deterministic"
            err = (got - ref).abs().max().item()
            assert err <= tol * ref.abs().max().item(), f"{what} {i}: {err:.3e} vs max {ref.abs().max().item():.3e}"


def test_grouped_step_with_launch_plans_tracks_the_ungrouped_step():
    """The default grouped schedule (split-K planned per launch) against the ungrouped one over a D,G,G cycle from the same weights:
    losses within 1e-5 relative (another summation order of the same products)."""
    S, N = 64, 16
    A, B = synthetic_batch(N, S, 5, DEV)
    ref = DiscoGANTrainer(default_args(), device=DEV, image_size=S, seed=1234, group_launch=False)
    tr = DiscoGANTrainer(default_args(), device=DEV, image_size=S, seed=1234, group_launch=True)
    assert tr.ctx.group_plan == "launch"
    for it in range(3):
        a = ref.losses_to_floats(ref.train_iteration(A, B, it))
        b = tr.losses_to_floats(tr.train_iteration(A, B, it))
        tol = 1e-5 if it == 0 else 2e-3           # (later iterations start from weights that differ in the last bits)
        for k in a:
            assert abs(a[k] - b[k]) <= tol * abs(a[k]) + 1e-7, f"iteration {it} {k}: {a[k]} vs {b[k]}"


def test_grouping_defaults():
    assert DiscoGANTrainer(default_args(), device=DEV, image_size=64, seed=1).group_launch
    assert DiscoGANTrainer(default_args(), device=DEV, image_size=64, seed=1, mfma_dtype="f32x3").group_launch
    assert not DiscoGANTrainer(default_args(), device=DEV, image_size=64, seed=1, mfma_dtype="bf16").group_launch
    assert not DiscoGANTrainer(default_args(model_arch="gan"), device=DEV, image_size=64, seed=1).group_launch
    with pytest.raises(ValueError):
        DiscoGANTrainer(default_args(model_arch="recongan"), device=DEV, image_size=16, seed=1, group_launch=True)


@pytest.mark.parametrize("grouped", [False, True])
def test_interleaved_trainers_with_different_arithmetic(grouped):
    """The conv arithmetic is an argument of every C-ABI call and a field of the trainer's own ops.Context ("dtype
    enum", "no global mutable state"): an exact-fp32 and an f32x3 trainer stepped alternately in one process each reproduce their solo
    run bitwise -- and differ from each other."""
    S, N = 32, 4
    A, B = synthetic_batch(N, S, 9, DEV)

    def make(m):
        return DiscoGANTrainer(default_args(), device=DEV, image_size=S, seed=1234, mfma_dtype=m, group_launch=grouped)

    solo = {}
    for m in ("f32", "f32x3"):
        solo[m] = _run(make(m), A, B, 6)
    t1, t2 = make("f32"), make("f32x3")
    v = {"f32": [], "f32x3": []}
    for it in range(6):
        v["f32"].append(t1.losses_to_floats(t1.train_iteration(A, B, it)))
        v["f32x3"].append(t2.losses_to_floats(t2.train_iteration(A, B, it)))
    torch.cuda.synchronize()
    for m, t in (("f32", t1), ("f32x3", t2)):
        assert v[m] == solo[m][0], m
        for n, net in t.nets.items():
            for k, p in net.state_dict().items():
                assert torch.equal(p, solo[m][1][f"{n}.{k}"]), f"{m}: {n}.{k}"
    assert v["f32"] != v["f32x3"]
    assert _lib.load().dg_set_option(b"bf16", 0) == 0          # (the process default was never touched)
