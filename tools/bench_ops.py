#!/usr/bin/env python3
"""Per-layer kernel timing at the BASELINE shapes (not the headline bench; a tuning aid).

    python tools/bench_ops.py --size 64 --batch 256 [--kt 16] [--splitk 0]
Prints ms and TFLOP/s (2*MACs) for conv fwd / dgrad / wgrad of every interior layer + edge kernels.
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from discogan_modernized_amd import _lib, ops  # noqa: E402
from discogan_modernized_amd.model import stage_channels  # noqa: E402


ITERS = 10


def timeit(fn, iters=None, warm=3):
    iters = iters or ITERS
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--kt", type=int, default=0)
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--target_wgs", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bf16", type=int, default=0)
    ap.add_argument("--shadow", type=int, default=0, help="with --bf16 1: hand the conv kernels bf16 shadow operands (what the trainer does)")
    ap.add_argument("--act16", type=int, default=0, help="time the BatchNorm / edge kernels on bf16-stored feature maps instead of the fp32 ones")
    ap.add_argument("--dma_mfma", type=int, default=0, help="32: the LDS-DMA kernel's 32x32x16 body (default 16x16x32)")
    ap.add_argument("--no_dma", type=int, default=0, help="keep bf16-operand convs on the register-staged tiles")
    ap.add_argument("--x3planes", type=int, default=0, help="with --bf16 2: plane operands (igemm_dma_x3.hip), what the f32x3 trainer does; 2 = forward on the transposed weight planes")
    ap.add_argument("--x3_mfma", type=int, default=0, help="16: the plane kernel's 16x16x32 body with planes paired along k (default 32x32x16)")
    ap.add_argument("--x3cm", type=int, default=0, help="with --x3planes: hand the window input-grad kernel (and the weight-grad) QUAD-CHUNK gradient planes, what the BatchNorm kernels write for it (ops.X3_CM)")
    ap.add_argument("--layers", default="", help="comma list of layer indices (1-based) to time; default all")
    ap.add_argument("--dbg_zero", type=int, default=0, help="timing experiment: drop the A (1) / B (2) / both (3) operand loads of the conv kernels")
    a = ap.parse_args()
    global ITERS
    ITERS = a.iters
    _lib.set_option("bf16", a.bf16)
    if a.dbg_zero:          # exists in the timing build only (make -C discogan_modernized_amd/csrc TIMING=1; DG_LIB=.../libdiscogan_hip_timing.so)
        _lib.set_option("dbg_zero", a.dbg_zero)
    _lib.set_option("no_dma", a.no_dma)
    _lib.set_option("dma_mfma", a.dma_mfma)
    if a.x3_mfma:           # exists in the experiments build only (make EXPERIMENTS=1)
        _lib.set_option("x3_mfma", a.x3_mfma)
    ops.SHADOW = bool(a.shadow)
    ops.X3 = bool(a.x3planes)
    only = {int(v) for v in a.layers.split(",") if v}

    def shadowed(t):
        if a.shadow:
            t16 = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.preserve_format)
            ops.f32_to_bf16(t, t16)
            ops.shadow_put(t, t16)
            t._dg_bf16, t._dg_bf16_ver = t16, t._version
        return t
    _lib.set_option("kt", a.kt)
    _lib.set_option("splitk", a.splitk)
    _lib.set_option("target_wgs", a.target_wgs)
    dev = "cuda"
    ch = stage_channels(a.size)
    N, S = a.batch, a.size
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = 0.0
    print(f"# size {S} batch {N} kt {a.kt} splitk {a.splitk}")
    print(f"{'layer':28s} {'GFLOP':>9s} | {'fwd ms':>8s} {'TF/s':>6s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s}")
    h = S // 2
    for i in range(1, len(ch)):
        C, K, H = ch[i - 1], ch[i], h
        if only and i not in only:
            h //= 2
            continue
        x = shadowed(ops.empty_nhwc(N, C, H, H, dev).normal_())
        w = shadowed(ops.empty_krsc(K, C, dev).normal_())
        dy = shadowed(ops.empty_nhwc(N, K, H // 2, H // 2, dev).normal_())
        gf = 2.0 * N * (H // 2) ** 2 * K * C * 16 / 1e9
        if a.x3planes == 2:      # with the transposed weight copy (weights of a flat Adam group have one)
            buf = torch.empty((3, w.numel()), device=dev, dtype=torch.bfloat16)
            w._dg_x3, w._dg_x3_ver = (buf, 0, torch.zeros_like(buf)), None
        if a.x3cm and ops.x3_window_dgrad(N, H, H, C, K):
            t3_ = torch.empty((3, dy.numel()), device=dev, dtype=torch.bfloat16)
            ops.f32_to_bf16x3(dy, t3_)
            ops.planes_put(dy, t3_.view(3, N * (H // 2) ** 2 // 4, 4, K // 16, 16).permute(0, 1, 3, 2, 4).contiguous().view(3, -1), cm=True)
        t1 = timeit(lambda: ops.conv_fwd(x, w, 2, 1))
        t2 = timeit(lambda: ops.conv_dgrad(dy, w, (H, H), 2, 1))
        t3 = timeit(lambda: ops.conv_wgrad(dy, x, 2, 1))
        amb = (x.numel() + w.numel() + dy.numel()) * 4 / 1e6
        print(f"conv s2 {C:4d}->{K:4d} @{H:3d} [{amb:7.1f} MB] {gf:9.2f} | {t1:8.3f} {gf / t1:6.1f} | {t2:8.3f} {gf / t2:6.1f} | {t3:8.3f} {gf / t3:6.1f}")
        tot["fwd"] += t1; tot["dgrad"] += t2; tot["wgrad"] += t3; totf += gf
        h //= 2
    if only and not a.act16:
        print(f"interior totals: {totf:.1f} GFLOP each dir | fwd {tot['fwd']:.3f} ms ({totf / max(tot['fwd'], 1e-9):.1f} TF/s) dgrad {tot['dgrad']:.3f} ms ({totf / max(tot['dgrad'], 1e-9):.1f}) wgrad {tot['wgrad']:.3f} ms ({totf / max(tot['wgrad'], 1e-9):.1f})")
        return
    ops.SHADOW = False
    # heads
    C = ch[-1]
    for K in (100, 1):
        x = ops.empty_nhwc(N, C, 4, 4, dev).normal_()
        w = ops.empty_krsc(K, C, dev).normal_()
        dy = ops.empty_nhwc(N, K, 1, 1, dev).normal_()
        gf = 2.0 * N * K * C * 16 / 1e9
        t1 = timeit(lambda: ops.conv_fwd(x, w, 1, 0))
        t2 = timeit(lambda: ops.conv_dgrad(dy, w, (4, 4), 1, 0))
        t3 = timeit(lambda: ops.conv_wgrad(dy, x, 1, 0))
        print(f"head    {C:4d}->{K:4d} @  4       {gf:9.2f} | {t1:8.3f} {gf / t1:6.1f} | {t2:8.3f} {gf / t2:6.1f} | {t3:8.3f} {gf / t3:6.1f}")
    # edge
    K = ch[0]
    x = torch.rand(N, 3, S, S, device=dev)
    w = torch.randn(K, 3, 4, 4, device=dev)
    dy = ops.empty_nhwc(N, K, S // 2, S // 2, dev).normal_()
    gf = 2.0 * N * (S // 2) ** 2 * K * 48 / 1e9
    t1 = timeit(lambda: ops.c3_fwd(x, w, ops.ACT_LEAKY, 0.2))
    t2 = timeit(lambda: ops.c3_dgrad(dy, w, ops.ACT_SIGMOID))
    t3 = timeit(lambda: ops.c3_wgrad(dy, x))
    mb = (x.numel() + dy.numel()) * 4 / 1e6
    print(f"edge c3    3->{K:4d} @{S:3d}       {gf:9.2f} | {t1:8.3f} {gf / t1:6.1f} | {t2:8.3f} {gf / t2:6.1f} | {t3:8.3f} {gf / t3:6.1f}   ({mb:.0f} MB -> {mb / t1 / 1e3:.2f}/{mb / t2 / 1e3:.2f}/{mb / t3 / 1e3:.2f} TB/s)")
    if a.act16:
        # bf16-stored feature maps: every BN shape of the nets through the typed kernels, and the edge kernels with a bf16 NHWC side
        for C, H in [(ch[0], S // 2)] + [(ch[i], S // (4 << (i - 1))) for i in range(1, len(ch))]:
            if H < 1:
                continue
            y = ops.empty_nhwc(N, C, H, H, dev, torch.bfloat16).normal_()
            dz = ops.empty_nhwc(N, C, H, H, dev, torch.bfloat16).normal_()
            g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            saved = ops.bn_train_stats(y, None, None, None, 1e-5, 0.1)
            mbt = y.numel() * 2 / 1e6
            t1 = timeit(lambda: ops.bn_train_stats(y, None, None, None, 1e-5, 0.1))
            t2 = timeit(lambda: ops.bn_act_fwd(y, saved, g, b, ops.ACT_LEAKY, 0.2))
            t3 = timeit(lambda: ops.bn_act_bwd(dz, y, saved, g, b, ops.ACT_LEAKY, 0.2))
            print(f"bn16 [{N}x{H}x{H}x{C}] {mbt:.0f} MB: stats {t1:.3f} ms ({mbt / t1 / 1e3:.2f} TB/s)  apply {t2:.3f} ms ({2 * mbt / t2 / 1e3:.2f} TB/s)  bwd {t3:.3f} ms ({5 * mbt / t3 / 1e3:.2f} TB/s)")
        ops.ACT16 = True
        dy16 = ops.empty_nhwc(N, K, S // 2, S // 2, dev, torch.bfloat16).normal_()
        t1 = timeit(lambda: ops.c3_fwd(x, w, ops.ACT_LEAKY, 0.2))
        t2 = timeit(lambda: ops.c3_dgrad(dy16, w, ops.ACT_SIGMOID))
        t3 = timeit(lambda: ops.c3_wgrad(dy16, x))
        ops.ACT16 = False
        mb1, mb2 = x.numel() * 4 / 1e6, dy16.numel() * 2 / 1e6
        print(f"edge c3 bf16 side: fwd {t1:.3f} ms ({(mb1 + mb2) / t1 / 1e3:.2f} TB/s)  dgrad {t2:.3f} ms ({(mb1 + mb2) / t2 / 1e3:.2f} TB/s)  wgrad {t3:.3f} ms ({(mb1 + mb2) / t3 / 1e3:.2f} TB/s)")
        return
    # BN + act streaming, every BN shape of the nets (decoder 64ch @S/2 ... bottleneck)
    for C, H in [(ch[0], S // 2)] + [(ch[i], S // (4 << (i - 1))) for i in range(1, len(ch))]:
        if H < 1:
            continue
        y = ops.empty_nhwc(N, C, H, H, dev).normal_()
        dz = ops.empty_nhwc(N, C, H, H, dev).normal_()
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        saved = ops.bn_train_stats(y, None, None, None, 1e-5, 0.1)
        mbt = y.numel() * 4 / 1e6
        t1 = timeit(lambda: ops.bn_train_stats(y, None, None, None, 1e-5, 0.1))
        t2 = timeit(lambda: ops.bn_act_fwd(y, saved, g, b, ops.ACT_LEAKY, 0.2))
        t3 = timeit(lambda: ops.bn_act_bwd(dz, y, saved, g, b, ops.ACT_LEAKY, 0.2))
        print(f"bn [{N}x{H}x{H}x{C}] {mbt:.0f} MB: stats {t1:.3f} ms ({mbt / t1 / 1e3:.2f} TB/s)  apply {t2:.3f} ms ({2 * mbt / t2 / 1e3:.2f} TB/s)  bwd {t3:.3f} ms ({5 * mbt / t3 / 1e3:.2f} TB/s)")
    print(f"interior totals: {totf:.1f} GFLOP each dir | fwd {tot['fwd']:.3f} ms ({totf / tot['fwd']:.1f} TF/s) dgrad {tot['dgrad']:.3f} ms ({totf / tot['dgrad']:.1f}) wgrad {tot['wgrad']:.3f} ms ({totf / tot['wgrad']:.1f})")


if __name__ == "__main__":
    main()
