#!/usr/bin/env python3
"""Do a saturating MFMA conv kernel and an HBM-bound streaming kernel run BESIDE each other on two HIP streams, or take turns?

    python tools/probe_corun.py [--arith x3|bf16] [--layer 3]

The conv kernels hold 2 waves x 217-240 VGPRs per SIMD and 96-128 KB of LDS: a second kernel's wave is only resident on the same
CU when its VGPR allocation fits what is left (512 - 2 x conv).  The probe times, on the 512 px / batch 32 layer shapes,
  conv alone (R launches on stream A), streaming kernel alone (Q launches on stream B), both together,
for streaming kernels of different register footprints (act_fwd: 19 VGPRs; Adam: 49-55; ATen add), and prints
  together / (conv + streaming)  -- 1.0 = they take turns;   together / max(conv, streaming)  -- 1.0 = fully hidden.
A tuning aid, not the bench.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from discogan_modernized_amd import _lib, ops  # noqa: E402
from discogan_modernized_amd.model import stage_channels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arith", default="x3")
    ap.add_argument("--layer", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--prio", type=int, default=0, help="1: the conv stream with high priority")
    ap.add_argument("--understory", type=int, default=0, help="experiments build (DG_LIB=.../libdiscogan_hip_experiments.so): act_fwd as the LDS-DMA-fed low-register kernel with this many 1-KiB pieces in flight per wave (16 | 8 | 4)")
    a = ap.parse_args()
    dev = "cuda"
    L = _lib.load()
    ctx = ops.Context()
    if a.arith == "x3":
        ctx.prec, ctx.x3 = ops.PREC_F32X3, True
    elif a.arith == "bf16":
        ctx.prec, ctx.shadow = ops.PREC_BF16, True
    ops.use(ctx).__enter__()
    ch = stage_channels(a.size)
    i = a.layer
    C, K, H = ch[i - 1], ch[i], a.size >> i
    N = a.batch
    x = ops.empty_nhwc(N, C, H, H, dev).normal_()
    w = ops.empty_krsc(K, C, dev).normal_()
    dy = ops.empty_nhwc(N, K, H // 2, H // 2, dev).normal_()
    if a.arith == "bf16":
        for t in (x, w, dy):
            t16 = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.preserve_format)
            ops.f32_to_bf16(t, t16)
            ops.shadow_put(t, t16)
            t._dg_bf16, t._dg_bf16_ver = t16, t._version
    gf = 2.0 * N * (H // 2) ** 2 * K * C * 16 / 1e9
    convs = {
        "fwd": lambda: ops.conv_fwd(x, w, 2, 1),
        "dgrad": lambda: ops.conv_dgrad(dy, w, (H, H), 2, 1),
        "wgrad": lambda: ops.conv_wgrad(dy, x, 2, 1),
    }
    # streaming kernels
    n = 64 << 20
    sx = torch.randn(n, device=dev)
    sy = torch.empty_like(sx)
    P = 48 << 20
    p, g, m, v = (torch.randn(P, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    state = torch.zeros(8, device=dev, dtype=torch.float64)
    ops.adam_advance(state, 2e-4, 0.5, 0.999)
    if a.understory < 0:
        _lib.set_option("understory", a.understory)          # negative: the plain kernel with at most -value workgroups (the product caps at 2048)
    if a.understory > 0:
        _lib.set_option("understory", a.understory)
        for nn in (n, 4096, 1 << 20, (1 << 20) + 4, 1000 * 4, 1027 * 4):          # the experimental kernel against the plain one, ragged tails included
            xx, yy = sx[:nn], torch.zeros(nn + 64, device=dev)
            _lib.check(L.dg_act_fwd(xx.data_ptr(), yy.data_ptr(), nn, ops.ACT_LEAKY, 0.2, torch.cuda.current_stream().cuda_stream), "act")
            assert torch.equal(yy[:nn], torch.nn.functional.leaky_relu(xx, 0.2)) and float(yy[nn:].abs().max()) == 0.0, nn
        print("# understory kernel: results equal to leaky_relu, nothing written past the end")

    def s_act():
        _lib.check(L.dg_act_fwd(sx.data_ptr(), sy.data_ptr(), n, ops.ACT_LEAKY, 0.2, torch.cuda.current_stream().cuda_stream), "act")

    def s_adam():
        ops.adam_step_flat(p, g, m, v, state, 0.5, 0.999, 1e-8, 1e-5)

    def s_add():
        torch.add(sx, sx, out=sy)

    streams = {"act_fwd (19 VGPRs, 8 B/elem)": (s_act, 8.0 * n), "adam (49 VGPRs, 28 B/param)": (s_adam, 28.0 * P), "aten add (12 B/elem)": (s_add, 12.0 * n)}
    sa, sb = torch.cuda.Stream(priority=-1 if a.prio else 0), torch.cuda.Stream()

    def run(fa, ra, fb, rb):
        """ms of ra launches of fa on stream A beside rb launches of fb on stream B"""
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(2_000_000)         # hold the device while the host enqueues
            e0.record()
            sa.wait_event(e0)
            sb.wait_event(e0)
            k = max(ra, rb)
            for j in range(k):                   # interleave the enqueue so neither queue starves
                if j < ra:
                    with torch.cuda.stream(sa):
                        fa()
                if j < rb:
                    with torch.cuda.stream(sb):
                        fb()
            with torch.cuda.stream(sa):
                ea.record()
            with torch.cuda.stream(sb):
                eb.record()
            torch.cuda.synchronize()
            best = min(best, max(e0.elapsed_time(ea), e0.elapsed_time(eb)))
        return best

    nop = lambda: None
    print(f"# {a.arith} layer {i}: {C}->{K} @{H}, batch {N}, {gf:.1f} GFLOP per conv launch")
    for cname, cf in convs.items():
        for _ in range(3):
            cf()
        tc = run(cf, a.reps, nop, 0)
        print(f"{cname}: alone {tc / a.reps * 1e3:.1f} us per launch = {gf / (tc / a.reps):.0f} TFLOP/s")
        for sname, (sf, nbytes) in streams.items():
            for _ in range(2):
                sf()
            t1 = run(nop, 0, sf, 4) / 4
            q = max(1, int(round(0.6 * tc / t1)))        # streaming work = 60 % of the conv work's duration
            ts = run(nop, 0, sf, q)
            tb = run(cf, a.reps, sf, q)
            print(f"   + {sname:30s}: {nbytes / t1 / 1e9:5.2f} TB/s alone, x{q}: conv {tc:7.3f} ms, stream {ts:7.3f} ms, together {tb:7.3f} ms "
                  f"-> together / sum {tb / (tc + ts):.3f}, hidden fraction of the shorter {(tc + ts - tb) / min(tc, ts):.2f}")


if __name__ == "__main__":
    main()
