#!/usr/bin/env python3
"""In-kernel timing of one implicit-GEMM conv launch (dg_debug_igemm_stamps): where the ~160 us of a 17-GFLOP
64 px layer go.  usage: python tools/igemm_stamps.py [fwd|dgrad|wgrad] [C K H N]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from discogan_modernized_amd import ops, _lib

op = sys.argv[1] if len(sys.argv) > 1 else "fwd"
C, K, H, N = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (64, 128, 32, 256)
MODE = sys.argv[6] if len(sys.argv) > 6 else "f32"     # f32 | bf16 (fp32 tensors, bf16 tiles) | x3 | x3p (x3 with plane operands: igemm_dma_x3.hip) | dma (bf16 tensors in and out: LDS-DMA kernel where the shape allows)
dev = "cuda"
_lib.set_option("bf16", {"f32": 0, "bf16": 1, "x3": 2, "x3p": 2, "dma": 1}[MODE])
ops.X3 = MODE == "x3p"
adt = torch.bfloat16 if MODE == "dma" else torch.float32
x = ops.empty_nhwc(N, C, H, H, dev, adt).normal_()
w = ops.krsc_param(torch.randn(K, C, 4, 4, device=dev) * 0.05)
dy = ops.empty_nhwc(N, K, H // 2, H // 2, dev, adt).normal_()
if MODE == "dma":
    ops.SHADOW = ops.ACT16 = True
    w16 = torch.empty_like(w, dtype=torch.bfloat16, memory_format=torch.preserve_format)
    ops.f32_to_bf16(w, w16)
    w._dg_bf16, w._dg_bf16_ver = w16, w._version
fn = {"fwd": lambda: ops.conv_fwd(x, w, 2, 1), "dgrad": lambda: ops.conv_dgrad(dy, w, (H, H), 2, 1),
      "wgrad": lambda: ops.conv_wgrad(dy, x, 2, 1)}[op]
for _ in range(5):
    fn()
torch.cuda.synchronize()
buf = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
L = _lib.load()
L.dg_debug_igemm_stamps(buf.data_ptr(), buf.numel() * 8)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record()
torch.cuda.synchronize()
L.dg_debug_igemm_stamps(None, 0)
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 1] != 0]
n = len(s)
wall0, wall1 = s[:, 0], s[:, 5]
t0 = wall0.min()
us = lambda ticks: ticks / 100.0          # 100 MHz constant clock
cyc = s[:, 7] - s[:, 1]
ghz = (cyc / np.maximum(us(wall1 - wall0), 1e-9)).mean() / 1e3
print(f"{op} C={C} K={K} H={H} N={N}: {n} workgroups, event time {e0.elapsed_time(e1)*1e3:.1f} us, "
      f"first start -> last end {us(wall1.max() - t0):.1f} us, shader clock ~{ghz:.2f} GHz")
print(f"  start skew (us): p50 {np.percentile(us(wall0 - t0), 50):.1f}  p90 {np.percentile(us(wall0 - t0), 90):.1f}  max {us(wall0.max() - t0):.1f}")
print(f"  end   (us after first start): min {us(wall1.min() - t0):.1f}  p50 {np.percentile(us(wall1 - t0), 50):.1f}  max {us(wall1.max() - t0):.1f}")
for name, a, b in (("prologue", 1, 2), ("K loop", 2, 3), ("epilogue issue", 3, 4), ("store drain", 4, 7)):
    d = (s[:, b] - s[:, a]) / (ghz * 1e3)
    print(f"  {name:15s} mean {d.mean():7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
hw = s[:, 6] & 0xFFFFFFFF
xcc = s[:, 6] >> 32
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 16 + cu
uniq, cnt = np.unique(key, return_counts=True)
print(f"  placement: {len(uniq)} distinct (XCC,SE,CU) slots; workgroups per slot: min {cnt.min()} max {cnt.max()}; per XCC {np.bincount(xcc.astype(int)).tolist()}")
