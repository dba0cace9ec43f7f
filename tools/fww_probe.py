#!/usr/bin/env python3
"""Probe of the window forward kernel (csrc/igemm_dma_x3_fww.hip): where its error against fp64 comes from (bf16-exact operands
leave only the hi x hi products) and what bounds its speed (operand loads dropped one at a time)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF
from discogan_modernized_amd import _lib, ops

DEV = "cuda"
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
def nhwc(t): return t.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
def krsc(t): return t.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
def rel(a, b): return ((a.double().cpu() - b).norm() / b.norm()).item()

def run(N, C, K, H, exact16):
    x, w = rnd(N, C, H, H, seed=1), rnd(K, C, 4, 4, seed=2, scale=1.0 / math.sqrt(16 * C))
    if exact16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    y64 = TF.conv2d(x.double(), w.double(), stride=2, padding=1)
    xg, wg = nhwc(x), krsc(w)
    e32 = rel(ops.conv_fwd(xg, wg, 2, 1), y64)
    _lib.set_option("bf16", 2)
    try:
        yreg = ops.conv_fwd(xg, wg, 2, 1)
        ops.X3 = True
        buf = torch.empty((3, wg.numel()), device=DEV, dtype=torch.bfloat16)
        wg._dg_x3, wg._dg_x3_ver = (buf, 0, torch.zeros_like(buf)), None
        y = ops.conv_fwd(xg, wg, 2, 1)
        torch.cuda.synchronize()
    finally:
        ops.X3 = False
        ops.planes_clear()
        _lib.set_option("bf16", 0)
    d = (y - yreg).abs()
    print(f"[{N},{C},{K},{H}] exact16={exact16}: fp32 {e32:.2e}  reg-x3 {rel(yreg, y64):.2e}  fww {rel(y, y64):.2e}  max|fww-reg|/max|y| {float(d.max() / y.abs().max()):.2e}  "
          f"frac differing {float((d > 0).float().mean()):.3f}")

for sh in [(1, 64, 128, 256), (2, 128, 64, 64), (1, 32, 8, 64)]:
    for ex in (False, True):
        run(*sh, ex)
