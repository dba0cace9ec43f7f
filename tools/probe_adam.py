#!/usr/bin/env python3
"""Adam kernel alone at the generator group's size (460 M parameters) and at 48 M: plain / + bf16 shadow / + three planes (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from discogan_modernized_amd import ops
from tools.bench_ops import timeit
dev = "cuda"
for P in (48 << 20, 460_385_936):
    p, g, m, v = (torch.randn(P, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    p16 = torch.empty(P, device=dev, dtype=torch.bfloat16)
    PE = (P + 7) // 8 * 8
    p3 = torch.empty((3, PE), device=dev, dtype=torch.bfloat16)
    state = torch.zeros(8, device=dev, dtype=torch.float64)
    ops.adam_advance(state, 2e-4, 0.5, 0.999)
    t0 = timeit(lambda: ops.adam_step_flat(p, g, m, v, state, 0.5, 0.999, 1e-8, 1e-5), iters=5)
    t1 = timeit(lambda: ops.adam_step_flat(p, g, m, v, state, 0.5, 0.999, 1e-8, 1e-5, p16=p16), iters=5)
    t3 = timeit(lambda: ops.adam_step_flat(p, g, m, v, state, 0.5, 0.999, 1e-8, 1e-5, p3=(p3.data_ptr(), p3.stride(0))), iters=5)
    print(f"{P / 1e6:.0f} M params: plain {t0:.3f} ms ({28 * P / t0 / 1e9:.2f} TB/s)  +bf16 {t1:.3f} ms ({30 * P / t1 / 1e9:.2f})  +planes {t3:.3f} ms ({34 * P / t3 / 1e9:.2f})")
    del p, g, m, v, p16, p3
