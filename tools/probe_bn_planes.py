import sys, os
sys.path.insert(0, os.getcwd())
import torch
from discogan_modernized_amd import ops, _lib
from tools.bench_ops import timeit
dev="cuda"
for (N,H,C) in [(32,256,64),(32,128,128),(32,64,256)]:
    y = ops.empty_nhwc(N, C, H, H, dev).normal_()
    t3 = torch.empty((3, y.numel()), device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.f32_to_bf16x3(y, t3))
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    saved = ops.bn_train_stats(y, None, None, None, 1e-5, 0.1)
    L=_lib.load()
    z = torch.empty_like(y)
    st = torch.cuda.current_stream().cuda_stream
    M=N*H*H
    t_po = timeit(lambda: _lib.check(L.dg_bn_act_fwd_x3(y.data_ptr(), None, t3.data_ptr(), t3.stride(0), 0, M, C, saved.data_ptr(), g.data_ptr(), b.data_ptr(), ops.ACT_LEAKY, 0.2, st), "x"))
    t_cm = timeit(lambda: _lib.check(L.dg_bn_act_fwd_x3(y.data_ptr(), None, t3.data_ptr(), t3.stride(0), 1, M, C, saved.data_ptr(), g.data_ptr(), b.data_ptr(), ops.ACT_LEAKY, 0.2, st), "x"))
    t_f = timeit(lambda: _lib.check(L.dg_bn_act_fwd(y.data_ptr(), z.data_ptr(), M, C, saved.data_ptr(), g.data_ptr(), b.data_ptr(), ops.ACT_LEAKY, 0.2, st), "x"))
    t_zp = timeit(lambda: _lib.check(L.dg_bn_act_fwd_x3(y.data_ptr(), z.data_ptr(), t3.data_ptr(), t3.stride(0), 0, M, C, saved.data_ptr(), g.data_ptr(), b.data_ptr(), ops.ACT_LEAKY, 0.2, st), "x"))
    e = y.numel()
    print(f"[{N}x{H}x{H}x{C}] split {t:.3f} ms ({10*e/t/1e9:.2f} TB/s) | bn planes-only pm {t_po:.3f} ({10*e/t_po/1e9:.2f}) cm {t_cm:.3f} ({10*e/t_cm/1e9:.2f}) | bn fp32 {t_f:.3f} ({8*e/t_f/1e9:.2f}) | bn fp32+planes {t_zp:.3f} ({14*e/t_zp/1e9:.2f})")
