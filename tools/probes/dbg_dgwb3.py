import sys, math, torch
sys.path.insert(0, ".")
from discogan_modernized_amd import ops, _lib
torch.manual_seed(0)
N, C, K, H = 1, 64, 64, 64
w = torch.randn(K, C, 4, 4)
_lib.set_option("bf16", 1)
ops.SHADOW = True
def sh(t):
    t16 = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.preserve_format)
    ops.f32_to_bf16(t, t16); ops.shadow_put(t, t16); t._dg_bf16, t._dg_bf16_ver = t16, t._version
    return t
wg = sh(ops.krsc_param(w.cuda()))
wr = w.bfloat16().float()
for k0 in (2, 3, 4, 5, 6, 11, 15, 19, 23, 27):
    dy = torch.zeros(N, K, H // 2, H // 2)
    dy[0, k0, 10, 12] = 1.0
    dyg = sh(dy.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2))
    _lib.set_option("no_dma", 1); a = ops.conv_dgrad(dyg, wg, (H, H), 2, 1).cpu()
    _lib.set_option("no_dma", 0); b = ops.conv_dgrad(dyg, wg, (H, H), 2, 1).cpu()
    # which k does the window kernel's output correspond to?  out[0, c, 2*10-1+r, 2*12-1+s] = w[k, c, r, s]
    patch = b[0, :, 19:23, 23:27]            # [C, 4, 4]
    best = min(range(K), key=lambda k: (patch - wr[k]).abs().max().item())
    nz = (b.abs() > 1e-6).nonzero()
    print(f"k0={k0}: err {(a-b).abs().max().item():.3f}; window output matches w[k={best}] (err {(patch - wr[best]).abs().max().item():.3f}); nonzero rows {sorted(set(nz[:,2].tolist()))[:8]} cols {sorted(set(nz[:,3].tolist()))[:8]}")
    ops.shadow_clear()
