import sys, math, torch
sys.path.insert(0, ".")
from discogan_modernized_amd import ops, _lib
torch.manual_seed(0)
N, C, K, H = 1, 64, 64, 64
w = torch.randn(K, C, 4, 4) / math.sqrt(16 * C)
_lib.set_option("bf16", 1)
ops.SHADOW = True
def sh(t):
    t16 = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.preserve_format)
    ops.f32_to_bf16(t, t16); ops.shadow_put(t, t16); t._dg_bf16, t._dg_bf16_ver = t16, t._version
    return t
wg = sh(ops.krsc_param(w.cuda()))
for name, lo, hi in (("only chunk 0", 0, 32), ("only chunk 1", 32, 64), ("k 32..39", 32, 40), ("k 40..47", 40, 48), ("k 48..63", 48, 64)):
    dy = torch.zeros(N, K, H // 2, H // 2)
    dy[:, lo:hi] = torch.randn(N, hi - lo, H // 2, H // 2)
    dyg = sh(dy.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2))
    _lib.set_option("no_dma", 1); a = ops.conv_dgrad(dyg, wg, (H, H), 2, 1).cpu()
    _lib.set_option("no_dma", 0); b = ops.conv_dgrad(dyg, wg, (H, H), 2, 1).cpu()
    print(name, "err", (a - b).abs().max().item(), "max", a.abs().max().item(), "window max", b.abs().max().item())
    ops.shadow_clear()
