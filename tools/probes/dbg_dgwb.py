import sys, math, torch
sys.path.insert(0, ".")
import torch.nn.functional as TF
from discogan_modernized_amd import ops, _lib
torch.manual_seed(0)
N, C, K, H = (int(v) for v in sys.argv[1:5])
r = lambda t: t.bfloat16().float()
w = torch.randn(K, C, 4, 4) / math.sqrt(16 * C)
dy = torch.randn(N, K, H // 2, H // 2)
ref = TF.conv_transpose2d(r(dy).double(), r(w).double(), stride=2, padding=1).float()
_lib.set_option("bf16", 1)
ops.SHADOW = True
wg = ops.krsc_param(w.cuda())
dyg = dy.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
for t in (wg, dyg):
    t16 = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.preserve_format)
    ops.f32_to_bf16(t, t16); ops.shadow_put(t, t16); t._dg_bf16, t._dg_bf16_ver = t16, t._version
_lib.set_option("no_dma", 1)
dx0 = ops.conv_dgrad(dyg, wg, (H, H), 2, 1).cpu()
_lib.set_option("no_dma", 0)
dx = ops.conv_dgrad(dyg, wg, (H, H), 2, 1).cpu()
err = (dx - dx0).abs()
print("window vs register-staged: max err", err.max().item(), "max", dx0.abs().max().item())
for ph in (0, 1):
    for pw in (0, 1):
        e = err[:, :, ph::2, pw::2]
        print("class", ph, pw, "max", e.max().item(), "frac bad", (e > 1e-4).float().mean().item())
for n in range(N):
    e = err[n].amax(dim=0)
    bad = e > 1e-4
    print("image", n, "bad rows:", bad.any(1).nonzero().flatten().tolist()[:48], "bad cols:", bad.any(0).nonzero().flatten().tolist()[:24])
ec = err.amax(dim=(0, 2, 3)); print("bad channels:", (ec > 1e-4).nonzero().flatten().tolist()[:70])
if len(sys.argv) > 5:
    _lib.set_option("splitk", int(sys.argv[5]))
    dxs = ops.conv_dgrad(dyg, wg, (H, H), 2, 1).cpu()
    print("splitk", sys.argv[5], "window vs register-staged:", (dxs - dx0).abs().max().item())
