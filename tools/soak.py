"""Determinism / stability soak: N iterations (default 64 px / batch 256) in every dispatch mode of one matrix path; all runs must end
bitwise identical and finite.    python tools/soak.py [f32|bf16|f32x3|bf16a] [iterations] [image_size] [batch]
(bf16a = bf16 MFMA + bf16-stored feature maps; from 256 px f32x3 runs on plane operands: quad-chunk planes, plane-only BatchNorm outputs,
statistics from the conv kernels)"""
import sys, torch, time
sys.path.insert(0, ".")
from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args, synthetic_batch
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
act = "f32"
if dtype == "bf16a":
    dtype, act = "bf16", "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 450
size = int(sys.argv[3]) if len(sys.argv) > 3 else 64
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 256
res = []
modes = [dict(use_graph=True), dict(use_graph=False), dict(use_graph=False, two_streams=False),
         dict(use_graph=False, overlap_comm=True, comm="capi", bucket_mb=4.0), dict(use_graph=True, overlap_comm=False, comm="capi")]
for kw in modes:
    tr = DiscoGANTrainer(default_args(), device="cuda", image_size=size, seed=1234, mfma_dtype=dtype, act_dtype=act, **kw)
    A, B = synthetic_batch(batch, size, 1000, "cuda")
    t0 = time.time()
    for it in range(iters):
        tr.train_iteration(A, B, it)
    tr.finish(); torch.cuda.synchronize()
    f = tr.losses_to_floats(tr.train_iteration(A, B, iters))
    cs = (tr.optim_gen.flat_p.double().sum().item(), tr.optim_dis.flat_p.double().sum().item(), tr.optim_gen.flat_p.abs().max().item())
    print(f"{dtype} {kw}: {time.time()-t0:.1f}s  checksums {cs}  gen_loss {f['gen_loss']:.6f} dis_loss {f['dis_loss']:.6f}", flush=True)
    res.append(cs)
    tr.close()
assert all(r == res[0] for r in res), "non-deterministic!"
print(f"{iters}-iteration soak ({dtype}, {size} px, batch {batch}): all {len(modes)} dispatch modes bitwise identical, finite:", all(abs(v) < 1e30 for v in res[0]))
