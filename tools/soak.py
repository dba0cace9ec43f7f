import sys, torch, time
sys.path.insert(0, ".")
from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args, synthetic_batch
res = []
for rep in range(2):
    for graph in (True, False):
        tr = DiscoGANTrainer(default_args(), device="cuda", image_size=64, seed=1234, use_graph=graph)
        A, B = synthetic_batch(256, 64, 1000, "cuda")
        t0 = time.time()
        for it in range(450):
            out = tr.train_iteration(A, B, it, need_losses=(it % 50 == 0))
        tr.finish(); torch.cuda.synchronize()
        f = tr.losses_to_floats(tr.train_iteration(A, B, 450))
        cs = (tr.optim_gen.flat_p.double().sum().item(), tr.optim_dis.flat_p.double().sum().item(), tr.optim_gen.flat_p.abs().max().item())
        print(f"rep {rep} graph {graph}: {time.time()-t0:.1f}s  checksums {cs}  gen_loss {f['gen_loss']:.6f} dis_loss {f['dis_loss']:.6f}", flush=True)
        res.append(cs)
assert all(r == res[0] for r in res), "non-deterministic!"
print("450-iteration soak: all four runs bitwise identical, finite:", all(abs(v) < 1e30 for v in res[0]))
