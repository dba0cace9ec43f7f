"""N>1 path on CPU: world_size-2 gloo runs of the data-parallel exchange (discogan_modernized_amd/dp.py)
against the single-process emulation of DDP semantics (oracle.dp_emulated_iteration).

The arithmetic in these tests is the ORACLE's (the HIP kernels need a GPU); what is under test is the
exchange step the GPU trainer uses verbatim: one flat buffer for the stepped side, sum over ranks,
1/W folded into the optimiser step, rank-local BatchNorm statistics, per-rank batches, no buffer
broadcast."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from discogan_modernized_amd import dp
from oracle import discogan_ref as O

S, N, W = 16, 4, 2


def _flat_views(params):
    n = sum(p.numel() for p in params)
    flat = torch.zeros(n)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    return flat


def _worker(rank, world, initfile, outdir):
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        st = O.build_state(image_size=S, seed=1234)                # identical replicas from the seed
        A, B = O.synthetic_batch(N, S, seed=dp.rank_data_seed(rank))
        res = []
        for it in range(3):
            dstep = O.is_dis_step(it, st.args)
            live = ("dis_A", "dis_B") if dstep else ("gen_A", "gen_B")
            params = [p for k in live for p in st.nets[k].parameters()]
            for net in st.nets.values():
                net.zero_grad()
            flat = _flat_views(params)                              # .grad = views of one flat buffer
            out = O.forward_losses(st, A, B, it)
            (out.dis_loss if dstep else out.gen_loss).backward()
            scale, _ = dp.all_reduce_flat(flat, None)
            assert scale == 1.0 / world
            flat.mul_(scale)                                        # the GPU path folds this into Adam
            (st.optim_dis if dstep else st.optim_gen).step()
            res.append(O.losses_to_floats(out))
        torch.save(dict(losses=res, params={k: [p.detach().clone() for p in n.parameters()] for k, n in st.nets.items()},
                        bufs={k: {bn: b.clone() for bn, b in n.named_buffers()} for k, n in st.nets.items()}),
                   os.path.join(outdir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_sequential_emulation():
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(W, initfile, d), nprocs=W, join=True)
        r = [torch.load(os.path.join(d, f"rank{k}.pt")) for k in range(W)]
    # replicas stay identical across ranks after every step
    for k in r[0]["params"]:
        for a, b in zip(r[0]["params"][k], r[1]["params"][k]):
            assert torch.equal(a, b), f"replicas diverged in {k}"
    # ... and equal the sequential emulation (per-rank BN stats, grads averaged, one Adam step)
    st = O.build_state(image_size=S, seed=1234)
    shards = [O.synthetic_batch(N, S, seed=dp.rank_data_seed(k)) for k in range(W)]
    for it in range(3):
        out0 = O.dp_emulated_iteration(st, [s[0] for s in shards], [s[1] for s in shards], it)
        ref = O.losses_to_floats(out0)
        for key, v in ref.items():
            assert abs(r[0]["losses"][it][key] - v) <= 1e-5 * abs(v) + 1e-7, (it, key)
    for k, net in st.nets.items():
        for a, p in zip(r[0]["params"][k], net.parameters()):
            # Adam's first steps are ~lr*sign(g): a gradient element that is ~0 can flip sign between
            # two reduction orders (worker processes run 2 OMP threads), moving that weight by <= 2*lr
            # per step.  Everything else must agree to rounding.
            d = (a - p.detach()).abs()
            assert float(d.max()) <= 3 * 2 * 2e-4, f"{k}: DP result != emulation (max {float(d.max()):.2e})"
            assert float((d > 2e-6).float().mean()) < 2e-3, f"{k}: too many weights differ from the emulation"
        for bn, b in net.named_buffers():                           # rank-0 buffers are authoritative
            assert torch.allclose(r[0]["bufs"][k][bn].float(), b.float(), rtol=1e-3, atol=1e-4), (k, bn)
    # BatchNorm statistics are rank-local: rank 1 saw other data, so its running stats differ
    diff = any(not torch.equal(r[0]["bufs"]["gen_A"][bn], r[1]["bufs"]["gen_A"][bn])
               for bn in r[0]["bufs"]["gen_A"] if "running_mean" in bn)
    assert diff


def test_single_process_is_identity():
    flat = torch.arange(8.0)
    scale, work = dp.all_reduce_flat(flat.clone(), None)
    assert scale == 1.0 and work is None
    assert dp.rank_data_seed(3) == 1003
