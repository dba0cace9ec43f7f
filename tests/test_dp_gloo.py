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
        xg = dp.ExchangeGroup(dist.group.WORLD)
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
            scale = xg.all_reduce_sum_(flat)                       # the transport object the GPU trainer uses
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


def _xg_worker(rank, world, initfile, outdir):
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        xg = dp.ExchangeGroup(dist.group.WORLD)                      # gloo -> the c10d transport
        assert xg.transport == "c10d" and xg.world == world and "gloo" in xg.describe()
        flat = torch.arange(10.0) * (rank + 1)
        scale = xg.all_reduce_sum_(flat[2:8])                         # a bucket = a slice of the flat buffer
        b = torch.full((4,), float(rank))
        xg.broadcast_(b, 0)
        xg.barrier()
        torch.save(dict(flat=flat, scale=scale, b=b), os.path.join(outdir, f"xg{rank}.pt"))
        xg.close()
    finally:
        dist.destroy_process_group()


def test_exchange_group_c10d_two_ranks():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_xg_worker, args=(W, os.path.join(d, "init"), d), nprocs=W, join=True)
        r = [torch.load(os.path.join(d, f"xg{k}.pt")) for k in range(W)]
    for k in range(W):
        want = torch.arange(10.0) * (k + 1)
        want[2:8] = torch.arange(10.0)[2:8] * 3                      # (1 + 2) x, only inside the bucket
        assert torch.equal(r[k]["flat"], want) and r[k]["scale"] == 0.5
        assert torch.equal(r[k]["b"], torch.zeros(4))


def test_distributed_indices_equal_torch_distributed_sampler():
    """dp.distributed_indices is DistributedSampler(shuffle=True) + set_epoch, bit for bit
    (distributed_image_translation.py:203-208,451-452): disjoint 1/W shards from ONE permutation per epoch."""
    from torch.utils.data import DistributedSampler
    for n, w in [(10, 4), (1000, 8), (7, 3), (3, 8), (64, 2)]:
        for epoch in (0, 1, 5):
            shards = []
            for r in range(w):
                ds = DistributedSampler(range(n), num_replicas=w, rank=r, shuffle=True)
                ds.set_epoch(epoch)
                got = dp.distributed_indices(n, w, r, epoch)
                assert got == list(ds), (n, w, epoch, r)
                shards.append(got)
            assert len({len(s) for s in shards}) == 1
            if n % w == 0:
                assert sorted(sum(shards, [])) == list(range(n))       # a partition of the dataset


def test_cli_epoch_batches_shard_the_dataset():
    from types import SimpleNamespace
    from discogan_modernized_amd import image_translation as it
    args = SimpleNamespace(batch_size=4)
    seen = []
    for r in range(2):
        bs = it.epoch_batches(args, 3, 22, r, 2, None, "cpu")
        assert len(bs) == it.batches_per_epoch(args, 22, 2) == 3 and [len(a) for a, _ in bs] == [4, 4, 3]
        assert all(torch.equal(a, b) for a, b in bs)                  # A_i paired with B_i (dataset.py:215-222)
        seen += [int(i) for a, _ in bs for i in a]
    assert sorted(seen) == list(range(22))
    # a trailing batch of ONE sample is skipped (train-mode BatchNorm needs two); single process: independent shuffles
    assert it.batches_per_epoch(args, 18, 2) == 2 and len(it.epoch_batches(args, 0, 18, 0, 2, None, "cpu")) == 2
    g = torch.Generator().manual_seed(0)
    one = it.epoch_batches(args, 0, 22, 0, 1, g, "cpu")
    assert len(one) == 5 and not all(torch.equal(a, b) for a, b in one)


def test_launcher_command_line():
    """launch.py mirrors distributed_training.sh's flags; world size = length of the --gpus list."""
    from discogan_modernized_amd import launch
    cmd, env, log = launch.build_command(["--gpus=0,1,4,5", "--task_name=celebA", "--style_A=Male", "--style_B=Smiling",
                                          "--batch_size=64", "--max_iters", "7", "--log_dir=/tmp/x"], {})
    assert "--nproc-per-node=4" in cmd and "--world_size=4" in cmd and env["HIP_VISIBLE_DEVICES"] == "0,1,4,5"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "discogan_modernized_amd.distributed_image_translation" in cmd
    for f in ("--task_name=celebA", "--style_A=Male", "--style_B=Smiling", "--batch_size=64", "--epochs=50", "--image_size=64", "--distributed"):
        assert f in cmd, f
    assert cmd[-2:] == ["--max_iters", "7"] and log == "/tmp/x/train.log"


def test_single_process_is_identity():
    flat = torch.arange(8.0)
    xg = dp.ExchangeGroup(None)                                      # no process group: world 1, c10d, nothing moves
    assert xg.transport == "c10d" and xg.all_reduce_sum_(flat) == 1.0 and torch.equal(flat, torch.arange(8.0))
    assert dp.host_allgather(2.5) == [2.5]
    assert dp.rank_data_seed(3) == 1003


# ---- guarded bootstrap of a communicator whose init is a blocking collective (dp.guarded_bootstrap) -----------------
# The real init (ncclCommInitRank via dg_dp_init) needs one GPU per rank; what is rehearsed here, over gloo's FileStore with
# stand-in prepare / init callables, is the protocol around it: the vote before the collective, the deadline inside it,
# and that every rank leaves non-zero when one of them fails or blocks.
def _boot_worker(rank, world, initfile, outdir, scenario):
    import time
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    entered = os.path.join(outdir, f"entered{rank}")

    def prepare():
        if scenario == "prepare_fails" and rank == 1:
            raise RuntimeError("cannot load librccl on this rank")
        return b"\x07" * 128 if rank == 0 else None

    def init(uid):
        open(entered, "w").write("1")
        assert uid == b"\x07" * 128
        if scenario == "init_hangs" and rank == 1:
            time.sleep(600)                                              # a rank blocked inside the collective
        if scenario == "init_raises" and rank == 1:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        return "comm"

    try:
        got = dp.guarded_bootstrap(dp.store_of(None), rank, world, prepare, init, tag=f"t/{scenario}", timeout_s=4.0)
        open(os.path.join(outdir, f"result{rank}"), "w").write(f"ok {got}")
        assert dp.host_allgather(10.0 + rank) == [10.0, 11.0]           # the store-based host collective, same run
    except dp.BootstrapVoteFailed as e:
        open(os.path.join(outdir, f"result{rank}"), "w").write(f"vote {e}")
        raise SystemExit(5)
    except dp._lib.DiscoganHipError as e:
        open(os.path.join(outdir, f"result{rank}"), "w").write(f"initerr {e}")
        raise SystemExit(6)
    dist.destroy_process_group()


def _run_boot(scenario):
    import time
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        ps = [ctx.Process(target=_boot_worker, args=(r, W, os.path.join(d, "init"), d, scenario)) for r in range(W)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(60)
        alive = [p.is_alive() for p in ps]
        for p in ps:
            if p.is_alive():
                p.kill()
        files = {f: open(os.path.join(d, f)).read() for f in os.listdir(d) if f.startswith(("result", "entered"))}
        return [p.exitcode for p in ps], alive, files, time.time() - t0


@pytest.mark.timeout(120)
def test_guarded_bootstrap_all_ranks_ready():
    codes, alive, files, _ = _run_boot("ok")
    assert codes == [0, 0] and not any(alive)
    assert files["result0"] == files["result1"] == "ok comm" and "entered0" in files and "entered1" in files


@pytest.mark.timeout(120)
def test_guarded_bootstrap_one_rank_not_ready_nobody_enters_the_collective():
    codes, alive, files, _ = _run_boot("prepare_fails")
    assert codes == [5, 5] and not any(alive)                           # the SAME verdict on every rank, non-zero
    assert "entered0" not in files and "entered1" not in files          # no rank entered the blocking init
    for r in range(W):
        assert files[f"result{r}"].startswith("vote") and "rank 1: RuntimeError: cannot load librccl" in files[f"result{r}"]


@pytest.mark.timeout(120)
def test_guarded_bootstrap_blocked_collective_exits_nonzero_within_the_deadline():
    codes, alive, files, dt = _run_boot("init_hangs")
    assert not any(alive) and dt < 40                                   # 4 s deadline per phase, not 600 s
    assert codes == [dp.HANG_EXIT_CODE, dp.HANG_EXIT_CODE]              # blocked rank AND the rank waiting for it
    assert "result0" not in files and "result1" not in files


@pytest.mark.timeout(120)
def test_guarded_bootstrap_init_error_is_raised_on_every_rank():
    codes, alive, files, _ = _run_boot("init_raises")
    assert codes == [6, 6] and not any(alive)
    for r in range(W):
        assert "rank 1: RuntimeError: ncclCommInitRank" in files[f"result{r}"]


def _capi_vote_worker(rank, world, initfile, outdir):
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        # no GPU in this process: dg_dp_ready fails on every rank -> the vote fails identically everywhere
        try:
            dp.ExchangeGroup(dist.group.WORLD, transport="capi")
            out = "built"
        except dp.BootstrapVoteFailed as e:
            out = f"vote {e}"
        open(os.path.join(outdir, f"r{rank}"), "w").write(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_exchange_group_capi_vote_fails_cleanly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("needs a process without a HIP device")
    os.environ["DG_COMM_INIT_TIMEOUT_S"] = "20"
    try:
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_capi_vote_worker, args=(W, os.path.join(d, "init"), d), nprocs=W, join=True)
            r = [open(os.path.join(d, f"r{k}")).read() for k in range(W)]
    finally:
        os.environ.pop("DG_COMM_INIT_TIMEOUT_S", None)
    assert all(x.startswith("vote") and "2 of 2 ranks not ready" in x for x in r), r
