"""Re-hosted training CLI (reference image_translation.py:21-81, 211-435).

Same flags and defaults (``--task_name --model_arch --image_size --batch_size --epochs
--learning_rate --beta1 --beta2 --gan_curriculum --starting_rate --default_rate --update_interval
--log_interval --model_save_interval ...``), same log-line format (:394-398) and checkpoint file
names (``gen_A_{iters}.pth`` ... ``*_final.pth``, :420-432).  The training loop dispatches into the
HIP kernels through ``DiscoGANTrainer``.

Data sources (``--data_source``):
  files      the reference's own layout under ``--data_root`` (default ./datasets; dataset.py:14-22): file lists per task
             (dataset.get_data), PIL decode in worker threads, uint8 over PCIe on a copy stream, crop / erosion / resize /
             normalise on the device (dataset.DeviceLoader + dg_image_prep), double-buffered against the training step
  shards     ``--shard_A/--shard_B``: pre-decoded uint8 [n,H,W,3] .npy files (dataset.write_shard), memory-mapped, same device path
  tensors    ``--data_A/--data_B``: ``torch.save``d float tensors [n,3,S,S] in [0,1], or uint8 tensors [n,S,S,3] of decoded image
             rows that stay uint8 in HBM and are normalised / re-laid-out per batch on the device (dg_u8hwc_to_f32chw)
  synthetic  uniform tensors (``--synthetic_size`` images per domain): what the benchmark metric is defined on
  auto       tensors if --data_A/--data_B are given, shards if --shard_A/--shard_B, files if the task's directory exists, else synthetic

Batch order.  Single process: A and B are shuffled independently every epoch (shuffle_data, dataset.py:24-35),
``data_size // batch_size`` batches.  Data parallel: the ``DistributedSampler`` contract of
distributed_image_translation.py:203-216,451-452 -- ONE permutation per epoch from a seed shared by all ranks
(``set_epoch``), rank r takes ``perm[r::W]``, A_i is paired with B_i (dataset.py:215-222), the last batch of the
shard may be short (no drop_last); a final batch of a single sample is skipped (train-mode BatchNorm needs two).

    python -m discogan_modernized_amd.image_translation --task_name edges2shoes --image_size 64 --batch_size 256
"""
from __future__ import annotations

import argparse
import os
import time
from datetime import datetime
from pathlib import Path

import torch

from .trainer import DiscoGANTrainer

TASKS = ["facescrub", "celebA", "edges2shoes", "edges2handbags", "handbags2shoes", "tops2hanbok", "hanbok2tops"]


def build_parser(description="HIP/MI355X implementation of the DiscoGAN training path"):
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--task_name", type=str, default="facescrub")
    p.add_argument("--results_dir", type=str, default="./results/")
    p.add_argument("--models_dir", type=str, default="./models/")
    p.add_argument("--model_arch", type=str, default="discogan", choices=["discogan", "recongan", "gan"])
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--learning_rate", type=float, default=0.0002)
    p.add_argument("--beta1", type=float, default=0.5)
    p.add_argument("--beta2", type=float, default=0.999)
    p.add_argument("--image_size", type=int, default=64)
    p.add_argument("--gan_curriculum", type=int, default=10000)
    p.add_argument("--starting_rate", type=float, default=0.01)
    p.add_argument("--default_rate", type=float, default=0.5)
    p.add_argument("--style_A", type=str, default=None)
    p.add_argument("--style_B", type=str, default=None)
    p.add_argument("--constraint", type=str, default=None)
    p.add_argument("--constraint_type", type=str, default=None)
    p.add_argument("--n_test", type=int, default=200)
    p.add_argument("--update_interval", type=int, default=3)
    p.add_argument("--log_interval", type=int, default=50)
    p.add_argument("--image_save_interval", type=int, default=1000)
    p.add_argument("--model_save_interval", type=int, default=10000)
    # additions of this implementation
    p.add_argument("--data_A", type=str, default=None, help="torch.save'd float tensor [n,3,S,S] for domain A")
    p.add_argument("--data_B", type=str, default=None, help="torch.save'd float tensor [n,3,S,S] for domain B")
    p.add_argument("--synthetic_size", type=int, default=1024, help="images per domain when no data files are given")
    p.add_argument("--data_source", type=str, default="auto", choices=["auto", "files", "shards", "tensors", "synthetic"])
    p.add_argument("--data_root", type=str, default=None, help="root of the reference's dataset layout (dataset.py:14-22; default ./datasets)")
    p.add_argument("--shard_A", type=str, nargs="*", default=None, help="pre-decoded uint8 [n,H,W,3] .npy shard(s) for domain A")
    p.add_argument("--shard_B", type=str, nargs="*", default=None, help="pre-decoded uint8 [n,H,W,3] .npy shard(s) for domain B")
    p.add_argument("--loader_workers", type=int, default=4, help="decode threads of the device loader (reference: num_workers=4)")
    p.add_argument("--max_iters", type=int, default=0, help="stop after this many iterations (0 = all epochs)")
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--no_graph", action="store_true", help="dispatch every kernel from Python (no hipGraph replay)")
    p.add_argument("--weight_decay", type=float, default=0.00001)
    p.add_argument("--save_train_state", action="store_true",
                   help="also write train_state_{iters}.pth (weights + Adam moments + iteration) at every model save")
    p.add_argument("--resume", type=str, default=None,
                   help="train_state_*.pth to continue from: weights, BN buffers, Adam moments, iteration and position in the data order")
    p.add_argument("--skip_log_only_passes", action="store_true",
                   help="on iterations that print no log line, D-steps skip the two reconstruction passes (they feed only the "
                        "log). Same weights; the generators' BatchNorm running statistics then differ from the reference's")
    p.add_argument("--comm", type=str, default="auto", choices=["auto", "capi", "c10d"],
                   help="data-parallel transport: the library's own RCCL communicator (capi) or torch.distributed (c10d)")
    p.add_argument("--mfma_dtype", type=str, default="f32", choices=["f32", "bf16", "f32x3"],
                   help="bf16: conv operands rounded to bf16 on the matrix cores, fp32 accumulate/BatchNorm/weights/Adam; "
                        "f32x3: fp32-accurate products from three bf16 planes per operand (six bf16 MFMAs per block)")
    p.add_argument("--act_dtype", type=str, default="f32", choices=["f32", "bf16"],
                   help="with --mfma_dtype bf16: feature maps and their gradients stored in bf16 (fp32 BatchNorm statistics and arithmetic, "
                        "fp32 weights / parameter gradients / Adam)")
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def load_domains(args, device, rank=0, world_size=1):
    """Both domains resident in HBM.  Synthetic data: per-rank seed in a single process (benchmark semantics), ONE
    shared dataset under data parallelism (every rank then takes its DistributedSampler shard of it)."""
    if args.data_A and args.data_B:
        A = torch.load(args.data_A, map_location="cpu")
        B = torch.load(args.data_B, map_location="cpu")
        A = A if A.dtype == torch.uint8 else A.float()
        B = B if B.dtype == torch.uint8 else B.float()
    else:
        g = torch.Generator().manual_seed(1000 + (rank if world_size == 1 else 0))
        A = torch.rand(args.synthetic_size, 3, args.image_size, args.image_size, generator=g)
        B = torch.rand(args.synthetic_size, 3, args.image_size, args.image_size, generator=g)
    return A.to(device), B.to(device)


class _ResidentData:
    """Both domains resident in HBM (tensor files / synthetic): a batch is an index_select (+ the uint8 ingest kernel)."""

    def __init__(self, A, B):
        self.A, self.B = A, B
        self.size = min(len(A), len(B))

    def epoch(self, batches, start=0):
        for ia, ib in batches[start:]:
            yield take_batch(self.A, ia), take_batch(self.B, ib)


class _StreamedData:
    """Image files or pre-decoded shards: dataset.DeviceLoader stages the next batch while the current one trains."""

    def __init__(self, src_A, src_B, domains, image_size, device, workers):
        self.src, self.domains, self.S, self.device, self.workers = (src_A, src_B), domains, image_size, device, workers
        self.size = min(len(src_A), len(src_B))

    def epoch(self, batches, start=0):
        from . import dataset as ds
        loader = ds.DeviceLoader(self.src[0], self.src[1], self.domains, self.S,
                                 [(a.cpu().numpy(), b.cpu().numpy()) for a, b in batches[start:]], device=self.device, workers=self.workers)
        try:
            yield from loader
        finally:
            loader.close()


def open_data(args, device, rank=0, world_size=1):
    """The training data behind one interface (``.size``, ``.epoch(index batches, start)``), chosen by --data_source."""
    from . import dataset as ds
    src = getattr(args, "data_source", "auto")
    ds.set_root(getattr(args, "data_root", None) or "./datasets")      # every run states its root: nothing leaks from an earlier call
    if src == "auto":
        if args.data_A and args.data_B:
            src = "tensors"
        elif getattr(args, "shard_A", None) and getattr(args, "shard_B", None):
            src = "shards"
        else:
            # the synthetic stand-in is for the bare command line on a machine without datasets ONLY: a run that names its data
            # (--data_root, --style_A / --style_B / --constraint) or whose dataset root exists raises what the reference raises
            # (a mistyped style: KeyError, an empty directory: ValueError, a wrong root: FileNotFoundError) instead of training on noise
            named = any(getattr(args, k, None) for k in ("data_root", "style_A", "style_B", "constraint"))
            if named or os.path.isdir(getattr(args, "data_root", None) or "./datasets"):
                ds.get_data(args)
                src = "files"
            else:
                src = "synthetic"
    domains = ds.task_domains(args.task_name)
    workers = getattr(args, "loader_workers", 4)
    if src == "files":
        data_A, data_B, _, _ = ds.get_data(args)
        return _StreamedData(ds.FileSource(data_A), ds.FileSource(data_B), domains, args.image_size, device, workers), src
    if src == "shards":
        return _StreamedData(ds.ShardSource(args.shard_A), ds.ShardSource(args.shard_B), domains, args.image_size, device, workers), src
    if src == "tensors" and not (args.data_A and args.data_B):
        raise ValueError("--data_source tensors needs --data_A and --data_B")
    if src == "synthetic":
        args = argparse.Namespace(**{**vars(args), "data_A": None, "data_B": None})
    return _ResidentData(*load_domains(args, device, rank, world_size)), src


def take_batch(data, idx):
    """Rows ``idx`` of a resident domain as the float NCHW batch the networks take (image_translation.py:332-333).
    uint8 [n,S,S,3] sources are normalised and transposed on the device (dataset.py:65-66)."""
    x = data.index_select(0, idx)
    if x.dtype == torch.uint8:
        from . import ops
        return ops.u8hwc_to_f32chw(x)
    return x


def run_dirs(args, rank_suffix=""):
    ts = datetime.now().strftime("%Y%m%d_%H%M%S") + rank_suffix
    sub = Path(args.task_name)
    if args.style_A:
        sub = sub / args.style_A
    sub = sub / args.model_arch / ts
    return Path(args.results_dir) / sub, Path(args.models_dir) / sub


def save_models(trainer, model_path, tag, next_iter=None, with_state=False, loader=None):
    trainer.finish()                      # join the communication stream before reading parameters
    if with_state and next_iter is not None:
        torch.save(trainer.train_state(next_iter, extra=loader), model_path / f"train_state_{tag}.pth")
    names = dict(gen_A=trainer.generator_A, gen_B=trainer.generator_B,
                 dis_A=trainer.discriminator_A, dis_B=trainer.discriminator_B)
    for k, net in names.items():
        sd = {n: (t.detach().contiguous().cpu()) for n, t in net.state_dict().items()}
        torch.save(sd, model_path / f"{k}_{tag}.pth")


def epoch_batches(args, epoch, data_size, rank, world_size, gperm, device):
    """Index tensors (idx_A, idx_B) of every batch of this epoch, in order (see the module docstring)."""
    bs = args.batch_size
    if world_size > 1:
        from .dp import distributed_indices
        idx = torch.tensor(distributed_indices(data_size, world_size, rank, epoch, seed=0), device=device)
        out = [(idx[i:i + bs], idx[i:i + bs]) for i in range(0, len(idx), bs)]
        return [b for b in out if len(b[0]) >= 2]
    perm_A = torch.randperm(data_size, generator=gperm).to(device)     # shuffle_data, dataset.py:24-35
    perm_B = torch.randperm(data_size, generator=gperm).to(device)
    return [(perm_A[i * bs:(i + 1) * bs], perm_B[i * bs:(i + 1) * bs]) for i in range(data_size // bs)]


def batches_per_epoch(args, data_size, world_size):
    if world_size > 1:
        shard = -(-data_size // world_size)
        n = -(-shard // args.batch_size)
        return n - 1 if shard % args.batch_size == 1 else n
    return data_size // args.batch_size


def train(args, trainer=None, rank=0, world_size=1, is_main=True, process_group=None):
    if args.task_name not in TASKS:
        raise ValueError(f"unknown task_name {args.task_name}; choose from {TASKS}")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: this implementation has no CPU path (use the reference for CPU runs)")
    device = torch.device("cuda", torch.cuda.current_device())
    result_path, model_path = run_dirs(args)
    if is_main:
        result_path.mkdir(parents=True, exist_ok=True)
        model_path.mkdir(parents=True, exist_ok=True)
    if trainer is None:
        trainer = DiscoGANTrainer(args, device=device, image_size=args.image_size, seed=args.seed,
                                  process_group=process_group, use_graph=not args.no_graph,
                                  mfma_dtype=getattr(args, "mfma_dtype", "f32"), act_dtype=getattr(args, "act_dtype", "f32"),
                                  comm=getattr(args, "comm", "auto"))
    data, data_kind = open_data(args, device, rank, world_size)
    data_size = data.size
    if is_main:
        print(f"data source: {data_kind} ({data_size} images per domain)", flush=True)
    n_batches = batches_per_epoch(args, data_size, world_size)
    if n_batches < 1:
        raise ValueError(f"batch_size {args.batch_size} leaves no full batch in {data_size} images on {world_size} rank(s)")
    total_iterations = args.epochs * n_batches
    log_file = result_path / "training_log.txt"
    if is_main:
        with open(log_file, "w") as f:
            f.write(f"Task: {args.task_name}, Model: {args.model_arch}\n")
            f.write(f"Batch size: {args.batch_size}, Learning rate: {args.learning_rate}\n\n")
    iters = 0
    start_epoch = start_batch = 0
    if getattr(args, "resume", None):
        st = torch.load(args.resume, map_location="cpu")
        iters = trainer.load_train_state(st)
        # position in the data order: the state's loader record, else derived from the iteration count
        pos = st.get("loader") or dict(epoch=iters // n_batches, batch=iters % n_batches)
        start_epoch, start_batch = int(pos["epoch"]), int(pos["batch"])
        if start_batch >= n_batches:
            start_epoch, start_batch = start_epoch + 1, 0
        if is_main:
            print(f"resumed from {args.resume}: next iteration {iters} (epoch {start_epoch}, batch {start_batch})")
    start_iters = iters
    t0 = time.time()
    lazy = bool(getattr(args, "skip_log_only_passes", False))
    gperm = torch.Generator(device="cpu").manual_seed(args.seed + 17 * rank)
    if world_size == 1:
        for _ in range(start_epoch):                      # replay the generator to the resumed epoch (2 draws per epoch)
            torch.randperm(data_size, generator=gperm)
            torch.randperm(data_size, generator=gperm)
    done = False
    for epoch in range(start_epoch, args.epochs):
        batches = epoch_batches(args, epoch, data_size, rank, world_size, gperm, device)
        first = start_batch if epoch == start_epoch else 0
        for i, (A, B) in enumerate(data.epoch(batches, first), start=first):
            logging = iters % args.log_interval == 0
            out = trainer.train_iteration(A, B, iters, need_losses=logging or not lazy)
            if is_main and logging:
                msg = trainer.format_log(iters, total_iterations, out)
                dt = time.time() - t0
                print(msg + f"  [{(iters - start_iters + 1) * args.batch_size * world_size / max(dt, 1e-9):.1f} img/s]", flush=True)
                with open(log_file, "a") as f:
                    f.write(msg + "\n")
            if is_main and iters % args.model_save_interval == 0:
                save_models(trainer, model_path, str(iters), iters + 1, getattr(args, "save_train_state", False),
                            loader=dict(epoch=epoch, batch=i + 1))
            iters += 1
            if args.max_iters and iters >= args.max_iters:
                done = True
                break
        if done:
            break
    if is_main:
        save_models(trainer, model_path, "final", iters, getattr(args, "save_train_state", False),
                    loader=dict(epoch=iters // n_batches, batch=iters % n_batches))
        print(f"Training completed. Final models saved to {model_path}")
        print(f"Results and logs saved to {result_path}")
    train.last_paths = (result_path, model_path)
    return trainer


def main(argv=None):
    args = parse_args(argv)
    return train(args)


if __name__ == "__main__":
    main()
