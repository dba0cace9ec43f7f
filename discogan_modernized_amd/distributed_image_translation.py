"""Data-parallel training CLI (reference distributed_image_translation.py:26-126, 326-636).

One process per GPU, launched by ``python -m torch.distributed.run --nproc-per-node W`` (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; the reference hard-codes
``localhost:12355`` and a ``--world_size`` default of 4, :31-35,57-58).  Backend ``nccl`` == RCCL over
xGMI.  Every rank builds identical replicas from ``torch.manual_seed(1234)`` (:372), draws its own
batch, keeps BatchNorm / feature-matching statistics rank-local, and the stepped side's flat gradient
buffer is all-reduced (sum, then divided by W inside the Adam kernel).  Rank 0 logs and saves.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import image_translation as it


def parse_args(argv=None):
    p = it.build_parser("HIP/MI355X implementation of DiscoGAN for distributed (data-parallel) training")
    p.add_argument("--distributed", action="store_true")
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--world_size", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    p.add_argument("--load_gen_A", type=str, default=None)
    p.add_argument("--load_gen_B", type=str, default=None)
    p.add_argument("--load_dis_A", type=str, default=None)
    p.add_argument("--load_dis_B", type=str, default=None)
    return p.parse_args(argv)


def setup(local_rank, world_size, backend="nccl"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12355")
    rank = int(os.environ.get("RANK", local_rank))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world_size)
    return rank


def cleanup():
    if dist.is_initialized():
        dist.destroy_process_group()


def load_checkpoints(args, trainer):
    """--load_gen_A/B --load_dis_A/B (:117-124, 379-393): weights only, like the reference."""
    for flag, net in (("load_gen_A", trainer.generator_A), ("load_gen_B", trainer.generator_B),
                      ("load_dis_A", trainer.discriminator_A), ("load_dis_B", trainer.discriminator_B)):
        path = getattr(args, flag)
        if path:
            net.load_state_dict(torch.load(path, map_location="cpu"))


def main_worker(local_rank, args):
    pg = None
    rank, world = 0, 1
    if args.distributed:
        rank = setup(local_rank, args.world_size)
        world = dist.get_world_size()
        pg = dist.group.WORLD
    else:
        torch.cuda.set_device(0)
    try:
        from .trainer import DiscoGANTrainer
        device = torch.device("cuda", torch.cuda.current_device())
        trainer = DiscoGANTrainer(args, device=device, image_size=args.image_size, seed=args.seed,
                                  process_group=pg, use_graph=not args.no_graph,
                                  mfma_dtype=getattr(args, "mfma_dtype", "f32"), act_dtype=getattr(args, "act_dtype", "f32"),
                                  comm=getattr(args, "comm", "auto"))
        load_checkpoints(args, trainer)
        # barriers go through the transport the gradients use (capi: the library's communicator -- the process then holds ONE
        # RCCL instance; torch.distributed's is created lazily by its first device collective and never is)
        sync = (trainer.xg.barrier if trainer.xg is not None else (lambda: None))
        sync()
        if rank == 0 and trainer.xg is not None:
            print("data-parallel exchange:", trainer.xg.describe(), flush=True)
        try:
            it.train(args, trainer=trainer, rank=rank, world_size=world, is_main=(rank == 0), process_group=pg)
            trainer.finish()
            sync()
        finally:
            trainer.close()                       # dg_dp_destroy before the process group goes away
    finally:
        if args.distributed:
            cleanup()


def main(argv=None):
    args = parse_args(argv)
    if "LOCAL_RANK" in os.environ:
        args.local_rank = int(os.environ["LOCAL_RANK"])
        args.distributed = True
        args.world_size = int(os.environ.get("WORLD_SIZE", args.world_size))
    main_worker(args.local_rank if args.distributed else 0, args)


if __name__ == "__main__":
    main()
