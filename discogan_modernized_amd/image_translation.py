"""Re-hosted training CLI (reference image_translation.py:21-81, 211-435).

Same flags and defaults (``--task_name --model_arch --image_size --batch_size --epochs
--learning_rate --beta1 --beta2 --gan_curriculum --starting_rate --default_rate --update_interval
--log_interval --model_save_interval ...``), same log-line format (:394-398) and checkpoint file
names (``gen_A_{iters}.pth`` ... ``*_final.pth``, :420-432).  The training loop dispatches into the
HIP kernels through ``DiscoGANTrainer``.

Image-file datasets (dataset.py) are outside the hot path: batches come either from ``--data_A/--data_B``
tensor files (``torch.save``d float tensors [n,3,S,S] in [0,1]) or, by default, from synthetic uniform
tensors (``--synthetic_size`` images per domain), which is what the benchmark metric is defined on.

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
    p.add_argument("--max_iters", type=int, default=0, help="stop after this many iterations (0 = all epochs)")
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--no_graph", action="store_true", help="dispatch every kernel from Python (no hipGraph replay)")
    p.add_argument("--weight_decay", type=float, default=0.00001)
    p.add_argument("--save_train_state", action="store_true",
                   help="also write train_state_{iters}.pth (weights + Adam moments + iteration) at every model save")
    p.add_argument("--resume", type=str, default=None, help="train_state_*.pth to continue from (exact resume)")
    p.add_argument("--mfma_dtype", type=str, default="f32", choices=["f32", "bf16"],
                   help="bf16: conv operands rounded to bf16 on the matrix cores, fp32 accumulate/BatchNorm/weights/Adam")
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def load_domains(args, device, rank=0):
    if args.data_A and args.data_B:
        A = torch.load(args.data_A, map_location="cpu").float()
        B = torch.load(args.data_B, map_location="cpu").float()
    else:
        g = torch.Generator().manual_seed(1000 + rank)
        A = torch.rand(args.synthetic_size, 3, args.image_size, args.image_size, generator=g)
        B = torch.rand(args.synthetic_size, 3, args.image_size, args.image_size, generator=g)
    return A.to(device), B.to(device)


def run_dirs(args, rank_suffix=""):
    ts = datetime.now().strftime("%Y%m%d_%H%M%S") + rank_suffix
    sub = Path(args.task_name)
    if args.style_A:
        sub = sub / args.style_A
    sub = sub / args.model_arch / ts
    return Path(args.results_dir) / sub, Path(args.models_dir) / sub


def save_models(trainer, model_path, tag, iters=None, with_state=False):
    trainer.finish()                      # join the communication stream before reading parameters
    if with_state and iters is not None:
        torch.save(trainer.train_state(iters), model_path / f"train_state_{tag}.pth")
    names = dict(gen_A=trainer.generator_A, gen_B=trainer.generator_B,
                 dis_A=trainer.discriminator_A, dis_B=trainer.discriminator_B)
    for k, net in names.items():
        sd = {n: (t.detach().contiguous().cpu()) for n, t in net.state_dict().items()}
        torch.save(sd, model_path / f"{k}_{tag}.pth")


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
                                  mfma_dtype=getattr(args, "mfma_dtype", "f32"))
    data_A, data_B = load_domains(args, device, rank)
    data_size = min(len(data_A), len(data_B))
    n_batches = data_size // args.batch_size
    total_iterations = args.epochs * n_batches
    log_file = result_path / "training_log.txt"
    if is_main:
        with open(log_file, "w") as f:
            f.write(f"Task: {args.task_name}, Model: {args.model_arch}\n")
            f.write(f"Batch size: {args.batch_size}, Learning rate: {args.learning_rate}\n\n")
    iters = 0
    if getattr(args, "resume", None):
        iters = trainer.load_train_state(torch.load(args.resume, map_location="cpu"))
        if is_main:
            print(f"resumed from {args.resume} at iteration {iters}")
    start_iters = iters
    t0 = time.time()
    gperm = torch.Generator(device="cpu").manual_seed(args.seed + 17 * rank)
    for epoch in range(args.epochs):
        perm_A = torch.randperm(data_size, generator=gperm).to(device)     # shuffle_data, dataset.py:24-35
        perm_B = torch.randperm(data_size, generator=gperm).to(device)
        for i in range(n_batches):
            sl = slice(i * args.batch_size, (i + 1) * args.batch_size)
            A = data_A.index_select(0, perm_A[sl])
            B = data_B.index_select(0, perm_B[sl])
            # loss values are only read on log iterations; elsewhere a D-step may skip its log-only passes
            out = trainer.train_iteration(A, B, iters, need_losses=(iters % args.log_interval == 0))
            if is_main and iters % args.log_interval == 0:
                msg = trainer.format_log(iters, total_iterations, out)
                dt = time.time() - t0
                print(msg + f"  [{(iters - start_iters + 1) * args.batch_size * world_size / max(dt, 1e-9):.1f} img/s]", flush=True)
                with open(log_file, "a") as f:
                    f.write(msg + "\n")
            if is_main and iters % args.model_save_interval == 0:
                save_models(trainer, model_path, str(iters), iters, getattr(args, "save_train_state", False))
            iters += 1
            if args.max_iters and iters >= args.max_iters:
                break
        if args.max_iters and iters >= args.max_iters:
            break
    if is_main:
        save_models(trainer, model_path, "final", iters, getattr(args, "save_train_state", False))
        print(f"Training completed. Final models saved to {model_path}")
        print(f"Results and logs saved to {result_path}")
    return trainer


def main(argv=None):
    args = parse_args(argv)
    train(args)


if __name__ == "__main__":
    main()
