#!/usr/bin/env python3
"""Headline benchmark: images/sec of the DiscoGAN training step on MI355X.

    python bench.py --gpus 1 --steps 30 --warmup 9
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): edges2shoes / discogan / image_size=64 / batch_size=256 PER GPU,
fp32, synthetic uniform [0,1) batches resident in HBM, models from torch.manual_seed(1234).  A "step"
is one training iteration (4 G passes + 4 D passes forward, backward + Adam of the side selected by
``iters % 3``; image_translation.py:336-390).  K steps walk the D,G,G cycle; images = batch x ranks
per step (one (A,B) index = one image, dataset.py:210-213).  Weak scaling: per-GPU batch is fixed.

One JSON line on rank 0 with the contract fields plus
  roofline     : the dominant kernel family (igemm_kernel, fp32 MFMA): algorithmic conv FLOPs of every
                 launch in one D,G,G cycle / summed HIP-event durations of those launches
  cpu_baseline : oracle/discogan_ref.py (PyTorch-CPU restatement of the reference step, pinned to the
                 reference's golden vectors) timed on the host cores for a bounded sample
  extra        : side measurements, never the headline: the HBM-bound kernel families (GB/s), whole-step TFLOP/s,
                 the D-step / G-step split, the rate on unlogged iterations, the 512 px / batch 32 configuration
                 (BASELINE configs[3]) with its own roofline leg, and both sizes with bf16 MFMA operands
                 (BASELINE configs[4] arithmetic)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (never the 2:1-sparsity figure); only used with --mfma_dtype bf16
PMC_TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                "r01_pmc_traffic_per_launch_64px_bs256.json")


def pmc_traffic_bytes(image_size, batch):
    """HBM-side bytes per launch of the dominant instantiation (igemm_kernel<0,2,2,32,true,0>) from the committed
    rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see profiles/); only valid for the profiled workload."""
    if (image_size, batch) != (64, 256) or not os.path.exists(PMC_TRAFFIC_FILE):
        return None
    try:
        k = json.load(open(PMC_TRAFFIC_FILE))["kernels"]["void igemm_kernel<0, 2, 2, 32, true, 0>(IgemmArgs)"]
        return int(k["hbm_MB_per_launch"] * 1024 * 1024)
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=9)
    ap.add_argument("--image_size", type=int, default=64)
    ap.add_argument("--batch_size", type=int, default=256, help="per GPU")
    ap.add_argument("--no_graph", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_512", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--single_stream", action="store_true", help="do not overlap the A/B chains on two HIP streams")
    ap.add_argument("--cu_partition", default=os.environ.get("DG_CU_PARTITION", ""), choices=["", "xcd", "half"],
                    help="run the two chains on CU-masked streams (disjoint halves of the chip)")
    ap.add_argument("--mfma_dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16 = BASELINE configs[4] arithmetic for the MAIN run (the JSON then says dtype bf16); default f32")
    ap.add_argument("--skew", type=int, default=int(os.environ.get("DG_SKEW", "0")), help="hold the B chain back by this many steps of the A chain")
    ap.add_argument("--turns", action="store_true", help="make the two chains take turns on the matrix cores (measured slower)")
    ap.add_argument("--async_wgrad", action="store_true",
                    help="weight-gradient kernels on a third stream (measured: neutral to slightly slower)")
    return ap.parse_args()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def timed_run(trainer, A, B, steps, warmup, world, start_iter=0, need_losses=True):
    it = start_iter
    for _ in range(warmup):
        trainer.train_iteration(A, B, it, need_losses=need_losses)
        it += 1
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_iteration(A, B, it, need_losses=need_losses)
        it += 1
    trainer.finish()
    barrier_sync(world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # max over ranks
        dt = float(t.item())
    return dt, it


def roofline_pass(trainer, A, B, start_iter):
    """One instrumented D,G,G cycle (eager dispatch): HIP events around every igemm launch."""
    from discogan_modernized_amd import ops
    ui = trainer.args.update_interval
    start_iter = (start_iter + ui - 1) // ui * ui            # align to a D-step
    was_graph, was_two, was_aw = trainer.use_graph, trainer.two_streams, trainer.wgrad_stream
    trainer.use_graph = False
    trainer.two_streams = False          # isolated kernel durations: one stream, one kernel at a time
    trainer.wgrad_stream = None
    ops.PROFILE = []
    ops.PROFILE_HBM = []
    for k in range(ui):
        trainer.train_iteration(A, B, start_iter + k)
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    hrec, ops.PROFILE_HBM = ops.PROFILE_HBM, None
    hbm = {}
    for name, nbytes, e0, e1 in hrec:
        d = hbm.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += nbytes
        d[2] += e0.elapsed_time(e1)
    roofline_pass.hbm = {k: dict(launches=v[0], algorithmic_GB=round(v[1] / 1e9, 3), ms=round(v[2], 3),
                                 GBps=round(v[1] / max(v[2], 1e-9) / 1e6, 1),
                                 frac_of_8TBps=round(v[1] / max(v[2], 1e-9) / 1e6 / 8000.0, 3)) for k, v in hbm.items()}
    rec = [r for r in rec if r[0] != "head1"]   # K==1 head uses plain reduction kernels, not the MFMA family
    trainer.use_graph, trainer.two_streams, trainer.wgrad_stream = was_graph, was_two, was_aw
    fam = [r for r in rec if r[0] != "c3_fwd"]   # the 3-channel forward is its own streaming kernel (edge.hip)
    flops = sum(r[1] for r in fam)
    ms = sum(r[2].elapsed_time(r[3]) for r in fam)
    by = {}
    for name, f, e0, e1 in rec:
        d = by.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += f
        d[2] += e0.elapsed_time(e1)
    return flops, ms, len(fam), by


def step_split_ms(trainer, A, B, start_iter, cycles=4):
    """D-step and G-step time separately: HIP events around single iterations, default dispatch mode."""
    ui = trainer.args.update_interval
    it = (start_iter + 2 * ui) // ui * ui
    for k in range(ui):
        trainer.train_iteration(A, B, it + k)
    it += ui
    acc = {"D": [], "G": []}
    for k in range(cycles * ui):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        trainer.train_iteration(A, B, it + k)
        e1.record()
        acc["D" if trainer.is_dis_step(it + k) else "G"].append((e0, e1))
    trainer.finish()
    torch.cuda.synchronize()
    return {f"ms_{k}_step": round(sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1), 3) for k, v in acc.items()}


def cpu_baseline(image_size, batch, update_interval=3):
    """Oracle step on the host cores: 1 warm-up iteration + one D,G,G cycle."""
    from oracle import discogan_ref as O
    st = O.build_state(image_size=image_size, seed=1234)
    A, B = O.synthetic_batch(batch, image_size, seed=1000)
    O.train_iteration(st, A, B, 1)                      # warm-up (G-step; allocs, oneDNN primitives)
    t0 = time.perf_counter()
    it, n = update_interval, 0
    while True:                                         # whole D,G,G cycles until ~10 s of CPU work
        for _ in range(update_interval):
            O.train_iteration(st, A, B, it)
            it += 1
            n += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or n >= 60:
            break
    return batch * n / dt, dt, n


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback for the product path)")
    if os.environ.get("DG_DIST_BACKEND", "nccl") != "nccl":
        local = local % max(torch.cuda.device_count(), 1)          # rehearsal: ranks share the visible GPU(s)
    torch.cuda.set_device(local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" == RCCL over xGMI; DG_DIST_BACKEND=gloo lets the launch be rehearsed with several ranks on ONE GPU
        backend = os.environ.get("DG_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD
    from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args, synthetic_batch
    dev = torch.device("cuda", local)

    trainer = DiscoGANTrainer(default_args(), device=dev, image_size=a.image_size, seed=1234, process_group=pg,
                              use_graph=not a.no_graph, two_streams=not a.single_stream,
                              async_wgrad=a.async_wgrad and not a.single_stream,
                              cu_partition=a.cu_partition or None, mfma_turns=a.turns, skew_steps=a.skew,
                              mfma_dtype=a.mfma_dtype)
    A, B = synthetic_batch(a.batch_size, a.image_size, 1000 + rank, dev)
    log(f"models built; running {a.warmup} warm-up + {a.steps} timed steps @{a.image_size}px batch {a.batch_size} x {world} GPU")
    dt, it = timed_run(trainer, A, B, a.steps, a.warmup, world)
    used_graph, used_overlap = bool(trainer.use_graph), bool(trainer.overlap_comm)
    log(f"timed region done: {dt / a.steps * 1e3:.3f} ms/step")
    images = a.batch_size * world * a.steps
    value = images / dt

    roof = None
    if not a.no_roofline:
        flops, ms, nlaunch, by = roofline_pass(trainer, A, B, it)
        log(f"roofline pass done: {flops / ms / 1e9:.1f} TFLOP/s over {nlaunch} igemm launches")
        ach = flops / (ms * 1e-3) / 1e12
        peak = MFMA_F32_PEAK_TFLOPS if a.mfma_dtype == "f32" else MFMA_BF16_PEAK_TFLOPS
        kname = ("igemm_kernel<*> (v_mfma_f32_32x32x2_f32 implicit-GEMM conv family)" if a.mfma_dtype == "f32" else
                 "igemm_kernel<*,PREC=1> (v_mfma_f32_32x32x16_bf16, fp32 tensors: operand movement bound, see DESIGN.md 3.1)")
        roof = dict(bound="mfma", kernel=kname,
                    achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                    frac=round(ach / peak, 4),
                    traffic=pmc_traffic_bytes(a.image_size, a.batch_size) if a.mfma_dtype == "f32" else None,
                    traffic_note="bytes/launch beyond L2 for igemm_kernel<0,2,2,32,true,0> from profiles/r01_pmc_traffic_per_launch_64px_bs256.json (PMC, offline)",
                    launches_per_cycle=nlaunch, algorithmic_gflop_per_cycle=round(flops / 1e9, 2),
                    avg_launch_us=round(ms * 1e3 / max(nlaunch, 1), 2),
                    by_op={k: dict(launches=v[0], gflop=round(v[1] / 1e9, 2), ms=round(v[2], 3),
                                   tflops=round(v[1] / max(v[2], 1e-9) / 1e9, 1)) for k, v in by.items()})
    extra = {}
    if roof is not None:
        extra["hbm_bound_families"] = getattr(roofline_pass, "hbm", {})
        extra["note_hbm"] = ("HBM-bound kernel families (SURVEY 8(d)): algorithmic bytes / HIP-event time per op in the same "
                             "instrumented single-stream cycle; an op may be several kernels (bn_backward = partial + finalize + apply)")
    # whole-step arithmetic rate from the live algorithmic FLOPs of a D,G,G cycle (5.568 GFLOP/img at 64 px,
    # 640.8 at 512 px), and the D-step / G-step split
    live = {64: 5.568e9, 512: 640.8e9}.get(a.image_size)
    if live is not None:
        extra["whole_step_tflops"] = round(value / world * live / 1e12, 2)
    extra.update(step_split_ms(trainer, A, B, it))
    if not a.no_512 and world == 1:
        trb = DiscoGANTrainer(default_args(), device=dev, image_size=a.image_size, seed=1234, process_group=pg,
                              use_graph=not a.no_graph, two_streams=not a.single_stream, mfma_dtype="bf16")
        dtb, _ = timed_run(trb, A, B, a.steps, a.warmup, world)
        extra["images_per_sec_bf16_mfma"] = round(a.batch_size * world * a.steps / dtb, 2)
        extra["note_bf16"] = ("same workload with mfma_dtype=bf16 (BASELINE configs[4] arithmetic: conv operands rounded to bf16, "
                              "v_mfma_f32_32x32x16_bf16, fp32 accumulate / BatchNorm / master weights / Adam). NOT the headline value.")
        del trb
        torch.cuda.empty_cache()
    if not a.no_512:     # (same switch as the other extra line)
        ui = trainer.args.update_interval
        it2 = (it + 2 * ui) // ui * ui
        dtl, _ = timed_run(trainer, A, B, a.steps, ui, world, start_iter=it2, need_losses=False)
        extra["images_per_sec_unlogged_iterations"] = round(a.batch_size * world * a.steps / dtl, 2)
        extra["note_unlogged"] = ("same workload when the iteration's loss values are not read (every iteration that prints "
                                  "no log line: log_interval 50 in the reference): D-steps skip the two reconstruction "
                                  "passes that feed only the log; weights identical. NOT the headline value.")
    del trainer
    torch.cuda.empty_cache()
    if not a.no_512:
        tr512 = DiscoGANTrainer(default_args(), device=dev, image_size=512, seed=1234, process_group=pg,
                                use_graph=not a.no_graph, two_streams=not a.single_stream,
                                async_wgrad=a.async_wgrad and not a.single_stream,
                              cu_partition=a.cu_partition or None, mfma_turns=a.turns, skew_steps=a.skew)
        A5, B5 = synthetic_batch(32, 512, 1000 + rank, dev)
        log("512px models built")
        dt5, it5 = timed_run(tr512, A5, B5, 6, 6, world)
        log(f"512px done: {dt5 / 6 * 1e3:.1f} ms/step")
        if not a.no_roofline:
            f5, ms5, n5, _ = roofline_pass(tr512, A5, B5, it5)
            extra["roofline_512px_bs32"] = dict(achieved=round(f5 / ms5 / 1e9, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                                                frac=round(f5 / ms5 / 1e9 / MFMA_F32_PEAK_TFLOPS, 4), launches_per_cycle=n5,
                                                whole_step_tflops=round(32 * world * 6 / dt5 * 640.8e9 / world / 1e12, 2))
        if world == 1:
            del tr512
            torch.cuda.empty_cache()
            tr512 = DiscoGANTrainer(default_args(), device=dev, image_size=512, seed=1234, process_group=pg,
                                    use_graph=not a.no_graph, two_streams=not a.single_stream, mfma_dtype="bf16")
            dtb5, _ = timed_run(tr512, A5, B5, 6, 6, world)
            extra["images_per_sec_512px_bs32_bf16_mfma"] = round(32 * world * 6 / dtb5, 2)
        extra.update(images_per_sec_512px_bs32=round(32 * world * 6 / dt5, 2), ms_per_step_512px_bs32=round(dt5 / 6 * 1e3, 2),
                     note="BASELINE configs[3]: tops2hanbok image_size=512 batch_size=32 per GPU, fp32, 6 timed steps")
        del tr512, A5, B5
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        torch.set_num_threads(min(16, os.cpu_count() or 1))   # the 1-GPU box's CPU share is 16 cores
        log("cpu baseline (oracle) ...")
        v, cdt, cn = cpu_baseline(a.image_size, 64)
        cpu = dict(value=round(v, 3), unit="images/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"oracle/discogan_ref.py, image_size={a.image_size} batch 64 (BASELINE configs[0]), "
                          f"{cn} iterations (whole D,G,G cycles) after 1 warm-up iteration, {cdt:.1f} s")
    if rank == 0:
        line = dict(metric="images/sec per DiscoGAN train step", value=round(value, 2), unit="images/s",
                    n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype=a.mfma_dtype, data="synthetic",
                    config=dict(workload=f"edges2shoes discogan image_size={a.image_size} batch_size={a.batch_size} per GPU "
                                         f"(BASELINE configs[1]); D,G,G cycle, fwd+bwd+Adam, dead backward work skipped"
                                         + ("" if a.mfma_dtype == "f32" else "; conv operands rounded to bf16 (bf16 MFMA, fp32 accumulate)"),
                                global_batch=a.batch_size * world, parallelism=f"dp{world}",
                                hipgraph=used_graph, hip_streams=1 if a.single_stream else 2,
                                allreduce_overlap=used_overlap),
                    roofline=roof, cpu_baseline=cpu, extra=extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
