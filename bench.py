#!/usr/bin/env python3
"""Headline benchmark: images/sec of the DiscoGAN training step on MI355X.

    python bench.py                                  # 1 GPU, the metric's 512 px / batch 32 configuration
    python bench.py --gpus N --steps K --warmup W    # N > 1: starts its own N ranks (one per GPU) and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W       # the driver's form: ranks already exist (RANK/WORLD_SIZE in the env)

Workload of `value` (BASELINE.json metric "images/sec per DiscoGAN train step (64px bs64, 512px bs32)", configs[3]):
tops2hanbok / discogan / image_size=512 / batch_size=32 PER GPU, fp32, synthetic uniform [0,1) batches resident in
HBM, models from torch.manual_seed(1234).  A "step" is one training iteration (4 G passes + 4 D passes forward, backward
+ Adam of the side selected by ``iters % 3``; image_translation.py:336-390).  The K timed steps start on a D-step
(warm-up is rounded up to whole D,G,G cycles) and K defaults to 12 = four whole cycles; images = batch x ranks per step
(one (A,B) index = one image, dataset.py:210-213).  Weak scaling: per-GPU batch is fixed.
``--image_size 64`` selects the 64 px network (configs[1]: batch 256; configs[2]'s per-GPU shape: --batch_size 64).

One JSON line on rank 0 with the contract fields plus
  roofline     : the dominant kernel family (igemm_kernel, fp32 MFMA): algorithmic conv FLOPs of every launch in one
                 D,G,G cycle / summed HIP-event durations of those launches (events on the launch stream); traffic =
                 HBM-side bytes per launch of the dominant instantiation from the committed rocprofv3 PMC passes
  cpu_baseline : oracle/discogan_ref.py (PyTorch-CPU restatement of the reference step, pinned to the reference's
                 golden vectors) timed on the host cores for a bounded sample of the same 512 px workload
  comm         : (N > 1) transport, RCCL world size, all-reduce ms per D-step / G-step exchange, overlap mode
  extra        : side measurements, never the headline: D-step / G-step split, HBM-bound kernel families, the 64 px
                 configurations (batch 256 with its own roofline leg, batch 64 in both data-parallel dispatch modes),
                 bf16-MFMA runs (configs[4] arithmetic), the CPU baseline at 64 px / 64
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (never the 2:1-sparsity figure)
LIVE_GFLOP_PER_IMAGE = {64: 5.568, 512: 640.8}   # live conv FLOPs of a D,G,G cycle / 3, per image
WORKLOADS = {512: "tops2hanbok discogan image_size=512 batch_size={b} per GPU (BASELINE configs[3])",
             64: "edges2shoes discogan image_size=64 batch_size={b} per GPU (BASELINE configs[1]; batch 64 = configs[2]'s per-GPU shape)"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 12 at 512 px, 30 at 64 px)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps, rounded up to whole D,G,G cycles (default 6 / 9)")
    ap.add_argument("--image_size", type=int, default=512, choices=[64, 512])
    ap.add_argument("--batch_size", type=int, default=None, help="per GPU (default 32 at 512 px, 256 at 64 px)")
    ap.add_argument("--no_graph", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_baseline_batch", type=int, default=0,
                    help="also time ONE D,G,G cycle of the CPU oracle at this batch of the main image size (32 = the metric's own batch at 512 px: "
                         "~4 min on 16 threads, ~40 GB) -> extra.cpu_baseline_<S>px_bs<N>; the in-line cpu_baseline keeps its bounded batch-2 sample")
    ap.add_argument("--no_extra", action="store_true", help="headline + roofline only")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--single_stream", action="store_true", help="do not overlap the A/B chains on two HIP streams")
    ap.add_argument("--mfma_dtype", default="f32x3", choices=["f32", "bf16", "f32x3"],
                    help="arithmetic of the MAIN run (the JSON's dtype follows it). f32x3 (default): fp32-accurate products from three "
                         "bf16 planes per operand (24 significand bits, exact products, fp32 accumulation); f32: the exact fp32 MFMA "
                         "(reported under extra with its own roofline when it is not the main run); bf16: BASELINE configs[4] arithmetic")
    ap.add_argument("--act_dtype", default="f32", choices=["f32", "bf16"],
                    help="with --mfma_dtype bf16: feature maps and their gradients STORED in bf16 (fp32 BatchNorm statistics / arithmetic)")
    ap.add_argument("--no_x3_planes", action="store_true", help="with --mfma_dtype f32x3: split the operands inside every conv kernel "
                    "(register-staged tiles) instead of reading plane triples written once per tensor (A/B)")
    ap.add_argument("--group_launch", default="auto", choices=["auto", "on", "off"],
                    help="grouped launches: the A-side / B-side pass of each pair (and a discriminator's real + fake pass) as ONE launch per "
                         "kernel; auto = below 256 px on the exact-fp32 / register-staged f32x3 arithmetic")
    ap.add_argument("--group_plan", default="launch", choices=["launch", "single"],
                    help="split-K plan of a grouped conv launch: sized for the whole launch (default) or per problem (bitwise the ungrouped step)")
    ap.add_argument("--comm", default="auto", choices=["auto", "capi", "c10d"], help="data-parallel transport (dp.ExchangeGroup)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off", "graph"],
                    help="data-parallel exchange overlapped with compute (eager dispatch) or behind a replayed hipGraph")
    a = ap.parse_args(argv)
    if a.batch_size is None:
        a.batch_size = 32 if a.image_size == 512 else 256
    if a.steps is None:
        a.steps = 12 if a.image_size == 512 else 30
    if a.warmup is None:
        a.warmup = 6 if a.image_size == 512 else 9
    return a


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------------
# N > 1 typed as `python bench.py --gpus N`: the parent starts the ranks.  It must not have touched the GPU
# (a process that initialised HIP must never be replaced or forked into GPU work), so this runs before
# anything imports torch.cuda state: only `subprocess` is used here.
def self_launch(a, argv):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    return rc if rc != 0 or line else 1


def barrier_sync(world, xg=None):
    """Device barrier over the ranks.  On the capi transport it goes through the library's own communicator
    (dg_dp_barrier), so the timed run has ONE RCCL instance; torch.distributed's is never created."""
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        if xg is not None and xg.transport == "capi":
            xg.barrier()
        else:
            dist.barrier()
    torch.cuda.synchronize()


def timed_run(trainer, A, B, steps, warmup, world, start_iter=0, need_losses=True):
    import torch
    import torch.distributed as dist
    it = start_iter
    for _ in range(warmup):
        trainer.train_iteration(A, B, it, need_losses=need_losses)
        it += 1
    trainer.finish()
    barrier_sync(world, trainer.xg)
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_iteration(A, B, it, need_losses=need_losses)
        it += 1
    trainer.finish()
    barrier_sync(world, trainer.xg)
    dt = time.perf_counter() - t0
    if world > 1:
        from discogan_modernized_amd import dp
        dt = max(dp.host_allgather(dt))                            # max over ranks (host-side, through the c10d store)
    return dt, it


def roofline_pass(trainer, A, B, start_iter):
    """One instrumented D,G,G cycle (eager dispatch, one stream): HIP events around every igemm launch."""
    import torch
    from discogan_modernized_amd import ops
    ui = trainer.args.update_interval
    start_iter = (start_iter + ui - 1) // ui * ui            # align to a D-step
    saved = (trainer.use_graph, trainer.two_streams)
    trainer.use_graph = False
    trainer.two_streams = False          # isolated kernel durations: one stream, one kernel at a time
    ops.PROFILE = []
    ops.PROFILE_HBM = []
    # Below 256 px eager dispatch is host-bound (kernels of 5-70 us): an event pair around a launch would then also time the wait for
    # the host.  So the stream is held (a spin kernel) while the host queues the whole iteration; the kernels then run back to back
    # and the event pairs time the kernels.
    hold_cycles = 0
    if trainer.image_size < 256:
        khz = getattr(torch.cuda.get_device_properties(trainer.device), "clock_rate", 2400000) or 2400000
        hold_cycles = int(0.06 * khz * 1e3)                     # ~60 ms per iteration
    try:
        for k in range(ui):
            if hold_cycles:
                torch.cuda.synchronize()
                torch.cuda._sleep(hold_cycles)
            trainer.train_iteration(A, B, start_iter + k)
        trainer.finish()
        torch.cuda.synchronize()
    finally:
        rec, ops.PROFILE = ops.PROFILE, None
        hrec, ops.PROFILE_HBM = ops.PROFILE_HBM, None
        trainer.use_graph, trainer.two_streams = saved
    hbm = {}
    for name, nbytes, e0, e1 in hrec:
        d = hbm.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += nbytes
        d[2] += e0.elapsed_time(e1)
    hbm = {k: dict(launches=v[0], algorithmic_GB=round(v[1] / 1e9, 3), ms=round(v[2], 3),
                   GBps=round(v[1] / max(v[2], 1e-9) / 1e6, 1),
                   frac_of_8TBps=round(v[1] / max(v[2], 1e-9) / 1e6 / 8000.0, 3)) for k, v in hbm.items()}
    rec = [r for r in rec if r[0] != "head1"]   # K==1 head uses plain reduction kernels, not the MFMA family
    fam = [r for r in rec if r[0] != "c3_fwd"]   # the 3-channel forward is its own streaming kernel (edge.hip)
    flops = sum(r[1] for r in fam)
    ms = sum(r[2].elapsed_time(r[3]) for r in fam)
    by = {}
    for name, f, e0, e1 in rec:
        d = by.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += f
        d[2] += e0.elapsed_time(e1)
    return flops, ms, len(fam), by, hbm, start_iter + ui


PMC_KERNEL = {   # dominant instantiation (most launches x bytes) of each arithmetic's conv family, by kernel-name prefix
    "f32": ("", "void igemm_kernel<0,"),
    "f32x3": ("_f32x3", "void igemm_dma_x3_kernel<"),
    "bf16": ("_bf16", "void igemm_dma_kernel<"),
}


def pmc_traffic(image_size, batch, mfma_dtype="f32"):
    """HBM-side bytes per launch of the dominant instantiation of the conv family from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 + WRITE_SIZE, separate passes; see profiles/README.md); only for the profiled workloads."""
    tag, prefix = PMC_KERNEL[mfma_dtype]
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_per_launch_{image_size}px_bs{batch}{tag}.json")
        if not os.path.exists(path):
            continue
        try:
            ks = json.load(open(path))["kernels"]
            cand = [(k, v) for k, v in ks.items() if k.startswith(prefix)]
            k, v = max(cand, key=lambda kv: kv[1]["launches"] * kv[1]["hbm_MB_per_launch"])
            return int(v["hbm_MB_per_launch"] * 1024 * 1024), f"bytes/launch beyond L2 for {k} from profiles/{os.path.basename(path)} (PMC, offline)"
        except Exception:
            continue
    return None, f"no PMC profile committed for this workload ({mfma_dtype})"


def step_split_ms(trainer, A, B, start_iter, cycles):
    """D-step and G-step time separately: HIP events around single iterations, default dispatch mode."""
    import torch
    ui = trainer.args.update_interval
    it = (start_iter + ui - 1) // ui * ui
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
    out = {f"ms_{k}_step": round(sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1), 3) for k, v in acc.items()}
    out["ms_per_step_whole_cycles"] = round((out["ms_D_step"] + (ui - 1) * out["ms_G_step"]) / ui, 3)
    return out, it + cycles * ui


def cpu_baseline(image_size, batch, budget_s, update_interval=3):
    """Oracle step on the host cores: 1 warm-up iteration + whole D,G,G cycles until the budget is spent."""
    from oracle import discogan_ref as O
    st = O.build_state(image_size=image_size, seed=1234)
    A, B = O.synthetic_batch(batch, image_size, seed=1000)
    O.train_iteration(st, A, B, 1)                      # warm-up (G-step; allocs, oneDNN primitives)
    t0 = time.perf_counter()
    it, n = update_interval, 0
    while True:
        for _ in range(update_interval):
            O.train_iteration(st, A, B, it)
            it += 1
            n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 60:
            break
    return batch * n / dt, dt, n


COMM_NOTE = []


def also_figures(a, head, extra):
    """First-class figures next to the headline (same run, same box): the exact-fp32 MFMA run of the headline workload -- `value` is the
    f32x3 arithmetic since round 3, so round-over-round readers find the quantity BENCH_r01 / r02 reported here -- and the metric's
    other configuration, 64 px / batch 64, on both arithmetics."""
    out = dict(note=("value / dtype above: the headline arithmetic; f32x3 carries every fp32 operand as three bf16 pieces (24 significand bits, "
                     "exact bf16 x bf16 products, fp32 accumulation) and drops the three lowest-order cross products (<= 2^-23 of a product): "
                     "measured closer to fp64 than the exact-fp32 MFMA chain (DESIGN.md 3.1)"))
    S, N = a.image_size, a.batch_size
    for label, key in ((f"{S}px_bs{N}_f32", f"{S}px_bs{N}_f32_mfma"), ("64px_bs64_f32x3", "64px_bs64_f32x3"), ("64px_bs64_f32", "64px_bs64_f32")):
        r = extra.get(key)
        if a.mfma_dtype == "f32" and label == f"{S}px_bs{N}_f32":
            r = head
        if r:
            e = dict(images_per_sec=r["images_per_sec"], ms_per_step=r["ms_per_step"], grouped_launches=r.get("grouped_launches"))
            if "roofline" in r:
                e["roofline_frac"] = r["roofline"]["frac"]
                e["roofline_achieved_tflops"] = r["roofline"]["achieved"]
                e["roofline_peak_tflops"] = r["roofline"]["peak"]
            out[label] = e
    return out


def make_trainer(a, dev, pg, image_size, mfma_dtype=None, graph=None, overlap=None, comm=None, act_dtype=None, group=None):
    from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args
    ov = {"auto": None, "on": True, "off": False, "graph": "graph"}[a.overlap] if overlap is None else overlap
    kw = dict(device=dev, image_size=image_size, seed=1234, process_group=pg,
              use_graph=(not a.no_graph) if graph is None else graph, two_streams=not a.single_stream,
              mfma_dtype=mfma_dtype or a.mfma_dtype, overlap_comm=ov,
              act_dtype=act_dtype or (a.act_dtype if (mfma_dtype or a.mfma_dtype) == "bf16" else "f32"),
              x3_planes=False if a.no_x3_planes else None,
              group_launch={"auto": None, "on": True, "off": False}[a.group_launch] if group is None else group, group_plan=a.group_plan)
    want = comm or a.comm
    # Multi-rank run: "auto" = the library's own RCCL communicator.  dp.ExchangeGroup votes on every rank's readiness BEFORE
    # the collective init (store keys, no collective) and runs the init under a deadline: a failed vote moves ALL ranks to
    # torch.distributed's RCCL together (tr.xg.note, copied into the JSON), a blocked init exits non-zero -- never a hang.
    tr = DiscoGANTrainer(default_args(), comm=want, **kw)
    if tr.xg is not None and tr.xg.note:
        COMM_NOTE.append(tr.xg.note)
        log(tr.xg.note)
    return tr


def measure(a, tr, A, B, batch, world, steps, warmup, roofline=True, split_cycles=0, image_size=None, exposed=False):
    """value + optional roofline leg + optional D/G split for one trainer; returns a dict.
    exposed (N > 1): the same timed window once more with the collectives stubbed (every rank keeps its own sum; the
    exchange path, buckets, events and Adam slices still run) -- the difference is the exchange time the step actually pays."""
    ui = tr.args.update_interval
    # K a multiple of the cycle: start on a D-step (whole D,G,G cycles, exact).  Otherwise start right AFTER a D-step, so
    # the partial cycle at the end holds G-steps only: the window then has the fewest cheap D-steps a K-step window can
    # have (conservative by <= 1.5 %), never the most.
    phase = 0 if steps % ui == 0 else 1
    warmup = warmup + (phase - warmup) % ui
    dt, it = timed_run(tr, A, B, steps, warmup, world)
    res = dict(images_per_sec=round(batch * world * steps / dt, 2), ms_per_step=round(dt / steps * 1e3, 3), steps=steps,
               warmup=warmup, hipgraph=bool(tr.use_graph), allreduce_overlap=("graph" if tr.graph_overlap else bool(tr.overlap_comm)),
               grouped_launches=bool(tr.group_launch))
    live = LIVE_GFLOP_PER_IMAGE.get(image_size or tr.image_size)
    if live:
        res["whole_step_tflops"] = round(batch * steps / dt * live / 1e3, 2)
    if exposed and tr.xg is not None:
        tr.xg.noop = True
        try:
            dt0, it = timed_run(tr, A, B, steps, (phase - it) % ui + ui, world, start_iter=it)
        finally:
            tr.xg.noop = False
        res["ms_per_step_exchange_stubbed"] = round(dt0 / steps * 1e3, 3)
        res["exposed_ms_per_step"] = round((dt - dt0) / steps * 1e3, 3)
    if roofline:
        flops, ms, nlaunch, by, hbm, it = roofline_pass(tr, A, B, it)
        bf = tr.mfma_dtype == "bf16"
        x3 = tr.mfma_dtype == "f32x3"
        # f32x3 issues SIX bf16 MFMAs per algorithmic product block: its ceiling in algorithmic FLOPs is bf16 peak / 6
        peak = MFMA_BF16_PEAK_TFLOPS if bf else (round(MFMA_BF16_PEAK_TFLOPS / 6, 1) if x3 else MFMA_F32_PEAK_TFLOPS)
        ach = flops / (ms * 1e-3) / 1e12
        res["roofline"] = dict(
            bound="mfma",
            kernel=("igemm_dma_kernel<*> + igemm_bf16_dgw_kernel<*> + igemm_kernel<*,PREC=1> (bf16 implicit-GEMM conv family: LDS-DMA tiles on "
                    "v_mfma_f32_16x16x32_bf16 where the GEMM is at least 192 wide, register-staged v_mfma_f32_32x32x16_bf16 tiles elsewhere)" if bf else
                    "igemm_dma_x3_kernel<*> + igemm_kernel<*,PREC=2> (fp32 operands as three bf16 planes -- written once per tensor and staged by "
                    "LDS-DMA where the GEMM is at least 192 wide, split in the conv kernel elsewhere --, six v_mfma_f32_32x32x16_bf16 per "
                    "product block; peak = dense bf16 peak / 6; the exact-fp32 MFMA peak is 157.3)" if x3 else
                    "igemm_kernel<*> (v_mfma_f32_32x32x2_f32 implicit-GEMM conv family)"),
            achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
            launches_per_cycle=nlaunch, algorithmic_gflop_per_cycle=round(flops / 1e9, 2),
            avg_launch_us=round(ms * 1e3 / max(nlaunch, 1), 2),
            by_op={k: dict(launches=v[0], gflop=round(v[1] / 1e9, 2), ms=round(v[2], 3),
                           tflops=round(v[1] / max(v[2], 1e-9) / 1e9, 1)) for k, v in by.items()})
        res["hbm_bound_families"] = hbm
    if split_cycles:
        sp, it = step_split_ms(tr, A, B, it, split_cycles)
        res.update(sp)
    res["_next_iter"] = it
    return res


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, argv))

    # stdout carries exactly ONE line (the JSON): everything else that lands on fd 1 meanwhile -- RCCL prints a version
    # banner there when a communicator is created -- is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback for the product path)")
    backend = os.environ.get("DG_DIST_BACKEND", "nccl")   # "nccl" == RCCL over xGMI; gloo = several ranks rehearsed on ONE GPU
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD
    from discogan_modernized_amd.trainer import synthetic_batch
    dev = torch.device("cuda", local)
    S, N = a.image_size, a.batch_size

    # ---- headline ------------------------------------------------------------------------------------------
    tr = make_trainer(a, dev, pg, S)
    tr.time_comm = world > 1
    A, B = synthetic_batch(N, S, 1000 + rank, dev)
    log(f"models built; {a.warmup} warm-up + {a.steps} timed steps @{S}px batch {N} x {world} GPU")
    head = measure(a, tr, A, B, N, world, a.steps, a.warmup, roofline=not a.no_roofline,
                   split_cycles=0 if a.no_extra else (2 if S == 512 else 4), exposed=world > 1)
    log(f"timed region done: {head['ms_per_step']:.3f} ms/step, {head['images_per_sec']:.1f} img/s")
    comm = None
    if world > 1:
        ms = tr.comm_ms()
        from discogan_modernized_amd import _lib
        rccl_ws = _lib.load().dg_dp_world_size() if tr.xg.transport == "capi" else (dist.get_world_size() if backend == "nccl" else None)
        comm = dict(transport=tr.xg.describe(), backend=backend, world_size=dist.get_world_size(), rccl_world_size=rccl_ws,
                    allreduce_ms_per_D_step=round(ms.get("D", 0.0), 3), allreduce_ms_per_G_step=None,
                    allreduce_overlap=bool(tr.overlap_comm), exchanges=tr.xg.calls,
                    exposed_ms_per_step=head.get("exposed_ms_per_step"),
                    ms_per_step_exchange_stubbed=head.get("ms_per_step_exchange_stubbed"),
                    note_times=("allreduce_ms_per_*_step = summed HIP-event durations of the collectives on the stream they run on "
                                "(with overlap: occupancy of the communication stream, not time the step waits); exposed_ms_per_step = "
                                "ms_per_step minus the same window with the collectives stubbed = what the exchange costs the step"),
                    rccl_instances=(1 if tr.xg.transport == "capi" else None),
                    payload_MB=dict(D=round(tr.optim_dis.numel * 4 / 1e6, 1), G=round(tr.optim_gen.numel * 4 / 1e6, 1)),
                    note=(COMM_NOTE[-1] if COMM_NOTE else None))
        if "G" in ms:
            comm["allreduce_ms_per_G_step"] = round(ms["G"], 3)
            comm["G_step_buckets"] = len(tr._buckets.buckets) if tr._buckets is not None else 1
    roof = head.get("roofline")
    if roof is not None:
        roof["traffic"], roof["traffic_note"] = pmc_traffic(S, N, a.mfma_dtype)
    extra = {k: v for k, v in head.items() if k in ("ms_D_step", "ms_G_step", "ms_per_step_whole_cycles", "whole_step_tflops",
                                                    "hbm_bound_families")}
    if "hbm_bound_families" in extra:
        extra["note_hbm"] = ("HBM-bound kernel families (SURVEY 8(d)): algorithmic bytes / HIP-event time per op in the same "
                             "instrumented single-stream cycle; an op may be several kernels (bn_backward = partial + finalize + apply)")
    tr.close()
    del tr, A, B
    torch.cuda.empty_cache()

    # ---- side measurements (1 GPU only) --------------------------------------------------------------------------
    if world == 1 and not a.no_extra:
        def side(label, image_size, batch, steps, warmup, **kw):
            t0 = time.time()
            t = make_trainer(a, dev, None, image_size, **{k: v for k, v in kw.items() if k in ("mfma_dtype", "graph", "overlap", "comm", "act_dtype", "group")})
            x, y = synthetic_batch(batch, image_size, 1000, dev)
            r = measure(a, t, x, y, batch, 1, steps, warmup, roofline=kw.get("roofline", False), image_size=image_size)
            r.pop("_next_iter", None)
            if "roofline" in r:
                if not kw.get("keep_by_op"):
                    r["roofline"].pop("by_op", None)
                r["roofline"]["traffic"], r["roofline"]["traffic_note"] = pmc_traffic(image_size, batch, kw.get("mfma_dtype") or a.mfma_dtype)
            if not kw.get("keep_by_op"):
                r.pop("hbm_bound_families", None)
            t.close()
            del t, x, y
            torch.cuda.empty_cache()
            extra[label] = r
            log(f"{label}: {r['images_per_sec']:.1f} img/s ({time.time() - t0:.0f} s)")

        for other in ("f32", "bf16", "f32x3"):
            if other != a.mfma_dtype:
                side(f"{S}px_bs{N}_{other}_mfma", S, N, a.steps, a.warmup, mfma_dtype=other, roofline=True, act_dtype="f32",
                     keep_by_op=(other == "f32"))
        if not (a.mfma_dtype == "bf16" and a.act_dtype == "bf16"):
            side(f"{S}px_bs{N}_bf16_mfma_bf16_activations", S, N, a.steps, a.warmup, mfma_dtype="bf16", act_dtype="bf16", roofline=True)
        if S == 512:
            # the metric's FIRST configuration (BASELINE.json "64px bs64"; the reference CLI's default image_size / batch_size): its own legs
            # with a roofline each, on the headline arithmetic and on the exact fp32 MFMA; grouped launches (trainer default at this size)
            side("64px_bs64_f32x3", 64, 64, 60, 9, mfma_dtype="f32x3", roofline=True, keep_by_op=True)
            side("64px_bs64_f32", 64, 64, 60, 9, mfma_dtype="f32", roofline=True, keep_by_op=True)
            side("64px_bs64_f32_ungrouped", 64, 64, 60, 9, mfma_dtype="f32", group=False)
            side("64px_bs256_f32", 64, 256, 30, 9, mfma_dtype="f32", roofline=True)
            side("64px_bs256_bf16_mfma", 64, 256, 30, 9, mfma_dtype="bf16", act_dtype="f32")
            side("64px_bs256_bf16_mfma_bf16_activations", 64, 256, 30, 9, mfma_dtype="bf16", act_dtype="bf16")
            side("64px_bs256_f32x3_mfma", 64, 256, 30, 9, mfma_dtype="f32x3")
        else:
            side("512px_bs32_f32", 512, 32, 12, 6, mfma_dtype="f32", roofline=True)
        # configs[2]'s per-GPU shape in the two data-parallel dispatch modes (world 1, 1-rank RCCL communicator):
        # hipGraph replay + exchange behind it vs eager dispatch with the exchange overlapped
        side("64px_bs64_dp_graph_mode", 64, 64, 30, 9, mfma_dtype="f32", graph=True, overlap=False, comm="capi")
        side("64px_bs64_dp_eager_overlap_mode", 64, 64, 30, 9, mfma_dtype="f32", graph=False, overlap=True, comm="capi")
        side("64px_bs64_dp_graph_overlap_mode", 64, 64, 30, 9, mfma_dtype="f32", graph=True, overlap="graph", comm="capi")
        extra["note_dp_modes"] = ("64 px / 64 per GPU (BASELINE configs[2] per-GPU shape) on ONE GPU with a 1-rank RCCL communicator: "
                                  "the exchange path runs for real, the collectives move nothing. Eager dispatch is host-bound at this "
                                  "size; graph_overlap = the default of data-parallel runs below 256 px since round 4: the iteration replays as "
                                  "a sequence of hipGraphs with gaps, D-step exchange + Adam under the next iteration's first graph, the "
                                  "generators' decoder halves under the encoder half of the backward (trainer._SegCapture).")
        extra["note_side"] = ("side measurements, NOT the headline value; bf16 = conv operands rounded to bf16 on the bf16 MFMA path, fp32 "
                              "accumulate/BatchNorm/weights/Adam (configs[4] arithmetic); f32x3 = fp32-ACCURATE conv products on the bf16 "
                              "MFMA path: each fp32 operand split into three bf16 planes (24 significand bits; by the tensor's producer / once per tensor, "
                              "csrc/igemm_dma_x3.hip), six MFMAs per block, fp32 "
                              "accumulate -- measured closer to fp64 than the exact-fp32 MFMA chain (tests/test_ops_gpu.py::test_conv_f32x3_is_fp32_accurate, "
                              "::test_conv_f32x3_plane_kernel), the training step passes the exact path's fp32 bounds (test_teacher_forced_iterations_f32x3); "
                              "*_bf16_mfma_bf16_activations = BASELINE configs[4]'s arithmetic in full: bf16 MFMA (LDS-DMA conv kernel, bf16-MFMA edge "
                              "kernels), feature maps and their gradients STORED in bf16, fp32 BatchNorm statistics / parameters / parameter gradients / "
                              "losses / master weights / Adam (tests: test_bf16_path_vs_reference_golden_512_n2, test_bf16_activation_storage_training_step)")

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        torch.set_num_threads(min(16, os.cpu_count() or 1))   # the 1-GPU box's CPU share is 16 cores
        log("cpu baseline (oracle) ...")
        cb = 2 if S == 512 else 64
        v, cdt, cn = cpu_baseline(S, cb, 20.0 if S == 512 else 10.0)
        cpu = dict(value=round(v, 3), unit="images/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"oracle/discogan_ref.py (CPU restatement pinned to the reference's golden vectors), image_size={S} "
                          f"batch {cb}" + (" (the same 512 px network and step at a batch the host finishes in seconds; "
                                           "the reference itself measured 0.23 img/s at batch 32 on 8 cores, BASELINE.md section 3)" if S == 512 else " (BASELINE configs[0])")
                          + f", {cn} iterations (whole D,G,G cycles) after 1 warm-up iteration, {cdt:.1f} s")
        if S == 512 and not a.no_extra:
            v2, cdt2, cn2 = cpu_baseline(64, 64, 10.0)
            extra["cpu_baseline_64px_bs64"] = dict(value=round(v2, 3), unit="images/s", cores=torch.get_num_threads(), kind="port",
                                                   sample=f"BASELINE configs[0]: image_size=64 batch 64, {cn2} iterations, {cdt2:.1f} s")
        if a.cpu_baseline_batch > 0:
            log(f"cpu baseline at batch {a.cpu_baseline_batch} (one D,G,G cycle) ...")
            v3, cdt3, cn3 = cpu_baseline(S, a.cpu_baseline_batch, 0.0)
            extra[f"cpu_baseline_{S}px_bs{a.cpu_baseline_batch}"] = dict(
                value=round(v3, 4), unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/discogan_ref.py, image_size={S} batch {a.cpu_baseline_batch}: {cn3} iterations (one D,G,G cycle) after 1 warm-up "
                       f"iteration, {cdt3:.1f} s")
    if rank == 0:
        line = dict(metric="images/sec per DiscoGAN train step", value=head["images_per_sec"], unit="images/s",
                    n_gpus=world, steps=a.steps, warmup=head["warmup"], ms_per_step=head["ms_per_step"],
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype=a.mfma_dtype, data="synthetic",
                    config=dict(workload=WORKLOADS[S].format(b=N) + "; D,G,G cycle, fwd+bwd+Adam, dead backward work skipped"
                                + {"f32": "; exact fp32 MFMA",
                                   "f32x3": "; fp32-accurate conv products on the bf16 matrix path: every fp32 operand carried as three bf16 "
                                            "planes (24 significand bits), exact bf16 x bf16 products, fp32 accumulation (no operand rounding)",
                                   "bf16": "; conv operands rounded to bf16 (bf16 MFMA, fp32 accumulate)"}[a.mfma_dtype]
                                + ("; feature maps stored in bf16, fp32 BatchNorm statistics" if (a.mfma_dtype == "bf16" and a.act_dtype == "bf16") else ""),
                                image_size=S, global_batch=N * world, parallelism=f"dp{world}",
                                hipgraph=head["hipgraph"], hip_streams=1 if a.single_stream else 2,
                                allreduce_overlap=head["allreduce_overlap"],
                                timed_region=("starts on a D-step: whole D,G,G cycles" if a.steps % 3 == 0 else
                                              f"{a.steps} steps are not a multiple of the 3-step cycle: the window starts right after a D-step "
                                              f"({a.steps // 3} D-steps + {a.steps - a.steps // 3} G-steps, the conservative mix); "
                                              "extra.ms_per_step_whole_cycles is the balanced figure")),
                    roofline=roof, cpu_baseline=cpu, comm=comm, also=also_figures(a, head, extra), extra=extra)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        from discogan_modernized_amd import dp
        dp.host_barrier()                 # store keys, not a device collective: no second RCCL communicator at exit either
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
