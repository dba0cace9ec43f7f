"""The DiscoGAN training iteration, re-hosted on the HIP kernels.

Mirrors the loop body of the reference (image_translation.py:336-390; DDP variant
distributed_image_translation.py:466-518): 4 generator passes, 2 reconstruction losses, 4
discriminator passes, GAN + feature-matching losses, the curriculum loss mix, then backward + Adam
of ONE side (D when ``iters % update_interval == 0`` else G).

Differences that do not change any result:
  * dead backward work is skipped: in a D-step the generators run without an autograd graph and the
    fakes are detached; in a G-step the discriminator parameters are frozen, so no D weight-grads
    and no backward through the real passes.  The reference computes those and discards them.
  * data parallelism averages ONLY the stepped side's flat gradient buffer (RCCL all-reduce: one message, or one
    per gradient bucket when the exchange is overlapped with the backward pass), never broadcasts BatchNorm
    buffers (the reference's per-forward broadcast is what makes its DDP backward raise), and keeps BN /
    feature-matching statistics rank-local like DDP does.
  * steady-state iterations replay a captured hipGraph (zero_grad + forward + backward) instead of
    re-dispatching ~600 kernels from Python.
  * the A-side chain (G_A, D_A) and the B-side chain (G_B, D_B) run on two HIP streams and are issued layer by
    layer in lock step (``forward_steps`` generators); every loss term lands in one device vector and the
    curriculum mix / its gradient seeds are one kernel each.
  * ``need_losses=False`` (opt-in, ``--skip_log_only_passes``): a D-step skips the two reconstruction passes,
    which feed only the log line there.  Weights and optimiser state are unaffected, but the generators' BatchNorm
    running statistics then see one forward per D-step instead of two, so saved ``gen_*.pth`` buffers (used by
    eval-mode inference) differ from the reference's; the default keeps the reference's full work.
"""
from __future__ import annotations

import contextlib
from itertools import chain
from types import SimpleNamespace

import torch

from . import dp
from . import functional as F_
from . import losses as L
from . import optim
from .model import Discriminator, Generator

LOG_KEYS = ("gen_loss_A", "gen_loss_B", "fm_loss_A", "fm_loss_B", "recon_loss_A", "recon_loss_B",
            "dis_loss_A", "dis_loss_B", "gen_loss", "dis_loss")

DEFAULTS = dict(learning_rate=2e-4, beta1=0.5, beta2=0.999, weight_decay=0.00001, gan_curriculum=10000,
                starting_rate=0.01, default_rate=0.5, update_interval=3, model_arch="discogan")


def default_args(**over):
    d = dict(DEFAULTS)
    d.update(over)
    return SimpleNamespace(**d)


class DiscoGANTrainer:
    def __init__(self, args=None, device="cuda", image_size=64, seed=1234, process_group=None,
                 use_graph=False, skip_dead_work=True, two_streams=True, overlap_comm=None, mfma_dtype="f32", comm="auto",
                 bucket_mb=128.0, act_dtype="f32", x3_planes=None, group_launch=None, group_plan="launch"):
        self.args = args or default_args()
        for k, v in DEFAULTS.items():
            if not hasattr(self.args, k):
                setattr(self.args, k, v)
        self.device = torch.device(device)
        self.image_size = image_size
        self.pg = process_group
        self.world_size = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        # exchange transport (dp.ExchangeGroup): "capi" = the library's own RCCL communicator (collectives on the
        # caller's stream), "c10d" = torch.distributed, "auto" = capi under backend nccl.  comm="capi" may be forced at
        # world size 1 (tests: the whole exchange path runs against a 1-rank RCCL communicator).
        self.xg = None
        if self.world_size > 1 or comm == "capi":
            self.xg = dp.ExchangeGroup(process_group, transport=comm, device=torch.device(device))
        self.time_comm = False                            # bench: HIP events around every exchange
        self.comm_events = {"D": [], "G": []}
        self.comm_steps = {"D": 0, "G": 0}
        self.use_graph = use_graph
        self.skip_dead_work = skip_dead_work
        if seed is not None:
            torch.manual_seed(seed)                      # distributed_image_translation.py:372
        # construction order G_A, G_B, D_A, D_B on the host RNG (identical replicas on every rank)
        self.generator_A = Generator(extra_layers=True, image_size=image_size).to(self.device)
        self.generator_B = Generator(extra_layers=True, image_size=image_size).to(self.device)
        self.discriminator_A = Discriminator(image_size=image_size).to(self.device)
        self.discriminator_B = Discriminator(image_size=image_size).to(self.device)
        self.nets = dict(gen_A=self.generator_A, gen_B=self.generator_B,
                         dis_A=self.discriminator_A, dis_B=self.discriminator_B)
        self.recon_criterion = L.MSELoss()
        self.gan_criterion = L.BCELoss()
        self.feat_criterion = L.HingeEmbeddingLoss()
        a = self.args
        self.optim_gen = optim.Adam(chain(self.generator_A.parameters(), self.generator_B.parameters()),
                                    lr=a.learning_rate, betas=(a.beta1, a.beta2), weight_decay=a.weight_decay)
        self.optim_dis = optim.Adam(chain(self.discriminator_A.parameters(), self.discriminator_B.parameters()),
                                    lr=a.learning_rate, betas=(a.beta1, a.beta2), weight_decay=a.weight_decay)
        self._graphs = {}
        self._static = None
        # The A-side chain (G_A, D_A) runs on a second HIP stream next to the B-side chain (G_B, D_B):
        # the two are independent except for two hand-overs (AB -> G_A, BA -> G_B), so prologues and
        # tails of one chain's kernels overlap the other chain's kernels.  Every network stays on ONE
        # stream, which keeps its two calls per iteration (and its BN running-stat updates) ordered.
        self.two_streams = two_streams
        self.side_stream = torch.cuda.Stream(device=self.device) if two_streams else None
        # (Round 1-2 experiments on top of the two streams -- CU-masked streams per chain, turn taking on the matrix cores, a phase skew
        # between the chains, weight gradients on a third stream -- all measured slower or equal and were
        # removed in round 4.)
        # mfma_dtype "bf16": the interior conv GEMMs round their operands to bf16 and run on the bf16 matrix path
        # with fp32 accumulation (BASELINE configs[4]); tensors, BatchNorm, losses, master weights, Adam stay fp32.
        # "f32x3": fp32-accurate products on the bf16 matrix path -- every operand is split into three bf16 planes
        # (24 significand bits), six MFMAs per product block, fp32 accumulation (csrc/igemm.hip PREC 2).
        # Process-global library option, set for the lifetime of this trainer's calls.
        if mfma_dtype not in ("f32", "bf16", "f32x3"):
            raise ValueError("mfma_dtype must be 'f32', 'bf16' or 'f32x3'")
        self.mfma_dtype = mfma_dtype
        # bf16 path: the conv kernels read bf16 SHADOWS of their operands written by the producers (ops.SHADOW)
        self.bf16_shadow = mfma_dtype == "bf16"
        # act_dtype "bf16" (needs mfma_dtype "bf16"): feature maps and their gradients are STORED in bf16 only -- the conv
        # epilogues round the fp32 accumulators once, BatchNorm reads bf16 and keeps its statistics and arithmetic in
        # fp32 / fp64 (BASELINE configs[4]: "bf16 MFMA + fp32 BatchNorm accum"); 8 / 10 bytes per element and pass instead
        # of 18 / 22.  Images, weights, BatchNorm parameters and buffers, every parameter gradient, losses, Adam: fp32.
        if act_dtype not in ("f32", "bf16"):
            raise ValueError("act_dtype must be 'f32' or 'bf16'")
        if act_dtype == "bf16" and mfma_dtype != "bf16":
            raise ValueError("act_dtype='bf16' needs mfma_dtype='bf16'")
        self.act_dtype = act_dtype
        if self.bf16_shadow:
            self.optim_gen.enable_bf16_shadow()
            self.optim_dis.enable_bf16_shadow()
        # f32x3 path: the conv kernels read the three bf16 PLANES of their operands, written once per tensor (Adam: weights;
        # ops.planes_of: activations and gradients) instead of splitting every fp32 value in every conv (ops.X3,
        # csrc/igemm_dma_x3.hip); x3_planes=False keeps the register-staged split everywhere.  None = by size: the planes cost
        # 6 B/element of extra BatchNorm output and a weight transpose per Adam step, which the faster conv kernels repay at
        # 512 px / batch 32 (257 -> 275 images/s) and do not at 64 px / batch 256 (25.0 k -> 23.0 k, same-box A/B) -> on from 256 px
        if x3_planes is None:
            x3_planes = image_size >= 256
        self.x3_planes = mfma_dtype == "f32x3" and bool(x3_planes)
        if self.x3_planes:
            self.optim_gen.enable_x3_planes()
            self.optim_dis.enable_x3_planes()
        # The arithmetic and operand forms of THIS trainer's calls: an ops.Context handed to every op with the call (the conv
        # arithmetic is an argument of the C ABI: DG_PREC_*), never a process-wide switch -- two trainers with different arithmetic
        # can be interleaved call by call (tests/test_group_gpu.py::test_interleaved_trainers_with_different_arithmetic).
        from . import ops as _ops
        self.ctx = _ops.Context(prec={"f32": _ops.PREC_F32, "bf16": _ops.PREC_BF16, "f32x3": _ops.PREC_F32X3}[mfma_dtype],
                                shadow=self.bf16_shadow, act16=self.act_dtype == "bf16", x3=self.x3_planes, group_plan=group_plan)
        # Grouped launches (round 4): the A-side / B-side pass of every pair of the iteration (image_translation.py:342-346,353-361)
        # goes out as ONE launch per kernel, and in a D-step each discriminator layer's real and fake pass of both sides as one
        # launch over four problems (same weights fetched once per launch; BatchNorm statistics / feature maps per pass).  Bitwise
        # equal to the ungrouped schedule (tests/test_group_gpu.py).  None = where it pays: below 256 px, where the launches are
        # tens of microseconds (64 px / batch 64: ~500 launches per iteration, a third of them under 6 us), on the arithmetics
        # whose operands are plain fp32 tensors (exact fp32, register-staged f32x3) and the symmetric `discogan` architecture.
        # group_plan "launch" (default): the split-K plan of a grouped conv is sized for all its problems together -- fewer slabs per
        # problem, another (fixed) summation order; "single": every problem keeps the plan of its own launch = bitwise the
        # ungrouped step (test_grouped_training_step_is_bitwise_the_ungrouped_one).
        can_group = (mfma_dtype == "f32" or (mfma_dtype == "f32x3" and not self.x3_planes)) and act_dtype == "f32" and \
            self.args.model_arch == "discogan" and skip_dead_work
        # Measured, same box, 64 px, images/s ungrouped -> grouped (tools/ab_group_64.sh): batch 64 12.7 k -> 14.6 k (f32), 15.4 k -> 17.4 k
        # (f32x3); batch 128 16.6 k -> 17.7 k, 21.5 k -> 22.0 k; batch 256 19.3 k -> 19.4 k, 25.7 k -> 24.5 k: with 4x the rows per launch the
        # two-chain schedule's overlap of one chain's BatchNorm under the other's conv is worth more than the saved launches.  So the
        # automatic mode groups a batch of up to group_max_pixels = 128 x 64 x 64 image pixels per domain (decided per call).
        self._group_auto = group_launch is None
        self.group_max_pixels = 128 * 64 * 64
        if group_launch is None:
            group_launch = can_group and image_size < 256
        if group_launch and not can_group:
            raise ValueError("group_launch needs mfma_dtype f32 (or f32x3 without planes), act_dtype f32, model_arch discogan, skip_dead_work")
        self.group_launch = bool(group_launch)
        self._one = torch.ones((), device=self.device, dtype=torch.float32)
        # Data-parallel exchange overlap: after a D-step the all-reduce of the D gradients and the D Adam
        # step run on a communication stream while the NEXT iteration's generator passes run; the
        # discriminator passes wait on the event.  (A G-step's update is needed by the very next kernel,
        # so it stays on the main stream.)  Needs mid-iteration waits -> eager dispatch, which keeps up
        # with the GPU (measured 14.8 ms vs 15.0 ms under graph replay at 64 px / batch 256).
        # Measured on one MI355X at 64 px / 64 per GPU (BASELINE configs[2]'s per-GPU shape): eager dispatch is
        # host-bound there (8.77 ms/step vs 4.86 under hipGraph replay), so the default for small images keeps the
        # captured graph and runs exchange + Adam behind it; from 256 px the kernels are long enough for eager
        # dispatch and the overlapped exchange is the default.
        # overlap_comm="graph" (round 4; default for data-parallel runs below 256 px on the grouped schedule): the exchange overlaps
        # UNDER hipGraph replay.  The iteration is captured as a SEQUENCE of graphs with host-side gaps between them (_SegCapture):
        #   D-step: one graph; behind it all-reduce + Adam of the discriminators go to the communication stream and run under the
        #           NEXT iteration's first graph (zero_grad + G_A(B) | G_B(A)), whose successor waits for them in the gap;
        #   G-step: [zero_grad + stage-1 forward] gap(wait for the D update) [stage-2 + discriminator passes + losses + the backward pass
        #           down to the stage-1 bottleneck] gap(all-reduce + Adam slices of both DECODERS leave on the communication stream)
        #           [backward of the stage-1 encoders]; the encoders' all-reduce + Adam slices are the only exposed part.
        # The backward is cut with two autograd calls (gradients w.r.t. the bottleneck activations, then from there): the kernels write
        # the parameter gradients into the flat buffer as a side effect, so the first call completes every decoder gradient.
        # On the two-chain schedule (from 256 px, or group_launch=False) the same mode has ONE gap: behind G_A(B) | G_B(A), where the two
        # streams meet anyway -- enough for the D-step half (at 512 px the discriminators' Adam is 1.4-2 ms of HBM-bound work that then runs
        # under the next iteration's first generator passes, also on ONE rank); the G-step update stays behind the backward pass.
        if overlap_comm is None:
            overlap_comm = (True if image_size >= 256 else ("graph" if (self.group_launch and use_graph) else False)) if self.world_size > 1 else False
        self.graph_overlap = overlap_comm == "graph"
        if self.graph_overlap and not skip_dead_work:
            raise ValueError("overlap_comm='graph' needs skip_dead_work")
        self.overlap_comm = bool(overlap_comm) and skip_dead_work and not self.graph_overlap
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.overlap_comm or self.graph_overlap) else None
        self._ev_dis_ready = None
        self._cap = None                                   # the _SegCapture being recorded, if any
        self._split_backward, self._dec_done = True, False
        if self.graph_overlap and self.group_launch:
            self._group_auto = False                       # the backward cut sits in the grouped schedule: grouped at every batch size
        if self.overlap_comm:
            self.use_graph = False
        # G-step exchange overlap: the generators' flat gradient buffer is cut into buckets of whole layers; a
        # bucket is all-reduced (and its Adam slice applied) on the communication stream as soon as the LAST backward
        # pass through its layers has been issued (functional.FINAL_HOOK), while the rest of the backward runs.
        self._buckets = _GradBuckets(self, bucket_mb) if self.overlap_comm else None
        self._eager_until = self.args.update_interval      # first cycle runs eagerly (warm-up)

    # ---------------------------------------------------------------------------------------------
    def active_ranges(self, dstep):
        """Networks that receive gradients in this kind of step (image_translation.py:374-382):
        discogan: both of the stepped side; recongan: D_B | G_A+G_B (G_B through recon_A's cycle);
        gan: D_B | G_B only.  Returned as flat ranges of the stepped optimiser (None = all)."""
        arch = self.args.model_arch
        if arch == "discogan":
            return None
        if dstep:
            return self.optim_dis.ranges_of([self.discriminator_B])
        if arch == "gan":
            return self.optim_gen.ranges_of([self.generator_B])
        return None                                       # recongan: ABA = G_A(G_B(A)) reaches both generators

    def is_dis_step(self, iters):
        return iters % self.args.update_interval == 0          # image_translation.py:385

    def rate(self, iters):
        a = self.args
        return a.starting_rate if iters < a.gan_curriculum else a.default_rate   # :367

    def _set_requires_grad(self, dstep):
        if not self.skip_dead_work:
            return
        for p in self.optim_dis.params:
            p.requires_grad_(dstep)

    def forward_losses(self, A, B, iters, need_losses=True):
        """image_translation.py:342-382.  need_losses=False (D-steps only): the reconstruction passes ABA / BAB and their
        MSE terms feed nothing but the log line in a D-step, so they are not computed (recon_loss_* read NaN)."""
        if self.group_launch and (not self._group_auto or A.shape[0] * A.shape[2] * A.shape[3] <= self.group_max_pixels):
            return self._forward_losses_grouped(A, B, iters, need_losses)
        a = self.args
        dstep = self.is_dis_step(iters)
        skip = self.skip_dead_work
        gen_ctx = torch.no_grad if (skip and dstep) else torch.enable_grad
        main = torch.cuda.current_stream(self.device)
        side = self.side_stream if self.two_streams else main
        on_side = (lambda: torch.cuda.stream(side)) if self.two_streams else contextlib.nullcontext
        # loss vector (layout: dg_loss_mix_fwd): allocated BEFORE the fork so that the side stream's writes are
        # ordered after whatever used this block on the main stream
        nfm = self.discriminator_A.n_stages - 1
        lv = torch.empty(8 + 2 * nfm, device=self.device, dtype=torch.float32)
        if self.two_streams:
            side.wait_stream(main)
            # tensors that cross streams tell the caching allocator (a block is otherwise reusable on its
            # allocation stream as soon as the host drops it, while the other stream may still read it)
            for t_ in (A, B, lv):
                t_.record_stream(side)
        # The A-side chain (G_A, D_A) is issued on `side`, the B-side chain (G_B, D_B) on `main`, layer by layer
        # in lock step (model.forward_steps): host issue order A.l1, B.l1, A.l2, B.l2, ...  autograd replays nodes in reverse
        # creation order, so the backward pass is interleaved the same way.

        def pair(gen_a, gen_b):
            ra = rb = pend = object()
            while ra is pend or rb is pend:
                if ra is pend:
                    with on_side():
                        try:
                            next(gen_a)
                        except StopIteration as e:
                            ra = e.value
                if rb is pend:
                    try:
                        next(gen_b)
                    except StopIteration as e:
                        rb = e.value
            return ra, rb

        want_recon = need_losses or not (dstep and skip)
        BA, AB, ABA, BAB, A_dis_real, A_feats_real, B_dis_real, B_feats_real, A_dis_fake, A_feats_fake, B_dis_fake, B_feats_fake = \
            self._two_chain_forward(A, B, lv, pair, gen_ctx, main, side, want_recon)
        assert nfm == len(A_feats_real)
        sl = [lv[i] for i in range(8 + 2 * nfm)]
        terms = {}

        def bce(p, label, k):
            terms[k] = F_.BCELossFn.apply(p.reshape(p.size(0), -1), label, sl[k])

        with on_side():
            if want_recon:
                with gen_ctx():
                    terms[0] = F_.MSELossFn.apply(ABA, A, sl[0])
            else:
                terms[0] = sl[0]
            bce(A_dis_real, 1.0, 2); bce(A_dis_fake, 0.0, 3); bce(A_dis_fake, 1.0, 4)
            for l, (r, f) in enumerate(zip(A_feats_real, A_feats_fake)):
                terms[8 + l] = F_.FeatureMatchFn.apply(r, f, sl[8 + l])
        if want_recon:
            with gen_ctx():
                terms[1] = F_.MSELossFn.apply(BAB, B, sl[1])
        else:
            terms[1] = sl[1]
        bce(B_dis_real, 1.0, 5); bce(B_dis_fake, 0.0, 6); bce(B_dis_fake, 1.0, 7)
        for l, (r, f) in enumerate(zip(B_feats_real, B_feats_fake)):
            terms[8 + nfm + l] = F_.FeatureMatchFn.apply(r, f, sl[8 + nfm + l])
        if self.two_streams:
            main.wait_stream(side)
        rate = self.rate(iters)
        arch = {"discogan": 0, "recongan": 1, "gan": 2}.get(a.model_arch)
        if arch is None:
            raise ValueError(f"unknown model_arch {a.model_arch}")
        fmA, fmB = list(range(8, 8 + nfm)), list(range(8 + nfm, 8 + 2 * nfm))
        if dstep:
            which, idx = 7, ([2, 3, 5, 6] if arch == 0 else [5, 6])
        else:
            which = 6
            idx = ([0, 1, 4, 7] + fmA + fmB) if arch == 0 else (([0, 7] if arch == 1 else [7]) + fmB)
        idx = tuple(idx)
        (gen_loss_A, gen_loss_B, fm_loss_A, fm_loss_B, dis_loss_A, dis_loss_B, gen_loss, dis_loss) = \
            F_.LossMixFn.apply(lv, nfm, rate, arch, which, idx, *[terms[i] for i in idx])
        recon_loss_A, recon_loss_B = terms[0], terms[1]
        return SimpleNamespace(
            gen_loss=gen_loss, dis_loss=dis_loss, gen_loss_A=gen_loss_A, gen_loss_B=gen_loss_B,
            fm_loss_A=fm_loss_A, fm_loss_B=fm_loss_B, recon_loss_A=recon_loss_A, recon_loss_B=recon_loss_B,
            dis_loss_A=dis_loss_A, dis_loss_B=dis_loss_B, AB=AB, BA=BA, ABA=ABA, BAB=BAB,
            A_dis_real=A_dis_real, A_dis_fake=A_dis_fake, B_dis_real=B_dis_real, B_dis_fake=B_dis_fake,
            A_feats_real=A_feats_real, B_feats_fake=B_feats_fake, lossvec=lv)

    def _two_chain_forward(self, A, B, lv, pair, gen_ctx, main, side, want_recon):
        """The symmetric two-chain order: G_A(B) | G_B(A), G_A(AB) | G_B(BA), D_A(A) | D_B(B), D_A(BA) | D_B(AB), each pair in lock step.
        (Round 4 tried a DEPHASED order -- side: D_A(A), G_A(B), G_A(AB), D_A(BA); main: G_B(A), D_B(B), G_B(BA), D_B(AB) -- so that one
        chain's HBM-bound BatchNorm kernels would meet the other chain's conv kernels instead of its BatchNorm kernels: bitwise
        neutral, and no faster: 295.0 / 293.9 -> 294.5 / 294.8 images/s (f32x3), 1014.9 -> 1002.8 (bf16), 194.2 -> 192.9 (f32) at
        512 px / batch 32, same box, alternating, profiles/r04_ab_dephased_chain_order.txt.  Removed.)"""
        # stage 1: the two first-stage translations are independent.  Their backward passes are the LAST ones to
        # touch each generator's parameters (autograd replays in reverse), which the bucketed exchange keys on.
        F_.FINAL_PASS = True
        try:
            with gen_ctx():
                BA, AB = pair(self.generator_A.forward_steps(B), self.generator_B.forward_steps(A))
        finally:
            F_.FINAL_PASS = False
        if self.two_streams:
            AB.record_stream(side)
            BA.record_stream(main)
        if self.graph_overlap:
            # a gap between two captured graphs (the wait for the discriminators' update happens there at replay): the chains join
            # in front of it and fork again behind it, which is also the hand-over of AB / BA
            if self.two_streams:
                main.wait_stream(side)
            self._cut("dis_ready")
            if self.two_streams:
                side.wait_stream(main)
        elif self.two_streams:
            ev_ab, ev_ba = torch.cuda.Event(), torch.cuda.Event()
            ev_ab.record(main)
            ev_ba.record(side)
            side.wait_event(ev_ab)                           # G_A(AB) needs AB
            main.wait_event(ev_ba)                           # G_B(BA) needs BA
        if self._ev_dis_ready is not None:                   # D parameters updated on the comm stream
            main.wait_event(self._ev_dis_ready)
            if self.two_streams:
                side.wait_event(self._ev_dis_ready)
            self._ev_dis_ready = None
        # stage 2 + discriminators.  Every loss term is written into its slot of one device vector (layout:
        # dg_loss_mix_fwd); the mix and its gradient seeds are one launch each instead of ~45 scalar kernels.
        ABA = BAB = None
        if want_recon:
            with gen_ctx():
                ABA, BAB = pair(self.generator_A.forward_steps(AB), self.generator_B.forward_steps(BA))
        else:
            lv[:2].fill_(float("nan"))
        (A_dis_real, A_feats_real), (B_dis_real, B_feats_real) = pair(
            self.discriminator_A.forward_steps(A), self.discriminator_B.forward_steps(B))
        (A_dis_fake, A_feats_fake), (B_dis_fake, B_feats_fake) = pair(
            self.discriminator_A.forward_steps(BA), self.discriminator_B.forward_steps(AB))
        return BA, AB, ABA, BAB, A_dis_real, A_feats_real, B_dis_real, B_feats_real, A_dis_fake, A_feats_fake, B_dis_fake, B_feats_fake

    def _forward_losses_grouped(self, A, B, iters, need_losses=True):
        """image_translation.py:342-382 with every pair of passes as grouped launches (model.group_generators /
        group_discriminators).  Main stream: G_A(B) | G_B(A), then G_A(AB) | G_B(BA) + the two reconstruction terms; side stream
        (from the hand-over of AB / BA on): the discriminator passes + GAN / feature-matching terms -- D-step: D_A(A) | D_A(BA) |
        D_B(B) | D_B(AB) as ONE group of four, G-step: the real pair without an autograd graph, then the fake pair."""
        from .model import group_discriminators, group_generators
        a = self.args
        dstep = self.is_dis_step(iters)
        gen_ctx = torch.no_grad if dstep else torch.enable_grad
        main = torch.cuda.current_stream(self.device)
        side = self.side_stream if self.two_streams else main
        on_side = (lambda: torch.cuda.stream(side)) if self.two_streams else contextlib.nullcontext
        gA, gB, dA, dB = self.generator_A, self.generator_B, self.discriminator_A, self.discriminator_B
        nfm = dA.n_stages - 1
        lv = torch.empty(8 + 2 * nfm, device=self.device, dtype=torch.float32)
        sl = [lv[i] for i in range(8 + 2 * nfm)]
        terms = {}
        F_.FINAL_PASS = True        # the backward of these two passes is the last to touch each generator's parameters
        try:
            cut = None
            with gen_ctx():
                if self.graph_overlap and not dstep and self._split_backward:
                    (BA, AB), hs, leaves = group_generators([gA, gB], [B, A], cut_at_bottleneck=True)
                    cut = (hs, leaves)
                else:
                    BA, AB = group_generators([gA, gB], [B, A])
        finally:
            F_.FINAL_PASS = False
        if self.graph_overlap:
            self._cut("dis_ready")                           # (a gap between two captured graphs: the wait happens at replay)
        elif self._ev_dis_ready is not None:                 # D parameters updated on the communication stream
            main.wait_event(self._ev_dis_ready)
            self._ev_dis_ready = None
        if self.two_streams:
            side.wait_stream(main)
            for t_ in (A, B, lv, AB, BA):
                t_.record_stream(side)
        want_recon = need_losses or not dstep
        ABA = BAB = None
        if want_recon:
            with gen_ctx():
                ABA, BAB = group_generators([gA, gB], [AB, BA])
                terms[0], terms[1] = F_.MSELossGroupFn.apply(2, ABA, BAB, A, B, sl[0], sl[1])
        else:
            lv[:2].fill_(float("nan"))
            terms[0], terms[1] = sl[0], sl[1]
        with on_side():
            if dstep:
                (A_dis_real, A_feats_real), (A_dis_fake, A_feats_fake), (B_dis_real, B_feats_real), (B_dis_fake, B_feats_fake) = \
                    group_discriminators([dA, dA, dB, dB], [A, BA, B, AB])
            else:
                with torch.no_grad():
                    (A_dis_real, A_feats_real), (B_dis_real, B_feats_real) = group_discriminators([dA, dB], [A, B])
                (A_dis_fake, A_feats_fake), (B_dis_fake, B_feats_fake) = group_discriminators([dA, dB], [BA, AB])
            # GAN terms (image_translation.py:157-166): slots 2 / 5 bce(real, 1), 4 / 7 bce(fake, 1), 3 / 6 bce(fake, 0)
            terms[2], terms[4], terms[5], terms[7] = F_.BCELossGroupFn.apply(
                4, (1.0, 1.0, 1.0, 1.0), A_dis_real, A_dis_fake, B_dis_real, B_dis_fake, sl[2], sl[4], sl[5], sl[7])
            terms[3], terms[6] = F_.BCELossGroupFn.apply(2, (0.0, 0.0), A_dis_fake, B_dis_fake, sl[3], sl[6])
            for l in range(nfm):
                terms[8 + l], terms[8 + nfm + l] = F_.FeatureMatchGroupFn.apply(
                    2, A_feats_real[l], B_feats_real[l], A_feats_fake[l], B_feats_fake[l], sl[8 + l], sl[8 + nfm + l])
        if self.two_streams:
            main.wait_stream(side)
        rate = self.rate(iters)
        fmA, fmB = list(range(8, 8 + nfm)), list(range(8 + nfm, 8 + 2 * nfm))
        which, idx = (7, (2, 3, 5, 6)) if dstep else (6, tuple([0, 1, 4, 7] + fmA + fmB))
        (gen_loss_A, gen_loss_B, fm_loss_A, fm_loss_B, dis_loss_A, dis_loss_B, gen_loss, dis_loss) = \
            F_.LossMixFn.apply(lv, nfm, rate, 0, which, idx, *[terms[i] for i in idx])
        return SimpleNamespace(
            gen_loss=gen_loss, dis_loss=dis_loss, gen_loss_A=gen_loss_A, gen_loss_B=gen_loss_B,
            fm_loss_A=fm_loss_A, fm_loss_B=fm_loss_B, recon_loss_A=terms[0], recon_loss_B=terms[1],
            dis_loss_A=dis_loss_A, dis_loss_B=dis_loss_B, AB=AB, BA=BA, ABA=ABA, BAB=BAB,
            A_dis_real=A_dis_real, A_dis_fake=A_dis_fake, B_dis_real=B_dis_real, B_dis_fake=B_dis_fake,
            A_feats_real=A_feats_real, B_feats_fake=B_feats_fake, lossvec=lv, backward_cut=cut)

    # ---- the iteration as a sequence of graphs with gaps (overlap_comm="graph") --------------------------------------------------
    def _cut(self, name):
        """A point of the iteration where the host must act between kernels: while a _SegCapture records, the current graph ends
        here and the next one begins (the action runs in the gap at every replay); in eager dispatch the action runs right away."""
        if self._cap is not None:
            self._cap.cut(name)
        else:
            self._gap(name)

    def _gap(self, name):
        main = torch.cuda.current_stream(self.device)
        if name == "dis_ready":
            if self._ev_dis_ready is not None:               # the discriminators' all-reduce + Adam of the last D-step (communication stream)
                main.wait_event(self._ev_dis_ready)
                self._ev_dis_ready = None
        elif name == "dec_bucket":
            # every gradient of both decoders is final: their exchange + Adam slices leave while the encoders' backward replays
            opt = self.optim_gen
            ev = torch.cuda.Event()
            ev.record(main)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self._adam_begin(opt)
                self._exchange_and_step_ranges(opt, self._gen_ranges()[1], "G")
            self._dec_done = True
        else:
            raise ValueError(name)

    def _gen_ranges(self):
        """(encoder ranges, decoder ranges) of the generators in the flat buffer of optim_gen."""
        if getattr(self, "_gen_rng", None) is None:
            enc = self.optim_gen.ranges_of([self.generator_A.encoder, self.generator_B.encoder])
            dec = self.optim_gen.ranges_of([self.generator_A.decoder, self.generator_B.decoder])
            self._gen_rng = (enc, dec)
        return self._gen_rng

    def _adam_begin(self, opt):
        g = opt.param_groups[0]
        opt._sync_foreign_grads()
        from . import ops
        ops.adam_advance(opt.state, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]))

    def _exchange_and_step_ranges(self, opt, ranges, kind):
        """all-reduce(sum) + Adam of flat ranges on the current stream (the step state was advanced by _adam_begin)."""
        from . import ops
        g = opt.param_groups[0]
        for b, e in ranges:
            scale = 1.0
            if self.xg is not None:
                if self.time_comm:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    scale = self.xg.all_reduce_sum_(opt.flat_g[b:e])
                    e1.record()
                    self.comm_events[kind].append((e0, e1))
                else:
                    scale = self.xg.all_reduce_sum_(opt.flat_g[b:e])
            ops.adam_step_flat(opt.flat_p[b:e], opt.flat_g[b:e], opt.exp_avg[b:e], opt.exp_avg_sq[b:e], opt.state,
                               float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(scale))

    def _fwd_bwd(self, A, B, iters, need_losses=True):
        dstep = self.is_dis_step(iters)
        self._set_requires_grad(dstep)
        if self.skip_dead_work:
            # only the stepped side's gradients are written this iteration; the other flat buffer is
            # left alone (it may still be in flight on the communication stream)
            if dstep and self._ev_dis_ready is not None and self._cap is None:
                torch.cuda.current_stream(self.device).wait_event(self._ev_dis_ready)
            (self.optim_dis if dstep else self.optim_gen).zero_grad()
        else:
            self.optim_gen.zero_grad()                   # image_translation.py:336-339
            self.optim_dis.zero_grad()
        from . import ops as _ops
        # (the attributes may have been switched between iterations: tests run one trainer in several arithmetics)
        self.ctx.prec = {"f32": _ops.PREC_F32, "bf16": _ops.PREC_BF16, "f32x3": _ops.PREC_F32X3}[self.mfma_dtype]
        self.ctx.shadow, self.ctx.act16, self.ctx.x3 = bool(self.bf16_shadow), self.act_dtype == "bf16", bool(self.x3_planes)
        self.ctx.clear()
        try:
            with _ops.use(self.ctx):           # this trainer's arithmetic / operand forms ride with every call
                out = self.forward_losses(A, B, iters, need_losses)
                cut = getattr(out, "backward_cut", None)
                if cut is not None:
                    # backward in two calls: down to the stage-1 bottleneck leaves (completes every decoder gradient of both
                    # generators -- stage-2 passes included --, written into the flat buffer by the kernels), gap, then the encoders
                    out.gen_loss.backward(gradient=self._one)
                    if self.two_streams:
                        torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
                    self._cut("dec_bucket")
                    hs, leaves = cut
                    torch.autograd.backward(list(hs), [l.grad for l in leaves])
                    out.backward_cut = None
                else:
                    (out.dis_loss if dstep else out.gen_loss).backward(gradient=self._one)
        finally:
            self.ctx.clear()
        if self.skip_dead_work and not dstep:
            for p in self.optim_dis.params:               # a G-step froze the D parameters: give them back
                p.requires_grad_(True)
        if self.two_streams:
            # the backward kernels of the A-side chain ran on the side stream and wrote the flat gradient
            # buffer directly (no AccumulateGrad leaf for autograd to sync): join before Adam / all-reduce
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
        return out

    def _graph_key(self, A, B, iters, need_losses=True):
        return ("D" if self.is_dis_step(iters) else "G", self.rate(iters), bool(need_losses) or not self.is_dis_step(iters),
                tuple(A.shape), tuple(B.shape))

    def _fwd_bwd_graphed(self, A, B, iters, need_losses=True):
        # static input buffers and captured graphs are per batch shape (a short last batch of an epoch gets its own)
        shp = (tuple(A.shape), tuple(B.shape))
        if self._static is None:
            self._static = {}
        if shp not in self._static:
            self._static[shp] = (torch.empty_like(A), torch.empty_like(B))
        sA, sB = self._static[shp]
        sA.copy_(A)
        sB.copy_(B)
        key = self._graph_key(A, B, iters, need_losses)
        ent = self._graphs.get(key)
        if ent is None:
            if self.graph_overlap:
                cap = _SegCapture(self)
                self._cap = cap
                try:
                    with cap:
                        out = self._fwd_bwd(sA, sB, iters, need_losses)
                finally:
                    self._cap = None
                ent = (cap, out)
            else:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    out = self._fwd_bwd(sA, sB, iters, need_losses)
                ent = (g, out)
            self._graphs[key] = ent
        ent[0].replay()
        return ent[1]

    def _exchange(self, opt, kind):
        """All-reduce (sum) of the stepped side's flat gradient buffer on the current stream; returns the 1/W scale."""
        if self.xg is None:
            return 1.0
        if self.time_comm:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            scale = self.xg.all_reduce_sum_(opt.flat_g)
            e1.record()
            self.comm_events[kind].append((e0, e1))
            return scale
        return self.xg.all_reduce_sum_(opt.flat_g)

    def train_iteration(self, A, B, iters, do_step=True, need_losses=True):
        """One full iteration; returns the namespace of (device) loss scalars.  need_losses=False tells the
        trainer that this iteration's loss values will not be read (no log line): a D-step then skips the two
        reconstruction passes, which feed only the log (weights and optimiser state are unaffected)."""
        dstep = self.is_dis_step(iters)
        opt = self.optim_dis if dstep else self.optim_gen
        if self.graph_overlap:
            return self._train_iteration_graph_overlap(A, B, iters, do_step, need_losses)
        bucketed = self._buckets is not None and not dstep and do_step and self.xg is not None
        if self.time_comm:
            self.comm_steps["D" if dstep else "G"] += 1
        if bucketed:
            self._buckets.begin(opt, self.active_ranges(dstep))
        try:
            if self.use_graph and iters >= self._eager_until:       # first cycle (and the first after a resume) runs eagerly
                out = self._fwd_bwd_graphed(A, B, iters, need_losses)
            else:
                out = self._fwd_bwd(A, B, iters, need_losses)
        finally:
            if bucketed:
                F_.FINAL_HOOK = None
        if bucketed:
            self._buckets.finish()
            return out
        # gradients of the stepped side only: one flat message, summed; the /W rides in the Adam kernel
        if self.overlap_comm and dstep and do_step:
            main = torch.cuda.current_stream(self.device)
            ev = torch.cuda.Event()
            ev.record(main)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                scale = self._exchange(opt, "D")
                opt.step(grad_scale=scale, active=self.active_ranges(dstep))
                self._ev_dis_ready = torch.cuda.Event()
                self._ev_dis_ready.record(self.comm_stream)
            return out
        scale = self._exchange(opt, "D" if dstep else "G")
        if do_step:
            opt.step(grad_scale=scale, active=self.active_ranges(dstep))
        return out

    def _train_iteration_graph_overlap(self, A, B, iters, do_step, need_losses):
        """overlap_comm="graph": see __init__.  Results are bitwise those of the plain schedule: the all-reduce and the Adam kernel are
        elementwise, the slices share one step state."""
        dstep = self.is_dis_step(iters)
        opt = self.optim_dis if dstep else self.optim_gen
        main = torch.cuda.current_stream(self.device)
        if self.time_comm:
            self.comm_steps["D" if dstep else "G"] += 1
        if dstep and self._ev_dis_ready is not None:          # (a D-step right behind a D-step: update_interval 1)
            main.wait_event(self._ev_dis_ready)
            self._ev_dis_ready = None
        self._split_backward, self._dec_done = bool(do_step), False
        if self.use_graph and iters >= self._eager_until and do_step:
            out = self._fwd_bwd_graphed(A, B, iters, need_losses)
        else:
            out = self._fwd_bwd(A, B, iters, need_losses)
        if not do_step:
            return out
        if not dstep and not self._dec_done:
            # the backward was not cut (two-chain schedule): the generators' update is needed by the very next kernel -- main stream
            scale = self._exchange(opt, "G")
            opt.step(grad_scale=scale, active=self.active_ranges(dstep))
            return out
        ev = torch.cuda.Event()
        ev.record(main)
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            if dstep:
                scale = self._exchange(opt, "D")
                opt.step(grad_scale=scale, active=self.active_ranges(dstep))
                self._ev_dis_ready = torch.cuda.Event()
                self._ev_dis_ready.record(self.comm_stream)
            else:
                self._exchange_and_step_ranges(opt, self._gen_ranges()[0], "G")      # the encoders: the exposed part
        if not dstep:
            main.wait_stream(self.comm_stream)                # the next iteration's first kernels read the generators' weights
        return out

    def comm_ms(self):
        """All-reduce time per D-step / per G-step (ms; a G-step's buckets are summed) from the recorded HIP events
        (time_comm=True)."""
        torch.cuda.synchronize(self.device)
        out = {}
        for k, evs in self.comm_events.items():
            if evs and self.comm_steps[k]:
                out[k] = sum(a.elapsed_time(b) for a, b in evs) / self.comm_steps[k]
        return out

    def finish(self):
        """Join the communication stream (call before reading parameters / saving / timing)."""
        if self._ev_dis_ready is not None:
            torch.cuda.current_stream(self.device).wait_event(self._ev_dis_ready)

    # ---------------------------------------------------------------------------------------------
    def losses_to_floats(self, out):
        vals = torch.stack([getattr(out, k).detach().reshape(()) for k in LOG_KEYS]).cpu().tolist()
        return dict(zip(LOG_KEYS, vals))

    def format_log(self, iters, total, out):
        f = self.losses_to_floats(out)
        return (f"Iter [{iters}/{total}] "
                f"GEN: {f['gen_loss_A']:.4f}/{f['gen_loss_B']:.4f}, "
                f"FM: {f['fm_loss_A']:.4f}/{f['fm_loss_B']:.4f}, "
                f"RECON: {f['recon_loss_A']:.4f}/{f['recon_loss_B']:.4f}, "
                f"DIS: {f['dis_loss_A']:.4f}/{f['dis_loss_B']:.4f}")

    def state_dicts(self):
        return {k: v.state_dict() for k, v in self.nets.items()}

    def train_state(self, next_iter, extra=None):
        """Everything needed to resume exactly: weights + BN buffers, both Adam states and ``next_iter`` = the index
        of the NEXT iteration to run (the reference resumes weights only and restarts Adam and the GAN curriculum,
        SURVEY.md section 5).  ``extra`` carries the caller's data-loader position (CLI: epoch, batch, RNG state)."""
        self.finish()
        torch.cuda.synchronize(self.device)
        st = dict(iters=int(next_iter), nets={k: {n: t.detach().contiguous().cpu() for n, t in v.state_dict().items()}
                                              for k, v in self.nets.items()},
                  optim_gen=self.optim_gen.state_dict(), optim_dis=self.optim_dis.state_dict())
        if extra:
            st["loader"] = dict(extra)
        return st

    def load_train_state(self, st):
        for k, v in self.nets.items():
            v.load_state_dict(st["nets"][k])
        self.optim_gen.load_state_dict(st["optim_gen"])
        self.optim_dis.load_state_dict(st["optim_dis"])
        self._graphs = {}
        start = int(st["iters"])
        # like a fresh run, the first D,G,G cycle after a resume is dispatched eagerly before anything is captured
        self._eager_until = start + self.args.update_interval
        return start

    def close(self):
        if self.xg is not None:
            self.finish()
            torch.cuda.synchronize(self.device)
            self.xg.close()
            self.xg = None


class _SegCapture:
    """One training iteration captured as a SEQUENCE of hipGraphs sharing a memory pool, with named host-side gaps between them
    (trainer._cut): at replay the graphs are launched in order and the trainer's gap action (an event wait, a hand-over to the
    communication stream) runs between two of them -- what a single graph cannot hold, because a captured stream cannot wait on an
    event recorded outside the capture.  Mirrors torch.cuda.graph's protocol (side capture stream, global capture error mode)."""

    def __init__(self, trainer):
        self.tr = trainer
        self.graphs, self.gaps = [], []
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(device=trainer.device)
        self._ctx = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.pool, capture_error_mode="global")
        self.graphs.append(g)

    def __enter__(self):
        import gc
        torch.cuda.synchronize(self.tr.device)
        gc.collect()
        torch.cuda.empty_cache()
        self.stream.wait_stream(torch.cuda.current_stream(self.tr.device))
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        self._begin()
        return self

    def cut(self, name):
        self.graphs[-1].capture_end()
        self.gaps.append(name)
        self._begin()

    def __exit__(self, et, ev, tb):
        try:
            self.graphs[-1].capture_end()
        finally:
            self._ctx.__exit__(et, ev, tb)
        torch.cuda.current_stream(self.tr.device).wait_stream(self.stream)
        return False

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.gaps):
                self.tr._gap(self.gaps[i])


class _GradBuckets:
    """Overlap of the G-step gradient exchange with the backward pass (reference: DDP's bucketed reducer hooks,
    distributed_image_translation.py:401-404,513-518).

    The generators' flat gradient range is cut into buckets of whole layers (>= ``bucket_mb`` each, never across
    the two networks).  A parameter's gradient is final once the backward of its network's FIRST forward call of the
    iteration (issued last by autograd) has written it; functional.py reports that through FINAL_HOOK right after
    enqueueing the kernel.  When every parameter of a bucket has reported, an event is recorded on the producing
    stream and the communication stream runs all-reduce(bucket) + Adam(bucket slice) behind it, concurrently with
    the rest of the backward.  Buckets that never report (networks outside the loss in recongan / gan, or the
    last-to-finish ones) are flushed after the backward.  Results are bitwise those of the one-message path: the
    all-reduce is elementwise and the Adam kernel is elementwise with one shared step state."""

    def __init__(self, trainer, bucket_mb):
        self.tr = trainer
        opt = trainer.optim_gen
        target = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []
        self.of_param = {}
        for net in (trainer.generator_A, trainer.generator_B):
            ids = {id(p) for p in net.parameters()}
            cur = None
            for p, off in zip(opt.params, opt.offsets):
                if id(p) not in ids:
                    continue
                end = off + (p.numel() + optim._ALIGN - 1) // optim._ALIGN * optim._ALIGN
                if cur is None or cur["end"] - cur["begin"] >= target:
                    cur = dict(begin=off, end=end, nparams=0, pending=0, done=False)
                    self.buckets.append(cur)
                cur["end"] = end
                cur["nparams"] += 1
                self.of_param[id(p)] = cur
        self.opt = opt
        self.launched = 0
        self.active = None

    def begin(self, opt, active=None):
        """``active``: the flat ranges that receive gradients in this step (trainer.active_ranges; None = all).  The same
        ranges gate the Adam slice of EVERY bucket, whether it is launched early from the backward or flushed afterwards --
        exactly what the one-message path's ``opt.step(active=...)`` does."""
        assert opt is self.opt
        self.active = active
        tr = self.tr
        main = torch.cuda.current_stream(tr.device)
        for b in self.buckets:
            b["pending"], b["done"] = b["nparams"], False
        self.launched = 0
        # the step counter / bias corrections advance once, on the communication stream, behind everything queued
        tr.comm_stream.wait_stream(main)
        with torch.cuda.stream(tr.comm_stream):
            g = opt.param_groups[0]
            opt._sync_foreign_grads()
            from . import ops
            ops.adam_advance(opt.state, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]))
        F_.FINAL_HOOK = self.notify

    def notify(self, param):
        b = self.of_param.get(id(param))
        if b is None or b["done"]:
            return
        # the kernels accumulated into the flat gradient view; a .grad replaced during the backward would not be exchanged
        if param.grad is not param._dg_flat_grad:
            raise RuntimeError("bucketed exchange: a parameter's .grad was replaced during the backward pass "
                               "(it must stay the view of the optimiser's flat gradient buffer)")
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b, torch.cuda.current_stream(self.tr.device), early=True, active=self.active)

    def _launch(self, b, producer_stream, early, active=None):
        tr, opt = self.tr, self.opt
        b["done"] = True
        ev = torch.cuda.Event()
        ev.record(producer_stream)
        tr.comm_stream.wait_event(ev)
        with torch.cuda.stream(tr.comm_stream):
            sl = slice(b["begin"], b["end"])
            scale = 1.0
            if tr.xg is not None:
                if tr.time_comm:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    scale = tr.xg.all_reduce_sum_(opt.flat_g[sl])
                    e1.record()
                    tr.comm_events["G"].append((e0, e1))
                else:
                    scale = tr.xg.all_reduce_sum_(opt.flat_g[sl])
            if active is None or any(lo <= b["begin"] and b["end"] <= hi for lo, hi in active):
                g = opt.param_groups[0]
                from . import ops
                p16 = getattr(opt, "flat_p16", None)
                p3 = getattr(opt, "flat_p3", None)          # f32x3 plane path: the slice's plane triples with the update
                ops.adam_step_flat(opt.flat_p[sl], opt.flat_g[sl], opt.exp_avg[sl], opt.exp_avg_sq[sl], opt.state,
                                   float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                                   float(scale), p16=None if p16 is None else p16[sl],
                                   p3=None if p3 is None else (p3.data_ptr() + 2 * b["begin"], opt.numel))
        if early:
            self.launched += 1

    def finish(self):
        """After the backward: flush the buckets that did not report, then make the main stream wait for the updates."""
        tr = self.tr
        main = torch.cuda.current_stream(tr.device)
        for b in self.buckets:
            if not b["done"]:
                self._launch(b, main, early=False, active=self.active)
        opt = tr.optim_gen
        if getattr(opt, "flat_p3t", None) is not None:      # every bucket's planes are written: the transposed weight copy follows
            from . import ops
            with torch.cuda.stream(tr.comm_stream):
                ops.x3_transpose_planes(opt.flat_p3, opt.flat_p3t, opt._x3t_table)
        main.wait_stream(tr.comm_stream)


def synthetic_batch(n, image_size, seed, device):
    """rand in [0,1) like dataset.py:65 (/255); A then B from one host generator."""
    g = torch.Generator().manual_seed(seed)
    A = torch.rand(n, 3, image_size, image_size, generator=g)
    B = torch.rand(n, 3, image_size, image_size, generator=g)
    return A.to(device), B.to(device)
