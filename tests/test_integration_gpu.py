"""The drop-in boundary exercised the way a maintainer of the reference would use it:

  * route A: the reference's loop body (image_translation.py:336-390) restated with ONLY ``model.Generator /
    Discriminator``, ``losses.*`` and ``optim.Adam`` of this package -- no trainer -- against the oracle fixture;
  * the criteria with the reference's call forms (label TENSORS, HingeEmbeddingLoss(x, ones));
  * the CLI: log-line format, checkpoint file names, ``--resume`` (bitwise continuation with graph replay on),
    ``--load_*`` checkpoints, uint8 image ingest;
  * the C-ABI exchange group on hardware (1-rank RCCL communicator), the bucketed G-step exchange, and
    ``python bench.py --gpus N`` starting its own ranks (gloo rehearsal on one GPU).
"""
import json
import os
import re
import subprocess
import sys
from itertools import chain

import pytest
import torch

pytestmark = pytest.mark.gpu

from discogan_modernized_amd import _lib, dp, losses, model, ops, optim  # noqa: E402
from discogan_modernized_amd import distributed_image_translation as dit  # noqa: E402
from discogan_modernized_amd import image_translation as it_cli  # noqa: E402
from discogan_modernized_amd.trainer import DiscoGANTrainer, default_args, synthetic_batch  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
LOG_RE = re.compile(r"^Iter \[(\d+)/(\d+)\] GEN: (-?\d+\.\d{4})/(-?\d+\.\d{4}), FM: (-?\d+\.\d{4})/(-?\d+\.\d{4}), "
                    r"RECON: (-?\d+\.\d{4}|nan)/(-?\d+\.\d{4}|nan), DIS: (-?\d+\.\d{4})/(-?\d+\.\d{4})$")


def test_route_a_reference_loop_body_on_package_modules():
    """image_translation.py:260-287,336-390 with the imports swapped for this package's (INTEGRATION.md route A)."""
    S, N = 16, 4
    fix = json.load(open(os.path.join(GOLD, "oracle_s16_n4.json")))
    device = torch.device(DEV)
    torch.manual_seed(1234)
    generator_A = model.Generator(extra_layers=True, image_size=S).to(device)
    generator_B = model.Generator(extra_layers=True, image_size=S).to(device)
    discriminator_A = model.Discriminator(image_size=S).to(device)
    discriminator_B = model.Discriminator(image_size=S).to(device)
    recon_criterion = losses.MSELoss()
    gan_criterion = losses.BCELoss()
    feat_criterion = losses.HingeEmbeddingLoss()
    optim_gen = optim.Adam(chain(generator_A.parameters(), generator_B.parameters()), lr=0.0002, betas=(0.5, 0.999),
                           weight_decay=0.00001)
    optim_dis = optim.Adam(chain(discriminator_A.parameters(), discriminator_B.parameters()), lr=0.0002, betas=(0.5, 0.999),
                           weight_decay=0.00001)
    A, B = synthetic_batch(N, S, 0, device)
    for iters, rec in enumerate(fix["iters"]):
        generator_A.zero_grad()
        generator_B.zero_grad()
        discriminator_A.zero_grad()
        discriminator_B.zero_grad()
        AB = generator_B(A)
        BA = generator_A(B)
        ABA = generator_A(AB)
        BAB = generator_B(BA)
        recon_loss_A = recon_criterion(ABA, A)
        recon_loss_B = recon_criterion(BAB, B)
        A_dis_real, A_feats_real = discriminator_A(A)
        A_dis_fake, A_feats_fake = discriminator_A(BA)
        dis_loss_A, gen_loss_A = losses.get_gan_loss(A_dis_real, A_dis_fake, gan_criterion, device)
        fm_loss_A = losses.get_fm_loss(A_feats_real, A_feats_fake, feat_criterion, device)
        B_dis_real, B_feats_real = discriminator_B(B)
        B_dis_fake, B_feats_fake = discriminator_B(AB)
        dis_loss_B, gen_loss_B = losses.get_gan_loss(B_dis_real, B_dis_fake, gan_criterion, device)
        fm_loss_B = losses.get_fm_loss(B_feats_real, B_feats_fake, feat_criterion, device)
        rate = 0.01 if iters < 10000 else 0.5
        gen_loss_A_total = (fm_loss_B * 0.9 + gen_loss_B * 0.1) * (1. - rate) + recon_loss_A * rate
        gen_loss_B_total = (fm_loss_A * 0.9 + gen_loss_A * 0.1) * (1. - rate) + recon_loss_B * rate
        gen_loss = gen_loss_A_total + gen_loss_B_total
        dis_loss = dis_loss_A + dis_loss_B
        if iters % 3 == 0:
            dis_loss.backward()
            optim_dis.step()
        else:
            gen_loss.backward()
            optim_gen.step()
        got = dict(gen_loss_A=gen_loss_A, gen_loss_B=gen_loss_B, fm_loss_A=fm_loss_A, fm_loss_B=fm_loss_B,
                   recon_loss_A=recon_loss_A, recon_loss_B=recon_loss_B, dis_loss_A=dis_loss_A, dis_loss_B=dis_loss_B,
                   gen_loss=gen_loss, dis_loss=dis_loss)
        strict = iters == 0
        for k, v in rec["losses"].items():
            g = float(got[k])
            if strict or not k.startswith(("gen_loss", "dis_loss")):
                assert abs(g - v) <= (1e-4 if strict else 0.15) * abs(v) + 1e-6, f"iter {iters} {k}: {g} vs {v}"
        if strict:                      # the first D-step's gradients and Adam update, against the fixture
            for name, net in (("dis_A", discriminator_A), ("dis_B", discriminator_B)):
                for pn, p in net.named_parameters():
                    ref_norm = rec["grad_norms"][name][pn]
                    assert abs(float(p.grad.double().norm()) - ref_norm) <= 2e-3 * ref_norm + 1e-9, f"grad norm {name}.{pn}"


def test_criteria_with_the_reference_call_forms():
    """nn.BCELoss(input, label TENSOR) (image_translation.py:157-166) and nn.HingeEmbeddingLoss(x, +-1 targets)."""
    g = torch.Generator().manual_seed(5)
    p = torch.rand(37, 1, generator=g).clamp(1e-4, 1 - 1e-4)
    p[3], p[5] = 1.0, 0.0                                      # the -100 clamp and the 1e-12 backward guard
    for t in (torch.ones(37, 1), torch.zeros(37, 1), torch.rand(37, 1, generator=g)):
        pc = p.clone().requires_grad_(True)
        ref = torch.nn.BCELoss()(pc, t)
        ref.backward()
        pg = p.clone().to(DEV).requires_grad_(True)
        got = losses.BCELoss()(pg, t.to(DEV))
        got.backward()
        assert abs(float(got) - float(ref)) <= 1e-6 * abs(float(ref)) + 1e-7
        assert torch.allclose(pg.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-9)
    x = torch.randn(5, 64, 8, 8, generator=g)
    y = torch.where(torch.rand(x.shape, generator=g) < 0.5, torch.ones(()), -torch.ones(()))
    for tgt in (torch.ones_like(x), y):
        xc = x.clone().requires_grad_(True)
        ref = torch.nn.HingeEmbeddingLoss()(xc, tgt)
        ref.backward()
        xg = x.clone().to(DEV).requires_grad_(True)
        got = losses.HingeEmbeddingLoss()(xg, tgt.to(DEV))
        got.backward()
        assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-7
        assert torch.allclose(xg.grad.cpu(), xc.grad, rtol=1e-6, atol=1e-12)
    # get_fm_loss == the reference expression through HingeEmbeddingLoss(l2, ones)
    r, f = torch.rand(4, 32, 4, 4, generator=g), torch.rand(4, 32, 4, 4, generator=g)
    l2 = (r.mean(0) - f.mean(0)) ** 2
    ref = torch.nn.HingeEmbeddingLoss()(l2, torch.ones_like(l2))
    got = losses.get_fm_loss([r.to(DEV)], [f.to(DEV)], losses.HingeEmbeddingLoss(), DEV)
    assert abs(float(got) - float(ref)) <= 1e-5 * float(ref)


def test_u8_ingest_matches_dataset_normalisation():
    """dataset.py:65-66: image.astype(np.float32) / 255., transpose(2, 0, 1)."""
    g = torch.Generator().manual_seed(1)
    img = torch.randint(0, 256, (5, 16, 24, 3), generator=g, dtype=torch.uint8)
    ref = (img.numpy().astype("float32") / 255.).transpose(0, 3, 1, 2)
    got = ops.u8hwc_to_f32chw(img.to(DEV)).cpu().numpy()
    assert got.shape == ref.shape and (got == ref).all()
    got = ops.u8hwc_to_f32chw(img.to(DEV), bgr=True).cpu().numpy()
    assert (got == ref[:, ::-1]).all()


@pytest.mark.parametrize("S,N", [(16, 5), (64, 3), (128, 1)])
def test_folded_inference_generator_matches_oracle_eval(S, N):
    """inference.py:149: generator.eval().  BatchNorm folded into the convolutions (scale into the weights, shift +
    activation in the conv epilogue) against the oracle's eval-mode forward, with running statistics moved away from
    their initial (0, 1) and non-trivial gamma / beta."""
    from oracle import discogan_ref as O
    from discogan_modernized_amd.inference import FoldedGenerator
    torch.manual_seed(11)
    og = O.Generator(True, image_size=S)
    og.train()
    with torch.no_grad():
        for _ in range(3):
            og(torch.rand(4, 3, S, S))
        for m in og.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    og.eval()
    mg = model.Generator(True, image_size=S).to(DEV)
    mg.load_state_dict(og.state_dict())
    mg.eval()
    x = torch.rand(N, 3, S, S)
    with torch.no_grad():
        ref = og(x)
        plain = mg(x.to(DEV))
    folded = FoldedGenerator(mg)(x.to(DEV))
    assert folded.shape == ref.shape
    for got, what in ((plain, "eval modules"), (folded, "folded")):
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 1e-4 * ref.abs().max().item() + 1e-5, f"{what}: max err {err:.2e}"


def test_inference_cli_roundtrip(tmp_path):
    """`python -m discogan_modernized_amd.inference`: checkpoints written by the training CLI, AtoB uses gen_B_final.pth
    and reconstructs with gen_A_final.pth (inference.py:127-136,171-187); uint8 image-row input."""
    from discogan_modernized_amd import inference as inf
    argv = ["--task_name", "edges2shoes", "--image_size", "16", "--batch_size", "4", "--synthetic_size", "8", "--epochs", "2",
            "--results_dir", str(tmp_path / "r"), "--models_dir", str(tmp_path / "m")]
    it_cli.main(argv)
    _, mp = it_cli.train.last_paths
    g = torch.Generator().manual_seed(4)
    torch.save(torch.randint(0, 256, (3, 16, 16, 3), generator=g, dtype=torch.uint8), tmp_path / "imgs.pt")
    res = inf.main(["--model_path", str(mp), "--input_path", str(tmp_path / "imgs.pt"), "--image_size", "16",
                    "--output_dir", str(tmp_path / "out"), "--direction", "AtoB"])
    stem, gen, rec = res[0]
    assert gen.shape == (3, 3, 16, 16) and rec.shape == (3, 3, 16, 16) and float(gen.min()) > 0 and float(gen.max()) < 1
    saved = torch.load(tmp_path / "out" / "imgs_result.pt")
    assert torch.equal(saved["generated"], gen.cpu())
    # the folded path == the training modules in eval mode
    res2 = inf.main(["--model_path", str(mp), "--input_path", str(tmp_path / "imgs.pt"), "--image_size", "16",
                     "--output_dir", str(tmp_path / "out2"), "--direction", "AtoB", "--no_fold"])
    assert torch.allclose(res2[0][1], gen, rtol=1e-4, atol=1e-6) and torch.allclose(res2[0][2], rec, rtol=1e-4, atol=1e-6)


def _cli(tmp, tag, extra):
    argv = ["--task_name", "edges2shoes", "--image_size", "16", "--batch_size", "4", "--synthetic_size", "16", "--epochs", "3",
            "--log_interval", "1", "--model_save_interval", "5", "--save_train_state",
            "--results_dir", str(tmp / f"res_{tag}"), "--models_dir", str(tmp / f"mod_{tag}")] + extra
    it_cli.main(argv)
    rp, mp = it_cli.train.last_paths
    return rp, mp


def test_cli_log_line_checkpoints_and_exact_resume(tmp_path, capsys):
    """`python -m discogan_modernized_amd.image_translation ...` for 12 iterations (3 epochs x 4 batches): the log file
    (format image_translation.py:394-398), the checkpoint file names (:420-432), and --resume from the periodic
    train_state written AFTER iteration 5 continuing bitwise (iteration, Adam state, data order, eager re-warm-up
    before graph capture)."""
    rp, mp = _cli(tmp_path, "full", [])
    lines = [l for l in open(rp / "training_log.txt").read().splitlines() if l.startswith("Iter")]
    assert len(lines) == 12
    for i, l in enumerate(lines):
        m = LOG_RE.match(l)
        assert m and int(m.group(1)) == i and int(m.group(2)) == 12, l
    names = sorted(os.listdir(mp))
    for tag in ("0", "5", "10", "final"):
        for net in ("gen_A", "gen_B", "dis_A", "dis_B"):
            assert f"{net}_{tag}.pth" in names
        assert f"train_state_{tag}.pth" in names
    sd = torch.load(mp / "gen_B_final.pth")
    assert list(sd.keys())[:3] == ["encoder.0.weight", "encoder.2.weight", "encoder.3.weight"]   # Appendix B key layout
    st = torch.load(mp / "train_state_5.pth")
    assert st["iters"] == 6 and st["loader"] == dict(epoch=1, batch=2)
    capsys.readouterr()
    rp2, mp2 = _cli(tmp_path, "resumed", ["--resume", str(mp / "train_state_5.pth")])
    lines2 = [l for l in open(rp2 / "training_log.txt").read().splitlines() if l.startswith("Iter")]
    assert lines2 == lines[6:], "resumed run must print the same log lines as the uninterrupted one"
    for net in ("gen_A", "gen_B", "dis_A", "dis_B"):
        a, b = torch.load(mp / f"{net}_final.pth"), torch.load(mp2 / f"{net}_final.pth")
        for k in a:
            assert torch.equal(a[k], b[k]), f"{net}.{k} differs after resume"
    # --load_* (distributed_image_translation.py:117-124,379-393): weights only, Adam restarts
    args = dit.parse_args(["--image_size", "16", "--load_gen_A", str(mp / "gen_A_final.pth"), "--load_dis_B", str(mp / "dis_B_final.pth")])
    tr = DiscoGANTrainer(args, device=DEV, image_size=16, seed=99)
    dit.load_checkpoints(args, tr)
    ref = torch.load(mp / "gen_A_final.pth")
    for k, v in tr.generator_A.state_dict().items():
        assert torch.equal(v.cpu(), ref[k]), k
    assert float(tr.optim_gen.state[0]) == 0.0


def test_cli_u8_data_files_and_lazy_flag(tmp_path):
    g = torch.Generator().manual_seed(2)
    for d in "AB":
        torch.save(torch.randint(0, 256, (8, 16, 16, 3), generator=g, dtype=torch.uint8), tmp_path / f"{d}.pt")
    argv = ["--task_name", "edges2shoes", "--image_size", "16", "--batch_size", "4", "--epochs", "2", "--log_interval", "2",
            "--data_A", str(tmp_path / "A.pt"), "--data_B", str(tmp_path / "B.pt"), "--skip_log_only_passes",
            "--results_dir", str(tmp_path / "r"), "--models_dir", str(tmp_path / "m")]
    it_cli.main(argv)
    rp, _ = it_cli.train.last_paths
    lines = [l for l in open(rp / "training_log.txt").read().splitlines() if l.startswith("Iter")]
    assert len(lines) == 2 and all(LOG_RE.match(l) for l in lines)


def test_exchange_group_capi_one_rank_rccl_communicator():
    """The C-ABI comm group on hardware: dlopen RCCL, unique id, ncclCommInitRank(1 rank), all-reduce / broadcast /
    barrier on the caller's stream, destroy.  (More ranks need more GPUs; the multi-rank arithmetic is RCCL's.)"""
    L = _lib.load()
    assert L.dg_dp_world_size() == 0 and L.dg_dp_rank() == -1
    x = torch.arange(1000, device=DEV, dtype=torch.float32)
    assert L.dg_dp_allreduce_sum(x.data_ptr(), x.numel(), None) != 0 and b"no communicator" in L.dg_last_error()
    xg = dp.ExchangeGroup(None, transport="capi", device=torch.device(DEV))
    try:
        assert L.dg_dp_world_size() == 1 and L.dg_dp_rank() == 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # collectives ride on the caller's stream
            scale = xg.all_reduce_sum_(x)
            xg.broadcast_(x, 0)
        torch.cuda.current_stream().wait_stream(side)
        xg.barrier()
        assert scale == 1.0 and torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float32))
        with pytest.raises(_lib.DiscoganHipError, match="already exists"):
            dp.ExchangeGroup(None, transport="capi", device=torch.device(DEV))
    finally:
        xg.close()
    assert L.dg_dp_world_size() == 0


@pytest.mark.parametrize("arch", ["discogan", "recongan", "gan"])
def test_bucketed_gstep_exchange_is_bitwise_neutral(arch):
    """overlap_comm=True: D-step exchange + Adam on the communication stream, G-step exchange per gradient bucket as
    soon as the bucket's last backward kernel is queued, each followed by its Adam slice.  With a 1-rank RCCL
    communicator the collectives move nothing, so every value must equal the plain path's bit for bit -- this pins
    the bucket boundaries, event ordering, the once-per-step Adam state advance and the flush of buckets whose
    network is outside the loss (recongan / gan)."""
    A, B = synthetic_batch(4, 16, 0, DEV)
    res = []
    for overlap in (False, True):
        tr = DiscoGANTrainer(default_args(model_arch=arch), device=DEV, image_size=16, seed=1234, overlap_comm=overlap,
                             comm="capi", bucket_mb=0.05)
        try:
            if overlap:
                assert len(tr._buckets.buckets) >= 6 and not tr.use_graph
            vals = [tr.losses_to_floats(tr.train_iteration(A, B, it)) for it in range(9)]
            tr.finish()
            torch.cuda.synchronize()
            if overlap:
                assert tr._buckets.launched > 0, "no bucket was exchanged before the end of the backward pass"
            res.append((vals, tr.optim_gen.flat_p.clone(), tr.optim_dis.flat_p.clone(), tr.optim_gen.exp_avg_sq.clone()))
        finally:
            tr.close()
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)


def test_bucketed_exchange_keeps_the_f32x3_weight_planes_current():
    """mfma_dtype="f32x3" with plane operands under overlap_comm: every bucket's Adam slice also writes its slice of the weight
    planes, and the transposed copy is rewritten once all buckets are in -- weights, planes and transposed planes equal the
    plain path's bit for bit, and the planes ARE the weights (hi + mid + lo)."""
    A, B = synthetic_batch(4, 64, 0, DEV)
    res = []
    for overlap in (False, True):
        tr = DiscoGANTrainer(default_args(), device=DEV, image_size=64, seed=1234, overlap_comm=overlap, comm="capi", bucket_mb=0.5,
                             mfma_dtype="f32x3", x3_planes=True)
        try:
            vals = [tr.losses_to_floats(tr.train_iteration(A, B, it)) for it in range(6)]
            tr.finish()
            torch.cuda.synchronize()
            if overlap:
                assert len(tr._buckets.buckets) >= 4 and tr._buckets.launched > 0
            res.append((vals, tr.optim_gen.flat_p.clone(), tr.optim_gen.flat_p3.clone(), tr.optim_gen.flat_p3t.clone(),
                        tr.optim_dis.flat_p3.clone(), tr.optim_dis.flat_p3t.clone()))
        finally:
            tr.close()
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)
    assert torch.equal(res[1][2].float().sum(0), res[1][1])


def test_refresh_derived_after_an_outside_write_to_the_flat_parameters():
    """optim.Adam.refresh_derived: the bf16 shadow / the f32x3 plane triples and their transposed copy follow a write to flat_p
    that went around the parameters (what ExchangeGroup.broadcast_(opt.flat_p) does)."""
    g = model.Generator(extra_layers=True, image_size=16).to(DEV)
    opt = optim.Adam(g.parameters(), lr=2e-4)
    opt.enable_bf16_shadow()
    opt.enable_x3_planes()
    opt.flat_p.mul_(1.5).add_(0.01)
    opt.refresh_derived()
    assert torch.equal(opt.flat_p16, opt.flat_p.bfloat16())
    assert torch.equal(opt.flat_p3.float().sum(0), opt.flat_p)
    for p_, off in zip(opt.params, opt.offsets):
        if p_._dg_x3[2] is not None:
            k, j = p_.shape[0], 16 * p_.shape[1]
            src = opt.flat_p3[:, off:off + k * j].view(3, k, j).transpose(1, 2).reshape(3, -1)
            assert torch.equal(opt.flat_p3t[:, off:off + k * j], src)


def test_lazy_d_steps_change_only_generator_bn_buffers():
    """need_losses=False (--skip_log_only_passes): weights, optimiser state and D buffers are bitwise the full run's;
    the generators' BatchNorm running statistics are NOT (one forward per D-step instead of two) -- documented."""
    A, B = synthetic_batch(4, 16, 0, DEV)
    runs = []
    for lazy in (False, True):
        tr = DiscoGANTrainer(default_args(), device=DEV, image_size=16, seed=1234)
        for it in range(6):
            tr.train_iteration(A, B, it, need_losses=not lazy)
        torch.cuda.synchronize()
        runs.append(tr)
    full, lz = runs
    assert torch.equal(full.optim_gen.flat_p, lz.optim_gen.flat_p) and torch.equal(full.optim_dis.flat_p, lz.optim_dis.flat_p)
    for (n, a), (_, b) in zip(full.discriminator_A.named_buffers(), lz.discriminator_A.named_buffers()):
        assert torch.equal(a, b), n
    nbt_full = int(full.generator_A.encoder[3].num_batches_tracked)
    nbt_lazy = int(lz.generator_A.encoder[3].num_batches_tracked)
    assert nbt_full == 12 and nbt_lazy == 10        # 2 D-steps skipped one generator call each


@pytest.mark.timeout(600)
def test_bench_starts_its_own_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2` as typed (no torchrun): the parent spawns the ranks before touching the GPU and relays
    rank 0's JSON line.  Rehearsed with gloo because two ranks share this box's one GPU."""
    env = dict(os.environ, DG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--image_size", "64", "--batch_size", "8",
                        "--steps", "6", "--warmup", "3", "--no_extra", "--no_cpu_baseline", "--no_roofline"],
                       env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["steps"] == 6 and d["value"] > 0
    assert d["comm"]["world_size"] == 2 and d["comm"]["allreduce_ms_per_D_step"] > 0 and d["comm"]["allreduce_ms_per_G_step"] > 0
    assert d["config"]["hipgraph"] is True and d["comm"]["allreduce_overlap"] is False   # 64 px: graph replay + exchange behind it
    # exposed exchange time: the same window with the collectives stubbed, next to the per-collective event sums
    assert d["comm"]["ms_per_step_exchange_stubbed"] > 0 and d["comm"]["exposed_ms_per_step"] is not None
    assert abs(d["comm"]["exposed_ms_per_step"] - (d["ms_per_step"] - d["comm"]["ms_per_step_exchange_stubbed"])) < 2e-3
    assert d["dtype"] == "f32x3" and "three bf16 planes" in d["config"]["workload"]
