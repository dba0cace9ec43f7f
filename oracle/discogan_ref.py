"""CPU ORACLE for the DiscoGAN training-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the reference algorithm
(fasion-image-generator-project/discogan_modernized).  It is the *checker* for the HIP
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``discogan_modernized_amd/`` imports it, and the product path
raises when the HIP library is missing instead of falling back to this file.

Parity pin: ``tests/golden/ref_s512_n2.json`` holds outputs of the TRUE reference
(``/root/reference/model.py`` + ``image_translation.get_gan_loss/get_fm_loss`` driven through
the loop body ``image_translation.py:336-390``), produced by ``tests/golden/make_golden.py``.
``tests/test_oracle_golden.py`` checks this restatement against that fixture (seeded weights
bit-equal, 3 iterations of losses / grad norms / Adam-updated samples).

What is restated (reference file:line):
  * ``Discriminator``            model.py:5-69
  * ``Generator``                model.py:72-225  (extra_layers True/False are structurally identical)
  * ``get_fm_loss``              image_translation.py:136-144
  * ``get_gan_loss``             image_translation.py:146-168
  * criteria / optimisers        image_translation.py:267-287
  * one training iteration       image_translation.py:336-390
  * DDP semantics (intended)     distributed_image_translation.py:401-404,513-518
    (per-rank local BN/FM statistics, gradients averaged over ranks, identical Adam step)

Depth rule (build-defined generalisation): ``n = log2(S) - 2`` stride-2
stages, channels ``min(64 * 2**(i-1), 2048)``.  At S=512 this is the reference network
weight-for-weight (same state_dict keys, shapes and RNG consumption order); at S=64 it is the
original DiscoGAN 64 px network.  The reference itself only runs at S=512 (model.py:8-35).
"""
from __future__ import annotations

import math
from itertools import chain
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.optim as optim


# --------------------------------------------------------------------------------------
# network definition
# --------------------------------------------------------------------------------------
def stage_channels(image_size: int):
    """Channels after each stride-2 stage (model.py:8-31 at S=512: 64..2048,2048)."""
    n = int(round(math.log2(image_size))) - 2
    if 2 ** (n + 2) != image_size or n < 1:
        raise ValueError(f"image_size must be a power of two >= 8, got {image_size}")
    return [min(64 * 2 ** i, 2048) for i in range(n)]


class Discriminator(nn.Module):
    """model.py:5-69.  conv1..conv{n} k4 s2 p1, BN on 2..n, LeakyReLU(0.2), head conv k4 s1 p0 -> 1."""

    def __init__(self, image_size: int = 512):
        super().__init__()
        ch = stage_channels(image_size)
        self.n_stages = len(ch)
        cin = 3
        for i, c in enumerate(ch, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, c, 4, 2, 1, bias=False))   # model.py:8,11,...
            if i >= 2:
                setattr(self, f"bn{i}", nn.BatchNorm2d(c))                      # model.py:12,...
            setattr(self, f"relu{i}", nn.LeakyReLU(0.2, inplace=True))          # model.py:9,13,...
            cin = c
        setattr(self, f"conv{len(ch) + 1}", nn.Conv2d(cin, 1, 4, 1, 0, bias=False))  # model.py:35
        self.sigmoid = nn.Sigmoid()                                              # model.py:36

    def forward(self, x):
        feats = []
        h = x
        for i in range(1, self.n_stages + 1):
            h = getattr(self, f"conv{i}")(h)
            if i >= 2:
                h = getattr(self, f"bn{i}")(h)
            h = getattr(self, f"relu{i}")(h)
            if i >= 2:
                feats.append(h)                                                  # model.py:69
        out = self.sigmoid(getattr(self, f"conv{self.n_stages + 1}")(h))
        return out, feats


class Generator(nn.Module):
    """model.py:72-225.  encoder/decoder nn.Sequential with the reference's index layout."""

    def __init__(self, extra_layers: bool = False, image_size: int = 512):
        super().__init__()
        ch = stage_channels(image_size)
        enc = []
        cin = 3
        for i, c in enumerate(ch):
            enc.append(nn.Conv2d(cin, c, 4, 2, 1, bias=False))                   # model.py:80-103
            if i >= 1:
                enc.append(nn.BatchNorm2d(c))
            enc.append(nn.LeakyReLU(0.2, inplace=True))
            cin = c
        enc += [nn.Conv2d(cin, 100, 4, 1, 0, bias=False), nn.BatchNorm2d(100),
                nn.LeakyReLU(0.2, inplace=True)]                                 # model.py:107-109
        self.encoder = nn.Sequential(*enc)
        dec = [nn.ConvTranspose2d(100, ch[-1], 4, 1, 0, bias=False), nn.BatchNorm2d(ch[-1]),
               nn.ReLU(True)]                                                     # model.py:114-116
        for i in range(len(ch) - 1, 0, -1):
            dec += [nn.ConvTranspose2d(ch[i], ch[i - 1], 4, 2, 1, bias=False),
                    nn.BatchNorm2d(ch[i - 1]), nn.ReLU(True)]                     # model.py:118-140
        dec += [nn.ConvTranspose2d(ch[0], 3, 4, 2, 1, bias=False), nn.Sigmoid()]  # model.py:142-143
        self.decoder = nn.Sequential(*dec)
        self.main = None                                                          # model.py:215

    def forward(self, x):
        return self.decoder(self.encoder(x))                                     # model.py:217-225


# --------------------------------------------------------------------------------------
# losses (image_translation.py:136-168)
# --------------------------------------------------------------------------------------
def get_fm_loss(real_feats, fake_feats, criterion, device="cpu"):
    losses = 0
    for real_feat, fake_feat in zip(real_feats, fake_feats):
        l2 = (real_feat.mean(0) - fake_feat.mean(0)) * (real_feat.mean(0) - fake_feat.mean(0))
        loss = criterion(l2, torch.ones(l2.size()).to(device))
        losses += loss
    return losses


def get_gan_loss(dis_real, dis_fake, criterion, device="cpu"):
    batch_size = dis_real.size(0)
    if len(dis_real.size()) > 2:
        dis_real = dis_real.view(batch_size, -1)
    if len(dis_fake.size()) > 2:
        dis_fake = dis_fake.view(batch_size, -1)
    labels_dis_real = torch.ones(batch_size, 1).to(device)
    labels_dis_fake = torch.zeros(batch_size, 1).to(device)
    labels_gen = torch.ones(batch_size, 1).to(device)
    dis_loss = (criterion(dis_real, labels_dis_real) + criterion(dis_fake, labels_dis_fake)) * 0.5
    gen_loss = criterion(dis_fake, labels_gen)
    return dis_loss, gen_loss


# --------------------------------------------------------------------------------------
# training state and one iteration (image_translation.py:260-287, 336-390)
# --------------------------------------------------------------------------------------
DEFAULTS = dict(learning_rate=2e-4, beta1=0.5, beta2=0.999, weight_decay=0.00001,
                gan_curriculum=10000, starting_rate=0.01, default_rate=0.5,
                update_interval=3, model_arch="discogan")


def default_args(**over):
    d = dict(DEFAULTS)
    d.update(over)
    return SimpleNamespace(**d)


def build_state(image_size=512, seed=1234, args=None, gen_cls=Generator, dis_cls=Discriminator,
                cls_kwargs=None):
    """Seeded construction in the DDP script's order G_A, G_B, D_A, D_B
    (distributed_image_translation.py:372-376) + the two Adam optimisers
    (image_translation.py:272-287)."""
    args = args or default_args()
    if cls_kwargs is None:
        cls_kwargs = dict(image_size=image_size)
    if seed is not None:
        torch.manual_seed(seed)
    st = SimpleNamespace()
    st.generator_A = gen_cls(extra_layers=True, **cls_kwargs)
    st.generator_B = gen_cls(extra_layers=True, **cls_kwargs)
    st.discriminator_A = dis_cls(**cls_kwargs)
    st.discriminator_B = dis_cls(**cls_kwargs)
    st.recon_criterion = nn.MSELoss()
    st.gan_criterion = nn.BCELoss()
    st.feat_criterion = nn.HingeEmbeddingLoss()
    st.optim_gen = optim.Adam(chain(st.generator_A.parameters(), st.generator_B.parameters()),
                              lr=args.learning_rate, betas=(args.beta1, args.beta2),
                              weight_decay=args.weight_decay)
    st.optim_dis = optim.Adam(chain(st.discriminator_A.parameters(), st.discriminator_B.parameters()),
                              lr=args.learning_rate, betas=(args.beta1, args.beta2),
                              weight_decay=args.weight_decay)
    st.args = args
    st.nets = dict(gen_A=st.generator_A, gen_B=st.generator_B,
                   dis_A=st.discriminator_A, dis_B=st.discriminator_B)
    return st


def forward_losses(st, A, B, iters, fm_fn=get_fm_loss, gan_fn=get_gan_loss, device="cpu"):
    """Loop body image_translation.py:342-382 (forward of all four nets + loss mix)."""
    args = st.args
    AB = st.generator_B(A)
    BA = st.generator_A(B)
    ABA = st.generator_A(AB)
    BAB = st.generator_B(BA)
    recon_loss_A = st.recon_criterion(ABA, A)
    recon_loss_B = st.recon_criterion(BAB, B)
    A_dis_real, A_feats_real = st.discriminator_A(A)
    A_dis_fake, A_feats_fake = st.discriminator_A(BA)
    dis_loss_A, gen_loss_A = gan_fn(A_dis_real, A_dis_fake, st.gan_criterion, device)
    fm_loss_A = fm_fn(A_feats_real, A_feats_fake, st.feat_criterion, device)
    B_dis_real, B_feats_real = st.discriminator_B(B)
    B_dis_fake, B_feats_fake = st.discriminator_B(AB)
    dis_loss_B, gen_loss_B = gan_fn(B_dis_real, B_dis_fake, st.gan_criterion, device)
    fm_loss_B = fm_fn(B_feats_real, B_feats_fake, st.feat_criterion, device)
    rate = args.starting_rate if iters < args.gan_curriculum else args.default_rate
    gen_loss_A_total = (fm_loss_B * 0.9 + gen_loss_B * 0.1) * (1 - rate) + recon_loss_A * rate
    gen_loss_B_total = (fm_loss_A * 0.9 + gen_loss_A * 0.1) * (1 - rate) + recon_loss_B * rate
    if args.model_arch == "discogan":
        gen_loss = gen_loss_A_total + gen_loss_B_total
        dis_loss = dis_loss_A + dis_loss_B
    elif args.model_arch == "recongan":
        gen_loss = gen_loss_A_total
        dis_loss = dis_loss_B
    elif args.model_arch == "gan":
        gen_loss = gen_loss_B * 0.1 + fm_loss_B * 0.9
        dis_loss = dis_loss_B
    else:
        raise ValueError(args.model_arch)
    return SimpleNamespace(
        gen_loss=gen_loss, dis_loss=dis_loss,
        gen_loss_A=gen_loss_A, gen_loss_B=gen_loss_B, fm_loss_A=fm_loss_A, fm_loss_B=fm_loss_B,
        recon_loss_A=recon_loss_A, recon_loss_B=recon_loss_B,
        dis_loss_A=dis_loss_A, dis_loss_B=dis_loss_B,
        AB=AB, BA=BA, ABA=ABA, BAB=BAB,
        A_dis_real=A_dis_real, A_dis_fake=A_dis_fake, B_dis_real=B_dis_real, B_dis_fake=B_dis_fake)


LOG_KEYS = ("gen_loss_A", "gen_loss_B", "fm_loss_A", "fm_loss_B", "recon_loss_A", "recon_loss_B",
            "dis_loss_A", "dis_loss_B", "gen_loss", "dis_loss")


def is_dis_step(iters, args):
    return iters % args.update_interval == 0                                     # :385


def train_iteration(st, A, B, iters, do_step=True):
    """One full reference iteration: zero_grad, 8 passes, losses, backward of the selected side,
    Adam step of that side (image_translation.py:336-390).  Returns the loss namespace."""
    for net in st.nets.values():
        net.zero_grad()                                                          # :336-339
    out = forward_losses(st, A, B, iters)
    if is_dis_step(iters, st.args):
        out.dis_loss.backward()
        if do_step:
            st.optim_dis.step()
    else:
        out.gen_loss.backward()
        if do_step:
            st.optim_gen.step()
    return out


def losses_to_floats(out):
    return {k: float(getattr(out, k).detach()) for k in LOG_KEYS}


def format_log(iters, total, out):
    """Log line format image_translation.py:394-398."""
    f = losses_to_floats(out)
    return (f"Iter [{iters}/{total}] "
            f"GEN: {f['gen_loss_A']:.4f}/{f['gen_loss_B']:.4f}, "
            f"FM: {f['fm_loss_A']:.4f}/{f['fm_loss_B']:.4f}, "
            f"RECON: {f['recon_loss_A']:.4f}/{f['recon_loss_B']:.4f}, "
            f"DIS: {f['dis_loss_A']:.4f}/{f['dis_loss_B']:.4f}")


# --------------------------------------------------------------------------------------
# data-parallel semantics (intended behaviour of distributed_image_translation.py:401-404,513-518)
# --------------------------------------------------------------------------------------
def dp_emulated_iteration(st, A_shards, B_shards, iters):
    """Single-process emulation of a W-rank DDP iteration: every rank runs forward/backward on its
    own shard with rank-local BatchNorm / feature-matching statistics, gradients are summed and
    divided by W, then one identical Adam step.  BN running buffers follow rank 0 (only rank 0
    saves, distributed_image_translation.py:552-568).  Returns rank-0 losses."""
    W = len(A_shards)
    live = ("dis_A", "dis_B") if is_dis_step(iters, st.args) else ("gen_A", "gen_B")
    params = [p for k in live for p in st.nets[k].parameters()]
    acc = [torch.zeros_like(p) for p in params]
    buf0 = None
    out0 = None
    saved = {k: {n: b.clone() for n, b in net.named_buffers()} for k, net in st.nets.items()}
    for r in range(W):
        for k, net in st.nets.items():                       # every rank starts from the same buffers
            for n, b in net.named_buffers():
                b.copy_(saved[k][n])
        out = train_iteration(st, A_shards[r], B_shards[r], iters, do_step=False)
        for a, p in zip(acc, params):
            a += p.grad
        if r == 0:
            out0 = out
            buf0 = {k: {n: b.clone() for n, b in net.named_buffers()} for k, net in st.nets.items()}
    for k, net in st.nets.items():
        for n, b in net.named_buffers():
            b.copy_(buf0[k][n])
    for a, p in zip(acc, params):
        p.grad = a / W
    (st.optim_dis if is_dis_step(iters, st.args) else st.optim_gen).step()
    return out0


def synthetic_batch(n, image_size, seed=0):
    """A then B from one generator: rand in [0,1) like dataset.py:65 (/255)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.rand(n, 3, image_size, image_size, generator=g)
    B = torch.rand(n, 3, image_size, image_size, generator=g)
    return A, B
