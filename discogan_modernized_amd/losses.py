"""Loss criteria and helpers with the reference's signatures (image_translation.py:136-168, 267-269).

``get_gan_loss`` / ``get_fm_loss`` keep the argument order ``(…, criterion, device)`` and the return
order ``(dis_loss, gen_loss)``.  Labels are constants (ones / zeros), so they are passed to the BCE
kernel as a scalar instead of being materialised on the host and copied every call
(image_translation.py:157-159 — pure overhead in the reference).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F


class MSELoss(nn.Module):
    """nn.MSELoss() (mean reduction)."""

    def forward(self, input, target):
        return F.MSELossFn.apply(input, target)


class BCELoss(nn.Module):
    """nn.BCELoss() (mean reduction, log clamped at -100).  ``target`` is a tensor of the input's size, as in the
    reference (image_translation.py:157-166), or a Python scalar label (1.0 / 0.0), which skips materialising it."""

    def forward(self, input, target):
        if isinstance(target, torch.Tensor):
            label = getattr(target, "_dg_label", None)
            if label is None:
                return F.BCETargetLossFn.apply(input, target.to(input.device))
        else:
            label = float(target)
        return F.BCELossFn.apply(input, label)


class HingeEmbeddingLoss(nn.Module):
    """nn.HingeEmbeddingLoss(margin=1.0) (mean reduction), targets in {+1, -1}.  The reference only calls it with
    all-ones targets (image_translation.py:141-142), where it is ``input.mean()``;
    ``get_fm_loss`` below fuses that whole layer term into one kernel and does not go through this module."""

    def __init__(self, margin: float = 1.0):
        super().__init__()
        self.margin = float(margin)

    def forward(self, input, target):
        return F.HingeEmbeddingLossFn.apply(input, target.to(input.device), self.margin)


def label_like(batch_size, value, device):
    t = torch.full((batch_size, 1), float(value), device=device)
    t._dg_label = float(value)
    return t


def get_fm_loss(real_feats, fake_feats, criterion=None, device=None):
    """image_translation.py:136-144: sum over layers of mean_{chw}((mean_n real - mean_n fake)^2)."""
    losses = 0
    for real_feat, fake_feat in zip(real_feats, fake_feats):
        losses = losses + F.FeatureMatchFn.apply(real_feat, fake_feat)
    return losses


def get_gan_loss(dis_real, dis_fake, criterion, device=None):
    """image_translation.py:146-168."""
    batch_size = dis_real.size(0)
    if len(dis_real.size()) > 2:
        dis_real = dis_real.view(batch_size, -1)
    if len(dis_fake.size()) > 2:
        dis_fake = dis_fake.view(batch_size, -1)
    dis_loss = (criterion(dis_real, 1.0) + criterion(dis_fake, 0.0)) * 0.5
    gen_loss = criterion(dis_fake, 1.0)
    return dis_loss, gen_loss
