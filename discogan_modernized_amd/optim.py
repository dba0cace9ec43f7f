"""Adam over ONE flat fp32 buffer per parameter group (optim.Adam, image_translation.py:272-287).

All parameters handed to ``Adam`` are moved (keeping each one's memory layout) into a single flat
buffer; ``.grad`` of every parameter is a view of a matching flat gradient buffer.  One optimiser step
is then two kernel launches (scalar advance + multi-tensor update), and data parallelism all-reduces
the flat gradient buffer as one message.  Step count and bias corrections live on the device so a
step can be captured in a hipGraph.
"""
from __future__ import annotations

import torch

from . import ops

_ALIGN = 64  # floats (256 B)


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("discogan_modernized_amd.optim.Adam needs parameters on a HIP device (no CPU fallback)")
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(params=self.params, **self.defaults)]
        offs, total = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("all parameters must be fp32 on one device")
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.flat_p = torch.zeros(total, device=dev)
        self.flat_g = torch.zeros(total, device=dev)
        self.exp_avg = torch.zeros(total, device=dev)
        self.exp_avg_sq = torch.zeros(total, device=dev)
        self.state = torch.zeros(4, device=dev, dtype=torch.float64)
        self.offsets = offs
        with torch.no_grad():
            for p, off in zip(self.params, offs):
                n = p.numel()
                view = self.flat_p[off:off + n].as_strided(p.shape, p.stride())
                view.copy_(p.data)
                p.data = view
                gview = self.flat_g[off:off + n].as_strided(p.shape, p.stride())
                p._dg_flat_grad = gview
                p.grad = gview

    def enable_bf16_shadow(self):
        """Keep a bf16 (RNE) shadow of every parameter for the bf16 matrix path: one flat bf16 buffer, refreshed by the
        Adam kernel itself (+2 B/param), exposed as ``param._dg_bf16`` views with the parameter's own memory layout."""
        if getattr(self, "flat_p16", None) is not None:
            return
        self.flat_p16 = torch.empty(self.numel, device=self.flat_p.device, dtype=torch.bfloat16)
        ops.f32_to_bf16(self.flat_p, self.flat_p16)
        for p, off in zip(self.params, self.offsets):
            p._dg_bf16 = self.flat_p16[off:off + p.numel()].as_strided(p.shape, p.stride())
            p._dg_bf16_ver = p._version

    def enable_x3_planes(self):
        """Keep the three bf16 planes (hi / mid / lo, ops.f32_to_bf16x3) of every parameter for the f32x3 matrix path: one
        flat [3, numel] bf16 buffer refreshed by the Adam kernel itself (+6 B/param), exposed as ``param._dg_x3`` =
        (buffer, element offset)."""
        if getattr(self, "flat_p3", None) is not None:
            return
        self.flat_p3 = torch.empty((3, self.numel), device=self.flat_p.device, dtype=torch.bfloat16)
        ops.f32_to_bf16x3(self.flat_p, self.flat_p3)
        # the forward form of the plane kernel wants the conv weights' planes TRANSPOSED ([(r, s, c)][k]): a second buffer,
        # same flat offsets, rewritten for all conv weights of the group by one launch behind every Adam step
        convs = [(off, p.shape[0], 16 * p.shape[1]) for p, off in zip(self.params, self.offsets)
                 if p.dim() == 4 and tuple(p.shape[2:]) == (4, 4) and p.shape[0] > 1 and ops.is_krsc(p)]
        self.flat_p3t = torch.zeros((3, self.numel), device=self.flat_p.device, dtype=torch.bfloat16) if convs else None
        self._x3t_table = ops.x3_transpose_table(convs)
        conv_offs = {c[0] for c in convs}
        for p, off in zip(self.params, self.offsets):
            p._dg_x3 = (self.flat_p3, off, self.flat_p3t if off in conv_offs else None)
            p._dg_x3_ver = p._version
        if convs:
            ops.x3_transpose_planes(self.flat_p3, self.flat_p3t, self._x3t_table)

    def refresh_derived(self):
        """Recompute everything derived from ``flat_p`` (bf16 shadow, f32x3 plane triples and their transposed copy) after the
        flat buffer was written from outside the optimiser -- e.g. ``ExchangeGroup.broadcast_(opt.flat_p)``; writes through the
        parameters themselves (``load_state_dict``) are noticed per weight by their version counters."""
        if getattr(self, "flat_p16", None) is not None:
            ops.f32_to_bf16(self.flat_p, self.flat_p16)
        if getattr(self, "flat_p3", None) is not None:
            ops.f32_to_bf16x3(self.flat_p, self.flat_p3)
            if self.flat_p3t is not None:
                ops.x3_transpose_planes(self.flat_p3, self.flat_p3t, self._x3t_table)

    # -- torch.optim.Optimizer surface used by the reference loop ------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        self.flat_g.zero_()
        for p in self.params:
            if p.grad is not p._dg_flat_grad:
                p.grad = p._dg_flat_grad

    def _sync_foreign_grads(self):
        # a caller may have replaced .grad (e.g. zero_grad(set_to_none=True) on a plain nn.Module)
        for p in self.params:
            if p.grad is None:
                p._dg_flat_grad.zero_()
                p.grad = p._dg_flat_grad
            elif p.grad is not p._dg_flat_grad and p.grad.data_ptr() != p._dg_flat_grad.data_ptr():
                p._dg_flat_grad.copy_(p.grad)
                p.grad = p._dg_flat_grad

    def ranges_of(self, modules):
        """Flat [begin, end) ranges covering the parameters of the given modules (merged when adjacent)."""
        ids = {id(p) for m in modules for p in m.parameters()}
        out = []
        for p, off in zip(self.params, self.offsets):
            if id(p) in ids:
                end = off + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                if out and out[-1][1] == off:
                    out[-1][1] = end
                else:
                    out.append([off, end])
        return [tuple(r) for r in out]

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, active=None):
        """``active``: flat ranges that received gradients this iteration (from ``ranges_of``).  Like
        torch.optim.Adam, parameters whose ``.grad`` would be None (a network outside the loss of the
        ``recongan`` / ``gan`` architectures, image_translation.py:377-382) are left untouched: no weight
        decay, no moment update.  None = the whole group."""
        g = self.param_groups[0]
        self._sync_foreign_grads()
        ops.adam_advance(self.state, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]))
        p16 = getattr(self, "flat_p16", None)
        p3 = getattr(self, "flat_p3", None)
        for b, e in (active if active is not None else [(0, self.numel)]):
            ops.adam_step_flat(self.flat_p[b:e], self.flat_g[b:e], self.exp_avg[b:e], self.exp_avg_sq[b:e], self.state,
                               float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                               float(grad_scale), p16=None if p16 is None else p16[b:e],
                               p3=None if p3 is None else (p3.data_ptr() + 2 * b, self.numel))
        if p3 is not None and self.flat_p3t is not None:
            ops.x3_transpose_planes(p3, self.flat_p3t, self._x3t_table)

    def state_dict(self):
        return dict(step=self.state[0:1].clone(), exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.state.zero_()
        self.state[0:1].copy_(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v
