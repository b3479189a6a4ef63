"""Optimizer / training-step plumbing for the HIP MAE path (host-side counterpart of run_swin_mae3d.py:588-598,644-669).

FusedAdamW: clip_grad_norm_(max_norm) + AdamW over the model's flat fp32 parameter/gradient buffers in three kernel
launches (norm reduction, clip coefficient, update); hyper-parameters live in a small device tensor so the step is
hipGraph-replayable.  OneCycle: the reference's OneCycleLR defaults (pct_start .3, cos, div 25, final_div 1e4,
beta1 cycling .95->.85->.95), stepped every iteration."""
from __future__ import annotations

import math

import torch

from . import ops


class OneCycle:
    def __init__(self, max_lr: float, total_steps: int, pct_start=0.3, div_factor=25.0, final_div_factor=1e4, base_m=0.85, max_m=0.95):
        self.max_lr, self.total = max_lr, total_steps
        self.init_lr, self.min_lr = max_lr / div_factor, max_lr / div_factor / final_div_factor
        self.up_end = float(pct_start * total_steps) - 1
        self.base_m, self.max_m = base_m, max_m

    @staticmethod
    def _cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

    def at(self, step: int):
        """(lr, beta1) used for optimizer step number `step` (0-based), same as torch OneCycleLR(two-phase, cos)."""
        if step <= self.up_end:
            pct = step / self.up_end if self.up_end > 0 else 1.0
            return self._cos(self.init_lr, self.max_lr, pct), self._cos(self.max_m, self.base_m, pct)
        pct = (step - self.up_end) / (self.total - 1 - self.up_end)
        return self._cos(self.max_lr, self.min_lr, pct), self._cos(self.base_m, self.max_m, pct)


class FusedAdamW:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3, max_grad_norm=0.1):
        if model._flat is None:
            model.flatten_parameters()
        self.model, self.lr, self.betas, self.eps, self.wd, self.max_norm = model, lr, tuple(betas), eps, weight_decay, max_grad_norm
        flat = model._flat
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.acc = torch.zeros(1, dtype=torch.float64, device=flat.device)
        self.coef = torch.ones(1, device=flat.device)
        self.norm = torch.zeros(1, device=flat.device)
        self.hyper = torch.zeros(8, device=flat.device)
        self.t = 0
        self.b1_pow = self.b2_pow = 1.0

    def set_hyper(self, lr=None, beta1=None):
        if lr is not None:
            self.lr = lr
        if beta1 is not None:
            self.betas = (beta1, self.betas[1])

    def step(self):
        self.t += 1
        b1, b2 = self.betas
        self.b1_pow *= b1
        self.b2_pow *= b2
        h = torch.tensor([self.lr, b1, b2, self.eps, self.wd, 1.0 - self.b1_pow, 1.0 - self.b2_pow, 0.0], dtype=torch.float32)
        self.hyper.copy_(h, non_blocking=True)
        g = self.model._flat_grad
        ops.grad_sqnorm(g, self.acc)
        ops.clip_coef(self.acc, float(self.max_norm or 0.0), self.coef, self.norm)
        ops.adamw_step(self.model._flat, g, self.m, self.v, self.hyper, self.coef)
