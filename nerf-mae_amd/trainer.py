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
        # {lr, b1, b2, eps, wd, 1-b1^t, 1-b2^t}: lr 0 and unit bias corrections until update_hyper() => a launch is a no-op
        self.hyper = torch.tensor([0.0, betas[0], betas[1], eps, 0.0, 1.0, 1.0, 0.0], device=flat.device)
        self.t = 0
        self.b1_pow = self.b2_pow = 1.0

    def set_hyper(self, lr=None, beta1=None):
        if lr is not None:
            self.lr = lr
        if beta1 is not None:
            self.betas = (beta1, self.betas[1])

    def update_hyper(self):
        """advance the step counter and upload {lr, betas, eps, wd, bias corrections} (host->device; NOT graph-capturable)"""
        self.t += 1
        b1, b2 = self.betas
        self.b1_pow *= b1
        self.b2_pow *= b2
        h = torch.tensor([self.lr, b1, b2, self.eps, self.wd, 1.0 - self.b1_pow, 1.0 - self.b2_pow, 0.0], dtype=torch.float32)
        self.hyper.copy_(h, non_blocking=True)

    def launch(self):
        """the three device-side launches (graph-capturable): grad norm, clip coefficient, AdamW update"""
        g = self.model._flat_grad
        ops.grad_sqnorm(g, self.acc)
        ops.clip_coef(self.acc, float(self.max_norm or 0.0), self.coef, self.norm)
        ops.adamw_step(self.model._flat, g, self.m, self.v, self.hyper, self.coef)

    def step(self):
        self.update_hyper()
        self.launch()


class GraphedTrainStep:
    """One pre-training step (zero_grad, forward, backward, [all-reduce], clip+AdamW) replayed from HIP graphs.

    The step issues ~1.5k kernel launches; at 1 grid/GPU the eager Python/ctypes launch path costs as much wall time as the
    kernels, so the launch-bound loop is captured once (torch.cuda.CUDAGraph on the capture stream our C ABI launches on)
    and replayed.  Per-step inputs live in static device buffers refreshed before each replay: the padded batch, the valid
    extents, the token mask (host python `random`, as the reference) and the optimizer hyper-parameters.  Stochastic-depth
    noise is drawn inside the graph by torch's graph-safe Philox generator.
    With data parallelism the gradient exchange is ONE flat all-reduce between the backward graph and the optimizer graph
    (RCCL collectives are kept out of the captured region)."""

    def __init__(self, model, opt: FusedAdamW, batch: int, reducer=None, warmup: int = 2):
        self.model, self.opt, self.reducer = model, opt, reducer
        dev = model.mask_token.device
        R, g = model.resolution, model.resolution // 4
        self.x = torch.zeros((batch, 4, R, R, R), device=dev)
        self.ext = torch.full((batch, 3), R, dtype=torch.int32, device=dev)
        self.mask = torch.zeros(g ** 3, dtype=torch.uint8, device=dev)
        self.losses = None
        self._g1 = self._g2 = None
        self._warm = warmup

    def _fwd_bwd(self):
        self.model.zero_grad()
        out = self.model.forward_static(self.x, self.ext, self.mask)
        out[0].backward()
        return out

    def _capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warm):
                self._fwd_bwd()     # warm-up touches no optimizer state (lazy kernel attributes / allocator pools only)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g1):
            out = self._fwd_bwd()
            self.losses = torch.stack([o.detach() for o in out[:3]])
            if self.reducer is None or self.reducer.world == 1:
                self.opt.launch()
        if self.reducer is not None and self.reducer.world > 1:
            self._g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g2):
                self.opt.launch()

    def __call__(self, grids=None, block_mask=None):
        """grids: optional list of (4,a0,a1,a2) tensors (copied into the static batch); block_mask: uint8 (g,g,g) host tensor"""
        if grids is not None:
            self.x.zero_()
            ext = []
            for i, t in enumerate(grids):
                a0, a1, a2 = t.shape[1:]
                self.x[i, :, :a0, :a1, :a2].copy_(t, non_blocking=True)
                ext.append([a0, a1, a2])
            self.ext.copy_(torch.tensor(ext, dtype=torch.int32), non_blocking=True)
        if block_mask is not None:
            self.mask.copy_(block_mask.to(torch.uint8).reshape(-1), non_blocking=True)
        if self._g1 is None:
            self._capture()
        self.opt.update_hyper()
        self._g1.replay()
        if self._g2 is not None:
            import torch.distributed as dist
            dist.all_reduce(self.model._flat_grad, op=dist.ReduceOp.AVG, group=self.reducer.group)
            self._g2.replay()
        return self.losses
