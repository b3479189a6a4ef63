"""Optimizer / training-step plumbing for the HIP MAE path (host-side counterpart of run_swin_mae3d.py:588-598,644-669).

FusedAdamW: clip_grad_norm_(max_norm) + AdamW over the model's flat fp32 parameter/gradient buffers in three kernel
launches (norm reduction, clip coefficient, update); hyper-parameters live in a small device tensor so the step is
hipGraph-replayable.  OneCycle: the reference's OneCycleLR defaults (pct_start .3, cos, div 25, final_div 1e4,
beta1 cycling .95->.85->.95), stepped every iteration."""
from __future__ import annotations

import math
import time

import numpy as np

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


class PinnedRing:
    """Ring of pinned host slots for small per-step uploads (hyper-parameters, token mask, extents).  `upload(src, dst)` copies
    src -> next slot -> dst (non-blocking, stream-ordered) and records an event; a slot is only rewritten after the event of its
    previous upload has completed, so the host can run ahead of the device without corrupting a copy that has not executed yet."""

    def __init__(self, shape, dtype, depth: int = 8, pinned: bool = True):
        self.buf = torch.empty((depth,) + tuple(shape), dtype=dtype)
        if pinned:
            self.buf = self.buf.pin_memory()
        self.events = [None] * depth
        self.i, self.pinned = 0, pinned

    def upload(self, src: torch.Tensor, dst: torch.Tensor):
        k = self.i % len(self.events)
        self.i += 1
        if self.events[k] is not None:
            # polled, not hipEventSynchronize: on this runtime a blocking wait on an (already complete) event was measured to return
            # only when the stream had drained (~100 ms per step in Trainer.fit, the device idle meanwhile)
            while not self.events[k].query():
                time.sleep(1e-4)
        slot = self.buf[k]
        slot.copy_(src.reshape(slot.shape))
        dst.copy_(slot.view(dst.shape), non_blocking=True)
        if self.pinned and dst.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.events[k] = ev


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
        self.zero_grads_after_step = False   # the AdamW kernel also clears the flat gradient buffer (GraphedTrainStep turns it on)

    def set_hyper(self, lr=None, beta1=None):
        if lr is not None:
            self.lr = lr
        if beta1 is not None:
            self.betas = (beta1, self.betas[1])

    def update_hyper(self, upload: bool = True):
        """advance the step counter and hand {lr, betas, eps, wd, bias corrections} to the device (NOT graph-capturable: the values are
        kernel arguments of a tiny launch, ops.step_params -- no host-to-device copy, so the host never waits for the stream).
        upload=False: only compute `self.hyper_host` (GraphedTrainStep sends it together with the mask and the extents)."""
        self.t += 1
        b1, b2 = self.betas
        # torch.optim.AdamW semantics under OneCycleLR momentum cycling: bias corrections from the CURRENT betas, 1 - beta^t
        # (not the running product of the per-step beta1 values, which differs by ~20 % a few steps into a cycled schedule)
        self.hyper_host = [self.lr, b1, b2, self.eps, self.wd, 1.0 - b1 ** self.t, 1.0 - b2 ** self.t, 1.0 if self.zero_grads_after_step else 0.0]
        if upload:
            if self.hyper.is_cuda:
                ops.step_params(hyper=self.hyper_host, hyper_dev=self.hyper)
            else:
                self.hyper.copy_(torch.tensor(self.hyper_host, dtype=torch.float32))

    def launch(self):
        """the three device-side launches (graph-capturable): grad norm, clip coefficient, AdamW update"""
        g = self.model._flat_grad
        ops.grad_sqnorm(g, self.acc)
        ops.clip_coef(self.acc, float(self.max_norm or 0.0), self.coef, self.norm)
        ops.adamw_step(self.model._flat, g, self.m, self.v, self.hyper, self.coef)

    def step(self):
        self.update_hyper()
        self.launch()


# thread-local capture mode: HIP calls made meanwhile by OTHER threads (the input prefetcher's pinned allocations and copies, the RCCL
# watchdog's event queries) must not invalidate the capture of the training thread
_CAPTURE_MODE = "thread_local"


class GraphedTrainStep:
    """One pre-training step (zero_grad, forward, backward, [all-reduce], clip+AdamW) replayed from HIP graphs.

    The step issues ~1.5k kernel launches; at 1 grid/GPU the eager Python/ctypes launch path costs as much wall time as the
    kernels, so the launch-bound loop is captured once (torch.cuda.CUDAGraph on the capture stream our C ABI launches on)
    and replayed.  Per-step inputs live in static device buffers refreshed before each replay: the padded batch, the valid
    extents, the token mask (host python `random`, as the reference) and the optimizer hyper-parameters.  Stochastic-depth
    noise is drawn inside the graph by torch's graph-safe Philox generator.
    With data parallelism the gradient exchange is ONE flat all-reduce between the backward graph and the optimizer graph
    (RCCL collectives are kept out of the captured region)."""

    MAX_EXT_ROWS, MAX_MASK_BITS = 16, 4096   # capacity of one nmh_step_params launch (csrc/misc.hip: ext[48], bits[128])

    def __init__(self, model, opt: FusedAdamW, batch: int, reducer=None, warmup: int = 2):
        self.model, self.opt, self.reducer = model, opt, reducer
        dev = model.mask_token.device
        R, g = model.resolution, model.resolution // 4
        self.x = torch.zeros((batch, 4, R, R, R), device=dev)
        self.ext = torch.full((batch, 3), R, dtype=torch.int32, device=dev)
        self.mask = torch.zeros(g ** 3, dtype=torch.uint8, device=dev)
        # rows per sample of the compact patch embed of the kept tokens (model._EmbedFn): fixed at capture from the mask distribution; every mask is checked
        # against it in __call__ (a mask that keeps more tokens re-captures the step with a row for every token)
        from .model import embed_capacity_rows
        self._embed_cap = embed_capacity_rows(g, p_remove=model.masking_prob)
        self.losses = None
        self._ext_host = None
        self._g1 = self._g2 = self._gb1 = self._gb2 = None
        self._gb, self._ranges = [], []
        self._warm = warmup
        self.overlap = True    # False: every collective on the main stream (A/B measurement of what the overlap hides)
        if reducer is not None and reducer.world > 1 and len(reducer.stage_bounds) != 7:
            raise ValueError("GraphedTrainStep: the data-parallel step expects the gradient segments of a 4-stage encoder + decoder")
        self.comm_captured = False

    def __del__(self):
        st = getattr(self, "_capture_stream", None)
        if st is not None:   # workspaces keyed by this step's capture stream go with it (ops.swin_mlp_split_ws)
            try:
                ops.swin_mlp_split_ws_drop_stream(st.cuda_stream)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def set_extents(self, ext):
        """valid extents [B][3] of the batch now in `self.x` (host list / tensor) -> static device buffer, through a pinned ring"""
        self._ext_host = [[int(v) for v in row] for row in (ext.tolist() if hasattr(ext, "tolist") else ext)]   # sent with the next step's parameters

    def _fwd_bwd(self, zero=True):
        if zero:
            self.model.zero_grad()
        out = self.model.forward_static(self.x, self.ext, self.mask)
        out[0].backward()
        return out

    def _capture(self):
        cap_was, self.model._embed_cap = self.model._embed_cap, self._embed_cap
        try:
            self._capture_graphs()
        finally:
            self.model._embed_cap = cap_was

    def _capture_graphs(self):
        import os
        import torch.distributed as tdist
        ops.side_stream.auto(self.x.shape[0])
        # NMH_DP_FORCE_SPLIT=1 / NMH_DP_FORCE=1: the data-parallel step also with a one-rank group (tools/bench_dp_overhead.py: what the machinery costs)
        dp = self.reducer is not None and (self.reducer.active or os.environ.get("NMH_DP_FORCE_SPLIT") == "1")
        # RCCL collectives CAN be captured into a HIP graph on this stack (tools/probe_rccl_capture.py: replay == eager).  With
        # NMH_DP_CAPTURE_COMM=1 the whole step -- forward, backward with the range all-reduces launched by the autograd triggers on the comm
        # stream, join, clip + AdamW -- is ONE graph without host-side replay boundaries.  Measured over a one-rank RCCL group
        # (tools/bench_dp_overhead.py, ms/step at 1 / 8 grids): single process 12.38 / 58.93, split graphs 13.07 / 60.20, one graph with the
        # collectives inside 13.41 / 60.67 -- no gain on one GPU (what both pay is the bf16 bucket casts and the joins of the weight-gradient
        # side stream in front of every exchange), and plain collectives between graph replays are the more conservative use of RCCL for a
        # first multi-GPU run, so the DEFAULT keeps the collectives outside: the step is cut into pieces at the gradient-range boundaries.
        self.comm_captured = (dp and self.reducer.active and self.reducer.on_gpu and tdist.is_initialized() and tdist.get_backend(self.reducer.group) == "nccl"
                              and os.environ.get("NMH_DP_CAPTURE_COMM", "0") == "1" and os.environ.get("NMH_DP_FORCE_SPLIT") != "1")
        split = dp and not self.comm_captured
        # split mode: the eager path's autograd triggers would issue collectives on the comm stream inside the capture, graph mode owns the
        # exchange (see __call__); captured mode: the triggers ARE the exchange
        self.model._reducer = self.reducer if self.comm_captured else None
        # ONE stream for the warm-up and for every capture below: workspaces that are keyed by the launch stream (ops.swin_mlp_split_ws) are then
        # allocated eagerly by the warm-up, never from a graph's private pool, and their arrival counters are cleared before the capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        ops.swin_mlp_split_ws_reset()
        self._capture_stream = s
        with torch.cuda.stream(s):
            for _ in range(self._warm):
                # warm-up touches no optimizer state (lazy kernel attributes / allocator pools only)
                if split:
                    self._split_a(zero=True); self._split_b1(); self._split_b2()
                else:
                    self._fwd_bwd()
                    if self.comm_captured:
                        self.reducer.finish()
        torch.cuda.current_stream().wait_stream(s)
        # inside the replayed step the gradient buffer is cleared by the optimizer kernel of the previous step (hyper[7]); only the
        # first replay needs it cleared here
        self.opt.zero_grads_after_step = True
        self.model.zero_grad()
        torch.cuda.synchronize()
        self._g1 = torch.cuda.CUDAGraph()
        if not split:
            with torch.cuda.graph(self._g1, stream=s, capture_error_mode=_CAPTURE_MODE):
                out = self._fwd_bwd(zero=False)
                if self.comm_captured:
                    self.reducer.finish()      # ranges whose trigger did not fire (the embed tail) + join of the comm stream
                self.losses = torch.stack([o.detach() for o in out[:3]])
                self.opt.launch()
            return
        # data parallel, collectives outside the graphs: the step is cut where gradient ranges complete, and the collectives run BETWEEN the replays on the comm
        # stream (RCCL stays outside the captured regions), overlapping the next piece of the backward:
        #   g1  = forward + decoder backward                              -> all-reduce [decoders, head]
        #   gb[0] = backward of stage 3 and the LAST third of stage 2     -> all-reduce [stage 2 blocks 12.., stage 3]
        #   gb[1], gb[2] = backward of the middle / first third of stage 2 -> all-reduce of that third (the first exchange of the 48.9 M
        #                  stage-2/3 parameters is in flight while two thirds of stage 2 are still in backward)
        #   gb[3] = backward of stages 1, 0, the embed                    -> all-reduce [mask token, embed, stage 0, stage 1]
        #   g2 = clip + AdamW
        with torch.cuda.graph(self._g1, stream=s, capture_error_mode=_CAPTURE_MODE):
            out = self._split_a(zero=False)
            self.losses = torch.stack([o.detach() for o in out[:3]])
        pool = self._g1.pool()
        self._gb = []
        for k in range(len(self._pieces["back"])):
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, pool=pool, stream=s, capture_error_mode=_CAPTURE_MODE):
                self._split_back(k)
            self._gb.append(gk)
        self._g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g2, pool=pool, stream=s, capture_error_mode=_CAPTURE_MODE):
            self.opt.launch()
        self._gb1 = self._gb[0]            # (kept for callers that test for the split mode)
        self._pieces = None

    # ---- the step in autograd pieces (data-parallel graph mode) ----------------------------------------------------------------------
    def _stage2_chunks(self):
        """block groups of the chunked encoder stage (dist.GradReducer.chunk_groups: stage 2 in three groups of six blocks for swin_s)"""
        from .dist import stage_chunk_groups
        if self.reducer is not None and self.reducer.chunk_groups:
            return self.reducer.chunk_groups
        return stage_chunk_groups(self.model, 2, 3)

    def _split_a(self, zero=True):
        """forward + decoder backward.  The encoder features enter the decoder (and every later piece enters the next) through detached
        leaves, so each piece is its own autograd graph whose incoming gradients are the `.grad` of those leaves."""
        from .model import _EmbedFn, _StageFlushFn, SwinBlock3D
        m = self.model
        if zero:
            m.zero_grad()
        B, R = self.x.shape[0], m.resolution
        g = R // 4
        m._packer.run(late_split=False)   # the stages are captured into separate graphs
        tok = _EmbedFn.apply(m._anchor, m, self.x, self.mask).view(B, g, g, g, m.embed_dim)
        sd = m._draw_sd_noise(B, tok.device)
        m._wq.sync_after_flush = False
        f0, bi = m._run_stage(0, tok, sd, 0)
        f1, bi = m._run_stage(1, f0, sd, bi)
        e1 = f1.detach().requires_grad_()
        groups = self._stage2_chunks()
        grouped = ops.GROUPED_WGRAD and m.compute_dtype == torch.bfloat16
        x, ins, outs = e1, [], []
        for k, grp in enumerate(groups):
            ins.append(x)
            y = _StageFlushFn.apply(x, m._wq) if grouped else x     # backward of the group ends by issuing its queued weight gradients
            for mod in grp:
                if isinstance(mod, SwinBlock3D):
                    y = mod(y.contiguous(), None if sd is None else sd[bi])
                    bi += 1
                else:
                    y = mod(y.contiguous())
            outs.append(y)
            x = y.detach().requires_grad_() if k + 1 < len(groups) else y
        f2 = outs[-1]
        f3, bi = m._run_stage(3, f2, sd, bi)
        d = [f.detach().requires_grad_() for f in (f0, f1, f2, f3)]
        losses = m._decode_and_loss(d, self.x, self.ext, self.mask, None)
        losses[0].backward()
        # gradient ranges of the pieces (element offsets into the flat buffers; parameters lie in module order)
        off = m._offsets
        first = lambda mod: min(off[id(p)] for p in mod.parameters() if p.requires_grad)  # noqa: E731
        b = self.reducer.stage_bounds if self.reducer is not None else None
        s2_lo, s3_hi = (b[3], b[5]) if b is not None else (first(m.stages[2]), first(m.decoder4))
        starts = [s2_lo] + [first(grp[0]) for grp in groups[1:]]
        back = []
        n = len(groups)
        for k in range(n - 1, -1, -1):       # backward order: last group (together with stage 3) first
            hi = s3_hi if k == n - 1 else starts[k + 1]
            back.append({"kind": "s2", "k": k, "lo": starts[k], "hi": hi})
        back.append({"kind": "s01", "lo": 0, "hi": s2_lo})
        self._pieces = {"f": (f0, f1, f2, f3), "e1": e1, "d": d, "ins": ins, "outs": outs, "back": back}
        self._ranges = [(p["lo"], p["hi"]) for p in back]
        return losses[0], losses[1], losses[2]

    def _split_back(self, j: int):
        """piece j of the encoder backward (see _capture)"""
        P = self._pieces
        pc = P["back"][j]
        f0, f1, f2, f3 = P["f"]
        d, ins, outs = P["d"], P["ins"], P["outs"]
        if pc["kind"] == "s2":
            k = pc["k"]
            if k == len(outs) - 1:
                torch.autograd.backward([f3, f2], [d[3].grad, d[2].grad])
            else:
                torch.autograd.backward([outs[k]], [ins[k + 1].grad])
        else:
            e1 = P["e1"]
            ops.add_inplace(e1.grad, d[1].grad)     # stage 1's output feeds stage 2 and decoder3's skip connection
            torch.autograd.backward([f1, f0], [e1.grad, d[0].grad])

    # (compatibility with the three-piece step of round 2: the two encoder pieces by name)
    def _split_b1(self):
        for j, pc in enumerate(self._pieces["back"]):
            if pc["kind"] == "s2":
                self._split_back(j)

    def _split_b2(self):
        self._split_back(len(self._pieces["back"]) - 1)
        self._pieces = None

    def __call__(self, grids=None, block_mask=None):
        """grids: optional list of (4,a0,a1,a2) tensors (copied into the static batch); block_mask: uint8 (g,g,g) host tensor"""
        if grids is not None:
            self.x.zero_()
            ext = []
            for i, t in enumerate(grids):
                a0, a1, a2 = t.shape[1:]
                self.x[i, :, :a0, :a1, :a2].copy_(t, non_blocking=True)
                ext.append([a0, a1, a2])
            self.set_extents(ext)
        if self._g1 is None:
            self._capture()
        self.opt.update_hyper(upload=False)
        g = self.model.resolution // 4
        bits = None
        if block_mask is not None:
            from .model import block_bits_of_mask
            bits = block_mask if isinstance(block_mask, np.ndarray) and block_mask.shape != (g, g, g) else block_bits_of_mask(block_mask)
            if bits is None:   # an arbitrary (not block-structured) token mask: uploaded through the event-guarded pinned ring
                if getattr(self, "_pin_mask", None) is None:
                    self._pin_mask = PinnedRing((self.mask.numel(),), torch.uint8)
                self._pin_mask.upload(torch.as_tensor(block_mask).to(torch.uint8), self.mask)
            elif bits.shape[0] ** 3 > self.MAX_MASK_BITS:   # more blocks than the launch's argument struct holds (resolution > 256): full mask upload
                from .model import draw_block_mask  # noqa: F401  (same block fill, done on the host)
                full = np.zeros((g, g, g), dtype=np.uint8)
                blk = np.asarray(bits, dtype=np.uint8).repeat(4, 0).repeat(4, 1).repeat(4, 2)
                full[:blk.shape[0], :blk.shape[1], :blk.shape[2]] = blk
                if getattr(self, "_pin_mask", None) is None:
                    self._pin_mask = PinnedRing((self.mask.numel(),), torch.uint8)
                self._pin_mask.upload(torch.from_numpy(full), self.mask)
                bits = None
        if block_mask is not None and self._embed_cap < g ** 3:
            kept = g ** 3 - (64 * int(np.asarray(bits).sum()) if bits is not None else int(np.asarray(block_mask).sum()))
            if kept > self._embed_cap:   # (eight standard deviations above the mean kept count: not expected to happen in a training run)
                self._embed_cap = g ** 3
                torch.cuda.synchronize()
                self._capture()
        ext, self._ext_host = self._ext_host, None
        # mask bits + optimizer hyper-parameters + extents as kernel arguments of one launch: nothing in the step waits on a copy.
        # The argument struct holds 16 samples' extents (nmh_step_params): larger batches send the rest in further launches.
        first = ext[:self.MAX_EXT_ROWS] if ext is not None else None
        ops.step_params(tokmask=self.mask if bits is not None else None, block_bits=bits, nb=0 if bits is None else bits.shape[0], g=g,
                        hyper=self.opt.hyper_host, hyper_dev=self.opt.hyper, extents=first, extents_dev=self.ext if ext is not None else None)
        if ext is not None:
            for i0 in range(self.MAX_EXT_ROWS, len(ext), self.MAX_EXT_ROWS):
                ops.step_params(extents=ext[i0:i0 + self.MAX_EXT_ROWS], extents_dev=self.ext[i0:i0 + self.MAX_EXT_ROWS])
        self._g1.replay()
        if self._g2 is not None:
            red, b = self.reducer, self.reducer.stage_bounds
            red.allreduce_range(b[5], b[6], overlap=self.overlap)
            for gk, (lo, hi) in zip(self._gb, self._ranges):
                gk.replay()
                red.allreduce_range(lo, hi, overlap=self.overlap)
            red.wait()
            self._g2.replay()
        return self.losses


# --------------------------------------------------------------------------------------------------
# Trainer parity extras: metrics, checkpoints, epoch loop
# --------------------------------------------------------------------------------------------------
def mse(pred_rgb: torch.Tensor, target_rgb: torch.Tensor, valid_mask: torch.Tensor) -> torch.Tensor:
    """mean squared error over the RGB entries of the voxels selected by valid_mask (nerf_rpn/model/metrics.py:69-75);
    evaluation-only bookkeeping on the eval tuple (stock tensor ops, not part of the training step)"""
    err = (pred_rgb - target_rgb) ** 2
    return err[valid_mask.expand_as(err)].mean()


def psnr(pred_rgb: torch.Tensor, target_rgb: torch.Tensor, valid_mask: torch.Tensor) -> torch.Tensor:
    """-10 log10(mse) (metrics.py:78-79)"""
    return -10.0 * torch.log10(mse(pred_rgb, target_rgb, valid_mask))


def eval_metrics(eval_out) -> tuple:
    """(psnr, mse) of one `model(x, is_eval=True)` 6-tuple exactly as the reference's eval loop computes them
    (run_swin_mae3d.py:747-760).  Both also follow from the fused loss: mse == loss_rgb / 3, because loss_rgb divides the same
    masked sum by the voxel count instead of the entry count (swin_mae3d.py:1535)."""
    _, _, _, pred, mask, target = eval_out
    m = mask.bool()
    return psnr(pred[..., :3], target[..., :3], m), mse(pred[..., :3], target[..., :3], m)


def save_checkpoint(path: str, model, epoch: int, train_args: dict, opt: "FusedAdamW" = None, step: int = None):
    """the reference's checkpoint (run_swin_mae3d.py:471-489): {"epoch", "state_dict", "train_args"}; state_dict keys/shapes are the
    reference's, so nerf_rpn loads it unchanged.  With `opt` the file additionally carries what a TRUE resume needs (the reference
    cannot resume its optimizer): AdamW moments over the flat buffer, step count, beta powers, schedule position."""
    ck = {"epoch": epoch, "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "train_args": dict(train_args)}
    if opt is not None:
        # the moments are flat buffers in the model's parameter order: the layout (name -> offset, numel) travels with them, so a file
        # written under another flat layout (e.g. the 4-element granule of earlier builds) is remapped by name instead of misaligned
        ck["resume"] = {"m": opt.m.detach().cpu(), "v": opt.v.detach().cpu(), "t": opt.t, "lr": opt.lr, "betas": opt.betas, "step": step,
                        "layout": flat_layout(model)}
    torch.save(ck, path)


def flat_layout(model) -> dict:
    """{parameter name: [offset, numel]} of the model's flat fp32 parameter / gradient / moment buffers"""
    return {n: [int(model._offsets[id(p)]), int(p.numel())] for n, p in model.named_parameters() if id(p) in model._offsets}


def load_checkpoint(path: str, model, opt: "FusedAdamW" = None, strict: bool = True) -> dict:
    """loads a reference-format checkpoint (also one written by the reference itself); returns the checkpoint dict.  With `opt` and
    a "resume" section the optimizer continues where it stopped."""
    # tensors, numbers, strings, dicts/lists/tuples only: nothing in either checkpoint format needs unpickling of arbitrary objects
    ck = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(ck["state_dict"], strict=strict)
    if opt is not None and "resume" in ck:
        r = ck["resume"]
        if model._flat is None or not model._flat.is_cuda:
            raise RuntimeError("load_checkpoint: move the model to the HIP device before restoring the optimizer")
        lay_now, lay_ck = flat_layout(model), r.get("layout")
        if lay_ck is None or {k: list(v) for k, v in lay_ck.items()} == lay_now:
            if r["m"].numel() != opt.m.numel():
                raise RuntimeError(f"load_checkpoint: the optimizer state has {r['m'].numel()} elements, this build's flat layout {opt.m.numel()} "
                                   "(a file without a 'layout' section written under another parameter granule cannot be remapped)")
            opt.m.copy_(r["m"]); opt.v.copy_(r["v"])
        else:   # another flat layout: remap every parameter's moments by name
            if set(lay_ck) != set(lay_now) or any(lay_ck[k][1] != lay_now[k][1] for k in lay_now):
                raise RuntimeError("load_checkpoint: the optimizer state belongs to a different parameter set")
            opt.m.zero_(); opt.v.zero_()
            for k, (off, n) in lay_now.items():
                o2 = lay_ck[k][0]
                opt.m[off:off + n].copy_(r["m"][o2:o2 + n]); opt.v[off:off + n].copy_(r["v"][o2:o2 + n])
        opt.t, opt.lr, opt.betas = r["t"], r["lr"], tuple(r["betas"])
    return ck


class Trainer:
    """Epoch loop of run_swin_mae3d.py:600-709 without wandb: OneCycle schedule stepped every iteration, HIP-graph train step,
    evaluation every `eval_interval` epochs, `model_best.pt` by validation PSNR (:620-629) and `epoch_{k}.pt`; rank 0 writes.
    `train_scenes` / `val_scenes`: sequences of stored scenes ((W,L,H,4) arrays or paths); sharded across ranks with
    DistributedSampler semantics (dist.shard_indices)."""

    def __init__(self, model, train_scenes, val_scenes=None, batch_size: int = 4, num_epochs: int = 1, lr: float = 1e-4, weight_decay: float = 1e-3,
                 clip_grad_norm: float = 0.1, eval_interval: int = 1, save_path: str = None, normalize_density: bool = True, flip_prob: float = 0.0,
                 rotate_prob: float = 0.0, rank: int = 0, world: int = 1, reducer=None, log=print, train_args: dict = None, seed: int = 0):
        from . import data, dist as _dist
        self.model, self.train_scenes, self.val_scenes = model, train_scenes, val_scenes
        self.batch, self.epochs, self.eval_interval, self.save_path = batch_size, num_epochs, eval_interval, save_path
        self.rank, self.world, self.log, self.seed = rank, world, log, seed
        self.args = dict(train_args or {}, lr=lr, weight_decay=weight_decay, clip_grad_norm=clip_grad_norm, batch_size=batch_size * world,
                         num_epochs=num_epochs, normalize_density=normalize_density, flip_prob=flip_prob, rotate_prob=rotate_prob)
        self._shard = _dist.shard_indices
        dev = model.mask_token.device
        self.batcher = data.GridBatcher(model.resolution, dev, normalize_density, flip_prob, rotate_prob)
        self.val_batcher = data.GridBatcher(model.resolution, dev, normalize_density)
        self.opt = FusedAdamW(model, lr=lr, weight_decay=weight_decay, max_grad_norm=clip_grad_norm)
        # the reference's DataLoader has no drop_last: len(train_loader) is the ceiling, and it sizes OneCycleLR (run_swin_mae3d.py:594-598)
        self.steps_per_epoch = -(-len(self._shard(len(train_scenes), rank, world, 0)) // batch_size)
        if self.steps_per_epoch < 1:
            raise ValueError("Trainer: empty training shard")
        if world > 1:
            _dist.broadcast_parameters(model)   # DDP's initial broadcast: every rank starts from rank 0's weights
        self.sched = OneCycle(lr, num_epochs * self.steps_per_epoch)
        self.step_fn = GraphedTrainStep(model, self.opt, batch_size, reducer=reducer)
        self.global_step, self.best_metric, self.history = 0, None, []

    def _scene(self, s):
        from . import data
        return data.load_scene(s) if isinstance(s, str) else s

    def train_epoch(self, epoch: int):
        import random
        from . import data
        from .model import draw_block_mask
        self.model.train()
        idx = self._shard(len(self.train_scenes), self.rank, self.world, epoch, seed=self.seed)
        g = self.model.resolution // 4
        batches = [[self.train_scenes[i] for i in idx[b * self.batch:(b + 1) * self.batch]] for b in range(self.steps_per_epoch)]
        # input pipeline: batch k+1 is loaded, staged through pinned rings, copied and prepared on a copy stream by a background thread
        # while step k runs (the reference's DataLoader workers); the losses are read back once per epoch, not per step, so the host
        # keeps queueing replays ahead of the device
        if self.step_fn._g1 is None:
            self.step_fn._capture()    # before the producer thread exists: its allocations / copies would invalidate a global-mode capture
        pf = data.Prefetcher(self.batcher, batches, self.batch, load=self._scene, seed=1000003 * self.seed + 7919 * epoch + self.rank)
        main = torch.cuda.current_stream()
        loss_acc = torch.zeros((), device=self.step_fn.x.device)
        for j, xb, ext, ev in pf:
            main.wait_event(ev)
            n = xb.shape[0]
            self.step_fn.x[:n].copy_(xb, non_blocking=True)          # device-to-device into the graph's static input
            pf.done(j, main)
            if n < self.batch:
                # ragged last batch of the epoch (no drop_last in the reference): the unused slots of the static batch become empty
                # grids with zero valid extent -- no voxel of them enters either loss mask, so losses and gradients equal those of
                # the smaller batch
                self.step_fn.x[n:].zero_()
                ext = ext + [[0, 0, 0]] * (self.batch - n)
            self.step_fn.set_extents(ext)
            lr, b1 = self.sched.at(self.global_step)
            self.opt.set_hyper(lr=lr, beta1=b1)
            losses = self.step_fn(None, draw_block_mask((g, g, g), self.model.masking_prob, rng=random))
            self.global_step += 1
            loss_acc += losses[0]
        self.prefetch_stats = getattr(pf, "stats", None)
        return float(loss_acc) / self.steps_per_epoch

    @torch.no_grad()
    def evaluate(self):
        self.model.eval()
        ps, ms = [], []
        for s in self.val_scenes:
            xb, ext = self.val_batcher([self._scene(s)], flags=[0])
            a0, a1, a2 = ext[0].tolist()
            out = self.model([xb[0, :, :a0, :a1, :a2]], is_eval=True)
            p, m = eval_metrics(out)
            ps.append(float(p)); ms.append(float(m))
        self.model.train()
        return sum(ps) / len(ps), sum(ms) / len(ms)

    def fit(self):
        import os
        if self.save_path and self.rank == 0:
            os.makedirs(self.save_path, exist_ok=True)
        for epoch in range(1, self.epochs + 1):
            loss = self.train_epoch(epoch)
            rec = {"epoch": epoch, "train_loss": loss}
            if self.rank == 0 and self.val_scenes and (epoch % self.eval_interval == 0 or epoch == self.epochs):
                rec["psnr"], rec["mse"] = self.evaluate()
                if self.save_path:
                    if self.best_metric is None or rec["psnr"] > self.best_metric:
                        self.best_metric = rec["psnr"]
                        save_checkpoint(os.path.join(self.save_path, "model_best.pt"), self.model, epoch, self.args, self.opt, self.global_step)
                    save_checkpoint(os.path.join(self.save_path, f"epoch_{epoch}.pt"), self.model, epoch, self.args, self.opt, self.global_step)
            self.history.append(rec)
            if self.rank == 0:
                self.log(" ".join(f"{k}={v:.5g}" if isinstance(v, float) else f"{k}={v}" for k, v in rec.items()))
        return self.history
