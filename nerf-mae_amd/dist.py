"""Data-parallel gradient exchange for the MAE path: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

The reference wraps the model in DDP (run_swin_mae3d.py:355-357).  Here the gradients already live in ONE flat fp32 buffer
laid out in forward order (embed, stage0..3, decoder4..1, head), so the exchange is a handful of large contiguous
all-reduces launched from inside backward as soon as a segment is complete -- decoder first (it finishes first and carries
91 % of the FLOPs' worth of backward time behind it), then stage 3..0, then the embed tail -- on a side stream so they
overlap the remaining backward kernels.  xGMI is point-to-point (7 links/GPU), so few large messages beat many small
buckets.  AVG reduction = DDP's gradient averaging; no per-step barrier or scalar loss all-reduce (the reference's
:676-686) -- losses are reduced only when logged."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class _Trigger(torch.autograd.Function):
    """identity in forward; in backward (everything downstream has produced its gradients) launches the all-reduce of a segment"""

    @staticmethod
    def forward(ctx, x, reducer, seg):
        ctx.reducer, ctx.seg = reducer, seg
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.reducer.launch(ctx.seg)
        return g, None, None


def stage_chunk_groups(model, si: int, n: int):
    """the modules of encoder stage `si` in (at most) n consecutive groups of whole Swin blocks, the stage's patch merging with the first"""
    mods = list(model.stages[si])
    lead = [m for m in mods if not hasattr(m, "attn")]
    blocks = [m for m in mods if hasattr(m, "attn")]
    n = max(1, min(n, len(blocks)))
    per = -(-len(blocks) // n)
    groups = [blocks[i * per:(i + 1) * per] for i in range(n) if blocks[i * per:(i + 1) * per]]
    groups[0] = lead + groups[0]
    return groups


class GradReducer:
    """Gradient ranges ("segments") of the flat buffer in forward order:
        0 embed(+mask token) | 1 stage 0 | 2 stage 1 | 3 .. 3+nc-1 the nc block groups of stage 2 | 3+nc stage 3 | 4+nc decoders + head
    Stage 2 carries 70 % of the encoder's parameters (swin_s: 18 of 24 blocks): it is exchanged in `stage_chunks` pieces (default 3, i.e. 6
    blocks each) so that the first all-reduce of the big range is in flight while two thirds of the stage are still in backward."""
    CHUNK_STAGE = 2

    def __init__(self, model, process_group=None, comm_dtype: Optional[torch.dtype] = None, stage_chunks: Optional[int] = None):
        import os
        self.model, self.group = model, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # NMH_DP_FORCE=1: run the whole exchange machinery also over a one-rank group (single-GPU measurements of its cost)
        self.active = self.world > 1 or os.environ.get("NMH_DP_FORCE") == "1"
        if model._flat is None:
            model.flatten_parameters()
        off = model._offsets
        first = lambda mod: min(off[id(p)] for p in mod.parameters() if p.requires_grad)  # noqa: E731
        n = model._flat_grad.numel()
        if stage_chunks is None:
            stage_chunks = int(os.environ.get("NMH_DP_STAGE2_CHUNKS", "3"))
        self.chunk_stage = self.CHUNK_STAGE if len(model.stages) > self.CHUNK_STAGE else -1
        self.chunk_groups = stage_chunk_groups(model, self.chunk_stage, stage_chunks) if self.chunk_stage >= 0 else []
        nc = max(1, len(self.chunk_groups))
        self.nchunks = nc
        # coarse forward-order boundaries: [embed | stage0 | stage1 | stage2 | stage3 | decoders+head]
        sb = [0] + [first(st) for st in model.stages] + [first(model.decoder4), n]
        self.stage_bounds = sb
        b = []
        for si in range(len(model.stages) + 1):       # entry si = start of segment "stage si-1" (entry 0: embed)
            b.append(sb[si])
            if si - 1 == self.chunk_stage and nc > 1:
                b.extend(first(grp[0]) for grp in self.chunk_groups[1:])
        b += sb[len(model.stages) + 1:]
        self.bounds = b
        self.nseg = len(b) - 1
        self.on_gpu = model._flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if (self.active and self.on_gpu) else None
        self.pending: List = []
        self.timing = None            # timing_begin(): {"ranges": [(lo, hi, ev0, ev1)], "joins": [(ev_main, ev_comm_end)]} -- HIP events on the comm / main stream
        self.comm_dtype = comm_dtype
        self.staging = torch.empty(n, dtype=comm_dtype, device=model._flat_grad.device) if (comm_dtype and self.active) else None

    def seg_stage(self, si: int, chunk: int = 0) -> int:
        """segment id of encoder stage si (of its block group `chunk` for the chunked stage)"""
        extra = self.nchunks - 1 if self.chunk_stage >= 0 else 0
        if si < self.chunk_stage or self.chunk_stage < 0:
            return 1 + si
        if si == self.chunk_stage:
            return 1 + si + chunk
        return 1 + si + extra

    def seg_decoder(self) -> int:
        return self.nseg - 1

    def trigger(self, x, seg: int):
        """insert after the forward op whose *inputs* bound segment `seg` from below (see model.forward)"""
        if not self.active or not torch.is_grad_enabled():
            return x
        return _Trigger.apply(x, self, seg)

    def _exchange(self, lo: int, hi: int):
        """mean over ranks of flat_grad[lo:hi] on the CURRENT stream; bf16 buckets when comm_dtype is bf16 (HIP cast kernels, no ATen
        arithmetic).  RCCL averages inside the collective; other backends (gloo: CPU tests, single-GPU dry runs) sum and scale."""
        g = self.model._flat_grad[lo:hi]
        nccl = self.on_gpu and dist.get_backend(self.group) == "nccl"
        if self.staging is not None and self.on_gpu:
            from . import ops
            s = self.staging[lo:hi]
            ops.grad_to_bf16(g, s)
            dist.all_reduce(s, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, group=self.group)
            ops.grad_from_bf16(s, g, 1.0 if nccl else 1.0 / self.world)
        elif nccl:
            dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            g.div_(self.world)

    def allreduce_range(self, lo: int, hi: int, overlap: bool = True):
        """gradient mean of flat_grad[lo:hi] (element offsets, multiples of 8).  RCCL + overlap: issued on the comm stream after everything
        queued on the current stream so far (the producers of that range), so the caller can go on launching backward work; `wait()`
        joins.  Otherwise (gloo dry runs, overlap off): on the current stream."""
        if not self.active or hi <= lo:
            return
        if overlap and self.comm_stream is not None and dist.get_backend(self.group) == "nccl":
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                if self.timing is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                self._exchange(lo, hi)
                if self.timing is not None:
                    e1.record(self.comm_stream)
                    self.timing["ranges"].append((lo, hi, e0, e1))
        else:
            self._exchange(lo, hi)

    def wait(self):
        if self.comm_stream is not None:
            if self.timing is not None and self.timing["ranges"]:
                # what the compute stream really waits for: the end of the last exchange on the comm stream against the moment the compute stream gets here
                em = torch.cuda.Event(enable_timing=True)
                em.record(torch.cuda.current_stream())
                self.timing["joins"].append((em, self.timing["ranges"][-1][3]))
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def timing_begin(self):
        """record HIP events around every overlapped exchange (comm stream) and at every join (compute stream) until timing_report()"""
        self.timing = {"ranges": [], "joins": []}

    def timing_report(self, steps: int):
        """-> {"ranges": [{"lo", "hi", "bytes_on_the_wire", "ms"}...] (mean per step, in issue order), "comm_ms_exposed_events": mean per step of
        max(0, end of the last exchange - arrival of the compute stream at the join)}; call after torch.cuda.synchronize()"""
        t, self.timing = self.timing, None
        if not t or not t["ranges"] or steps <= 0:
            return None
        per = max(1, len(t["ranges"]) // steps)
        esz = 2 if self.comm_dtype == torch.bfloat16 else 4
        out = []
        for k in range(per):
            rs = t["ranges"][k::per]
            lo, hi = rs[0][0], rs[0][1]
            out.append({"lo": lo, "hi": hi, "bytes_on_the_wire": (hi - lo) * esz, "ms": round(sum(a.elapsed_time(b) for _, _, a, b in rs) / len(rs), 4)})
        exposed = sum(max(0.0, em.elapsed_time(ec)) for em, ec in t["joins"]) / steps
        return {"ranges": out, "comm_ms_exposed_events": round(exposed, 4), "joins_per_step": len(t["joins"]) // steps}

    def launch(self, seg: int):
        if not self.active:
            return
        lo, hi = self.bounds[seg], self.bounds[seg + 1]
        if not self.on_gpu:  # CPU/gloo (tests): synchronous
            self._exchange(lo, hi)
            self.pending.append(seg)
            return
        if dist.get_backend(self.group) != "nccl":
            # device tensors over gloo (single-GPU dry runs): no stream-ordered collectives, and a blocking all-reduce issued from the
            # autograd thread deadlocks -- nothing to overlap anyway, finish() reduces the whole buffer from the calling thread
            return
        self.allreduce_range(lo, hi)
        self.pending.append(seg)

    def allreduce_flat(self):
        """the whole flat gradient buffer in one message on the current stream"""
        if not self.active:
            return
        self._exchange(0, self.model._flat_grad.numel())

    def finish(self):
        """call after backward: reduce any segment whose trigger did not fire, then join the comm stream"""
        if not self.active:
            return
        if self.on_gpu and dist.get_backend(self.group) != "nccl":
            self.allreduce_flat()
            self.pending = []
            return
        for seg in range(self.nseg):
            if seg not in self.pending:
                self.launch(seg)
        if self.on_gpu:
            self.wait()
        self.pending = []


def broadcast_parameters(model, src: int = 0, group=None):
    """DDP's initial parameter broadcast: one flat message."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if model._flat is None:
            model.flatten_parameters()
        dist.broadcast(model._flat, src=src, group=group)
        dist.broadcast(model.pos_embed.data, src=src, group=group)


def shard_indices(n: int, rank: int, world: int, epoch: int, shuffle: bool = True, seed: int = 0):
    """torch DistributedSampler semantics (run_swin_mae3d.py:578-586,614): epoch-seeded permutation, padded to a multiple
    of world size, rank r takes r::world."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
    total = (n + world - 1) // world * world
    idx += idx[: total - n]
    return idx[rank:total:world]
