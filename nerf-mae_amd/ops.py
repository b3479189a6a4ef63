"""Thin tensor-level wrappers over the C ABI (include/nerfmae_hip.h).  PyTorch is used only for device
memory and the current HIP stream; all arithmetic happens in libnerfmae_hip.so.  Every function raises if
the library is missing or a tensor is not a contiguous CUDA(HIP) tensor -- there is no CPU fallback."""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from ._lib import lib

F32, BF16 = 0, 1
WS = 4
PROFILE = None  # set to a dict by bench.py: {(op, dims...): [(start_event, end_event), ...]} recorded on the launch stream
PROFILE_BYTES = {}  # key -> algorithmic HBM bytes per launch (the HBM-bound elementwise passes; bench.py's roofline.hbm_kernels)


def _prof(key, nbytes=None):
    if PROFILE is None:
        return None
    if nbytes is not None:
        PROFILE_BYTES[key] = nbytes
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    PROFILE.setdefault(key, []).append((a, b))
    a.record(torch.cuda.current_stream())
    return b


class side_stream:
    """`with ops.side_stream():` forks a side HIP stream from the current one (also under graph capture) for kernels that are off
    the critical path (weight/bias gradients); `ops.join_side()` makes the current stream wait for it.  The fork/join pair keeps
    allocator lifetimes trivial: every tensor touched on the side stream outlives the join."""
    _streams = {}
    enabled = __import__("os").environ.get("NMH_NO_SIDE", "0") != "1"
    min_rows = int(__import__("os").environ.get("NMH_SIDE_MIN_ROWS", "0"))  # blocks with fewer token rows stay single-stream

    def __init__(self, enable: bool = True):
        self.enable = enable

    @staticmethod
    def auto(grids_per_step: int):
        """Round 1-2: forked kernels cost extra dependency packets at launch, and with 1-2 grids per step the fork/join pairs only added to a
        launch-bound step (23.4 ms with, 19.4 ms without), so the side stream was used from 3 grids on.  Round 3: most of those joins were the
        per-block ones that also serialised the two queues (model._BlockFn.backward); without them the fork pays at every batch size (1 grid
        11.59 -> 11.46 ms, 2 grids 17.63 -> 17.55, 8 grids 51.4 -> 50.5).  NMH_NO_SIDE=1 / NMH_SIDE=1 override; NMH_SIDE_MIN_GRIDS sets the threshold."""
        env = __import__("os").environ
        if env.get("NMH_NO_SIDE", "0") == "1":
            side_stream.enabled = False
        elif env.get("NMH_SIDE", "0") == "1":
            side_stream.enabled = True
        else:
            side_stream.enabled = grids_per_step >= int(env.get("NMH_SIDE_MIN_GRIDS", "1"))

    def __enter__(self):
        if not (side_stream.enabled and self.enable):
            self.ctx = None
            return self
        dev = torch.cuda.current_device()
        if dev not in side_stream._streams:
            prio = __import__("os").environ.get("NMH_SIDE_PRIORITY")   # (experiment: HIP stream priority of the side queue; larger = lower)
            side_stream._streams[dev] = torch.cuda.Stream(device=dev) if prio is None else torch.cuda.Stream(device=dev, priority=int(prio))
        self.s = side_stream._streams[dev]
        self.s.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(self.s)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
        return False


def join_side():
    if side_stream.enabled:
        s = side_stream._streams.get(torch.cuda.current_device())
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("nerf_mae_amd ops need HIP device tensors (no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("nerf_mae_amd ops need contiguous tensors")


class WinGeom:
    """Token grid (B,H,W,D), padded dims and effective shifts for 4x4x4 shifted windows (swin_mae3d.py:62-81)."""

    def __init__(self, B: int, H: int, W: int, D: int, shift: Sequence[int]):
        self.B, self.H, self.W, self.D = B, H, W, D
        self.P = [(s + WS - 1) // WS * WS for s in (H, W, D)]
        self.shift = [0 if WS >= self.P[a] else int(shift[a]) for a in range(3)]
        self.rows = B * self.P[0] * self.P[1] * self.P[2]  # window-ordered rows (incl. pads)
        self.tokens = B * H * W * D
        self.carr = (ctypes.c_int * 10)(B, H, W, D, self.P[0], self.P[1], self.P[2], *self.shift)


def gemm_nt(A, W, bias=None, act=0, C2=None, resid=None, rowscale=None, rows_per_scale=1, out=None, accumulate=False,
            M=None, N=None, K=None):
    """out[M,N] = epi(A[M,K] @ W[N,K]^T)."""
    _chk(A, W, bias, C2, resid, rowscale, out)
    M = A.shape[0] if M is None else M
    K = A.shape[-1] if K is None else K
    N = W.shape[0] if N is None else N
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    lib().call("nmh_gemm_nt", dt_of(A), A, A.stride(0) if A.dim() > 1 else K, W, W.stride(0), M, N, K, out, out.stride(0), bias, act, C2,
               resid, rowscale, rows_per_scale, int(accumulate), _st())
    return out


_TN_WS = {}
TN_WS_FLOATS = int(__import__("os").environ.get("NMH_TN_WS_MB", "48")) * (1 << 18)   # 0 disables (fp32 atomics instead)


def _tn_workspace(device):
    """scratch for split-contraction partials, one per (device, stream): weight-gradient GEMMs of the main and the side stream overlap"""
    if TN_WS_FLOATS <= 0:
        return None
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _TN_WS.get(key)
    if ws is None:
        ws = _TN_WS[key] = torch.empty(TN_WS_FLOATS, dtype=torch.float32, device=device)
    return ws


def gemm_tn(A, B, dW, rowscale=None, rows_per_scale=1, omode=0, ldo=None, p0=0, p1=0, N=None, K=None, M=None, dbias=None):
    """dW[N,K] += A[M,N]^T @ B[M,K] (fp32 atomics into dW); dbias[N] += column sums of A (optional, same pass)."""
    _chk(A, B, dW, rowscale, dbias)
    M = A.shape[0] if M is None else M
    N = A.shape[1] if N is None else N
    K = B.shape[1] if K is None else K
    ws = _tn_workspace(A.device)
    lib().call("nmh_gemm_tn", dt_of(A), A, A.stride(0), B, B.stride(0), dW, M, N, K, rowscale, rows_per_scale, omode,
               K if ldo is None else ldo, p0, p1, dbias, ws, 0 if ws is None else ws.numel(), _st())
    return dW


class _TnProblem(ctypes.Structure):   # include/nerfmae_hip.h: nmh_tn_problem
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("B", ctypes.c_void_p), ("ldb", ctypes.c_int64), ("dW", ctypes.c_void_p),
                ("ldo", ctypes.c_int64), ("dbias", ctypes.c_void_p), ("rowscale", ctypes.c_void_p), ("M", ctypes.c_int64), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("rows_per_sample", ctypes.c_int), ("stride_k", ctypes.c_int64), ("up_k", ctypes.c_int), ("up_v", ctypes.c_int),
                ("bias_atomic", ctypes.c_int), ("n_inner", ctypes.c_int), ("stride_n2", ctypes.c_int64)]


GROUPED_WGRAD = __import__("os").environ.get("NMH_TNG", "1") != "0"
DEFER_DECODER_WGRAD = __import__("os").environ.get("NMH_DEFER_DEC", "1") == "1"
DEC_WGRAD_NOW = __import__("os").environ.get("NMH_DEC_WGRAD_NOW", "0") == "1"   # small decoder levels: weight gradients issued on the side stream as soon as their operands exist instead of with the stage-3 flush
UPW_EARLY = __import__("os").environ.get("NMH_UPW_EARLY", "0") == "1"   # decoder1 transpose-conv weight gradient on the side stream at the end of its block instead of queued (measured 51.1 / 51.0 vs 50.9 / 51.0 ms: off)
WQ_LATE_JOIN = __import__("os").environ.get("NMH_WQ_LATE_JOIN", "1") == "1"   # weight-gradient queue: join only at the end of the backward pass
# A stage's flush forks off behind the FIRST input-gradient kernel of the next stage instead of in front of it (round 5).  In the captured graph both are
# successors of the stage's last kernel; the HIP graph executor keeps the successor captured first on the predecessor's queue and gives the other one a new
# queue plus a cross-queue wait -- and that wait was released only ~10 side kernels later (1 grid: 0.58 ms with the input-gradient queue idle at the
# stage 3 -> 2 boundary, profiles/r5k_timeline_1grids.txt 6.86 -> 7.43 ms).  Captured first, the chain stays on its queue and the side launches take the wait.
WQ_FLUSH_AFTER_FIRST = __import__("os").environ.get("NMH_WQ_FLUSH_AFTER_FIRST", "1") == "1"
STAGE0_BLOCK_FLUSH = __import__("os").environ.get("NMH_STAGE0_BLOCK_FLUSH", "0") == "1"   # stage 0 flushes its queued weight gradients per block (measured 52.2-52.4 vs 52.0 ms at 8 grids: off)


TNG_FOREGROUND = __import__("os").environ.get("NMH_TNG_FOREGROUND", "1") != "0"
EARLY_MLP_WGRAD = __import__("os").environ.get("NMH_EARLY_MLP_WGRAD", "0") == "1"   # stage-0 blocks flush the MLP pair's weight gradients right behind the fused MLP backward
STAGE_FLUSH_BLOCKS = int(__import__("os").environ.get("NMH_STAGE_FLUSH_BLOCKS", "0"))   # single GPU: encoder stages longer than this flush their queued weight gradients in groups of this many blocks (0: once per stage)


class _LnReduceItem(ctypes.Structure):   # include/nerfmae_hip.h: nmh_ln_reduce_item
    _fields_ = [("partials", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p), ("partial_rows", ctypes.c_int64), ("C", ctypes.c_int),
                ("reserved", ctypes.c_int)]


def ln_param_grad_reduce(items):
    """items: [(partials, partial_rows, C, dgamma, dbeta)] -> one launch (nmh_layernorm_param_grad_reduce).  The items of a launch run as independent
    workgroups that ADD into dgamma / dbeta with plain read-modify-writes: the same parameter twice in one flush (two forward passes before one
    backward) is split over consecutive launches, like WgradQueue.flush does for a repeated dW"""
    group, seen = [], set()

    def launch(g):
        if not g:
            return
        arr = (_LnReduceItem * len(g))()
        for i, (part, nb, C, dg, db) in enumerate(g):
            _chk(part, dg, db)
            arr[i] = _LnReduceItem(part.data_ptr(), dg.data_ptr(), db.data_ptr(), nb, C, 0)
        lib().call("nmh_layernorm_param_grad_reduce", arr, len(g), _st())

    for it in items:
        if it[3].data_ptr() in seen:
            launch(group)
            group, seen = [], set()
        group.append(it)
        seen.add(it[3].data_ptr())
    launch(group)


class WgradQueue:
    """Deferred weight gradients of the encoder (bf16): `add` records one dW[N,K] += A[M,N]^T . B[M,K] problem (and keeps its operands
    alive), `flush` issues everything recorded so far through nmh_gemm_tn_grouped -- on the forked side stream when that is enabled, so
    the launch overlaps the input-gradient chain of the next stage -- and `join` makes the current stream wait for it and releases the
    operands.  The input-gradient chain of a stage thus runs without any weight-gradient launch in between."""

    def __init__(self):
        self.pending, self.inflight, self._cb, self.sync_after_flush = [], [], False, False
        self.ln_items = []   # LayerNorm parameter-gradient partials of the current flush group (add_ln_partials)
        self.pad_items = {}  # pad-row column sums of the current flush group, by (geometry, width, dtype) (add_pad_colsum)
        self.deferred = []   # closures (other weight-gradient launches) to issue with the next flush, inside the same fork
        self._due = False    # a flush requested by a stage boundary, issued by the next block behind its first kernel (WQ_FLUSH_AFTER_FIRST)

    def reset(self):
        """start of a forward pass: nothing may be queued here -- unless a previous backward pass raised half-way, in which case its
        operands (whole 160^3 gradients) would stay referenced for good and `_cb` would never re-arm the end-of-backward callback"""
        if self.pending or self.deferred or self.inflight or self._cb:
            join_side()
            self.pending, self.deferred, self.inflight, self._cb, self._due = [], [], [], False, False
            self.ln_items, self.pad_items = [], {}

    def defer(self, fn):
        """queue an arbitrary weight-gradient launch (a closure that keeps its operands alive) for the next flush: the decoder's small-level
        weight gradients then share ONE fork / join with the grouped launch of the first encoder stage instead of a fork each"""
        self.deferred.append(fn)
        if not self._cb:
            self._cb = True
            torch.autograd.Variable._execution_engine.queue_callback(self._final)

    def add_ln_partials(self, part, nb, C, dgamma, dbeta):
        """the dgamma / dbeta partial sums of a LayerNorm backward (ops.layernorm_bwd(wq=...)): all of a flush are reduced by ONE launch"""
        if not self.ln_items:
            items = self.ln_items = []
            self.defer(lambda: ln_param_grad_reduce(items))   # (the list object: filled until the flush runs the closure)
        self.ln_items.append((part, nb, C, dgamma, dbeta))

    def add_pad_colsum(self, x, out, geom):
        """the pad rows' share of a qkv bias gradient (ops.window_pad_rows_colsum): all of a flush with the same geometry go out as ONE launch"""
        key = (tuple(geom.carr), x.shape[1], x.dtype)
        groups = self.pad_items
        if key not in groups:
            lst = groups[key] = []
            self.defer(lambda: window_pad_rows_colsum_grouped(lst, geom))   # (the list object: filled until the flush runs the closure)
        groups[key].append((x, out))

    def launch_now(self, fn):
        """issue a weight-gradient launch on the forked side stream right away (no join: the closure keeps its operands alive until the end-of-backward
        join) -- for work whose operands are ready long before the next flush"""
        with side_stream():
            fn()
        self.inflight.append(fn)
        if not self._cb:
            self._cb = True
            torch.autograd.Variable._execution_engine.queue_callback(self._final)

    def add(self, A, B, dW, dbias=None, rowscale=None, rows_per_sample=None):
        _chk(A, B, dW, dbias, rowscale)
        if A.dtype != torch.bfloat16 or B.dtype != torch.bfloat16:
            raise RuntimeError("WgradQueue: the grouped weight-gradient kernel takes bf16 operands")
        self.pending.append((A, B, dW, dbias, rowscale, A.shape[0] if rows_per_sample is None else rows_per_sample))
        if not self._cb:   # safety net: whatever is still queued when the backward pass ends is issued and joined there
            self._cb = True
            torch.autograd.Variable._execution_engine.queue_callback(self._final)

    def request_flush(self):
        self._due = True
        if not self._cb:
            self._cb = True
            torch.autograd.Variable._execution_engine.queue_callback(self._final)

    def flush_due(self):
        if self._due:
            self._due = False
            self.flush()

    def _final(self):
        self._cb = False
        self._due = False
        self.flush()
        self.join()

    def _launch(self, group, foreground=False):
        arr = (_TnProblem * len(group))()
        for i, (A, B, dW, dbias, rs, rps) in enumerate(group):
            arr[i] = _TnProblem(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), dW.data_ptr(), B.shape[1],
                                0 if dbias is None else dbias.data_ptr(), 0 if rs is None else rs.data_ptr(), A.shape[0], A.shape[1], B.shape[1], rps)
        ws = _tn_workspace(group[0][0].device)
        lib().call("nmh_gemm_tn_grouped_fg" if foreground and TNG_FOREGROUND else "nmh_gemm_tn_grouped", BF16, arr, len(group), ws,
                   0 if ws is None else ws.numel(), _st())

    def flush(self, foreground=False):
        """foreground: the group will run (almost) alone -- the last flushes of a backward pass -- and may split for the whole chip"""
        if not self.pending and not self.deferred:
            return
        todo, self.pending = self.pending, []
        fns, self.deferred = self.deferred, []
        self.ln_items, self.pad_items = [], {}   # (the queued closures hold the lists they work on)
        with side_stream():
            for fn in fns:
                fn()
            if not todo:
                self.inflight.extend(fns)
                return
            group, seen = [], set()
            for pr in todo:
                if pr[2].data_ptr() in seen:   # the same parameter twice (two forward passes before one backward): separate launches
                    self._launch(group, foreground)
                    group, seen = [], set()
                group.append(pr)
                seen.add(pr[2].data_ptr())
            self._launch(group, foreground)
        self.inflight.extend(todo)   # operands stay referenced until the issuing stream has been joined
        self.inflight.extend(fns)

    def join(self):
        join_side()
        self.inflight = []


def conv3d_k3(X, Wp, Cout, out=None, accumulate=False):
    """X (B,D,H,W,Cin) channels-last, Wp packed [Cout][27][Cin] -> (B,D,H,W,Cout)."""
    _chk(X, Wp, out)
    B, D, H, W, Cin = X.shape
    if out is None:
        out = torch.empty((B, D, H, W, Cout), dtype=X.dtype, device=X.device)
    ev = _prof(("conv3d_k3", B, D, Cin, Cout))
    ws = _tn_workspace(X.device) if B * D * H * W <= 65536 else None     # split contraction on the small decoder levels
    lib().call("nmh_conv3d_k3", dt_of(X), X, Wp, out, B, D, H, W, Cin, Cout, int(accumulate), ws, 0 if ws is None else ws.numel(), _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


def conv3d_k3_c48(X, Wk, out=None, accumulate=False, stats_acc=None):
    """specialised Cin=Cout=48 bf16 conv (fragment-ordered weights Wk, pack modes 6/7); stats_acc: optional fp64 [B,48,2] buffer that
    receives the per-(sample,channel) sum / sum-of-squares of the outputs (fused InstanceNorm statistics)"""
    _chk(X, Wk, out, stats_acc)
    B, D, H, W, Cin = X.shape
    if Cin != 48 or X.dtype != torch.bfloat16:
        raise RuntimeError("conv3d_k3_c48 needs bf16 activations with 48 channels")
    if out is None:
        out = torch.empty((B, D, H, W, 48), dtype=X.dtype, device=X.device)
    ev = _prof(("conv3d_k3_c48", B, D, 48, 48))
    lib().call("nmh_conv3d_k3_c48", X, Wk, out, B, D, H, W, int(accumulate), stats_acc, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


TAIL_SIGN_MASK = __import__("os").environ.get("NMH_TAIL_SIGN_MASK", "1") != "0"   # tail backward reads [d0 > 0] bits written by the tail forward instead of the residual
C48_BWD_REDUCE = __import__("os").environ.get("NMH_C48_BWD_REDUCE", "1") != "0"   # decoder1 conv2 input gradient: InstanceNorm-backward sums in the conv epilogue


def conv3d_k3_c48_bwd_reduce(dY, Wkd, y1, stats1, sums, out=None, slope=0.01):
    """dX = conv^T(dY) (48 -> 48, dgrad pack) + the InstanceNorm-backward sums of (dX, y1) in the epilogue (include/nerfmae_hip.h:
    nmh_conv3d_k3_c48_bwd_reduce); sums: fp64 [B,48,2]"""
    _chk(dY, Wkd, y1, stats1, sums, out)
    B, D, H, W, Cin = dY.shape
    if Cin != 48 or dY.dtype != torch.bfloat16 or y1.dtype != torch.bfloat16 or y1.numel() != dY.numel() or sums.dtype != torch.float64:
        raise RuntimeError("conv3d_k3_c48_bwd_reduce needs bf16 tensors with 48 channels and fp64 sums")
    if out is None:
        out = torch.empty((B, D, H, W, 48), dtype=dY.dtype, device=dY.device)
    ev = _prof(("conv3d_k3_c48_bwd_reduce", B, D, 48, 48))
    lib().call("nmh_conv3d_k3_c48_bwd_reduce", dY, Wkd, out, B, D, H, W, y1, stats1, float(slope), sums, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


# decoder1 in the centered form (csrc/cconv.hip): the mean of conv1's output is computed from the coarse tensor before the launch, conv1 stores lrelu(y1 - mean), 1 / std goes
# into conv2's weights per sample -- no normalisation pass over the 160^3 tensor, no second copy of it
CCONV_CENTERED = __import__("os").environ.get("NMH_CCONV_CENTERED", "1") != "0"
C48_IMG = 41 * 3 * 512


def conv48_pack_scaled(W, stats, out, B):
    _chk(W, stats, out)
    if tuple(W.shape) != (48, 48, 3, 3, 3) or W.dtype != torch.float32 or out.numel() < B * C48_IMG or out.dtype != torch.bfloat16:
        raise ValueError("conv48_pack_scaled: shapes")
    lib().call("nmh_conv48_pack_scaled", W, stats, out, B, _st())
    return out


def conv3d_k3_c48_per_sample(X, Wk_per_sample, out=None, stats_acc=None):
    """conv3d_k3_c48 with one weight image per sample (conv48_pack_scaled)"""
    _chk(X, Wk_per_sample, out, stats_acc)
    B, D, H, W, Cin = X.shape
    if Cin != 48 or X.dtype != torch.bfloat16 or Wk_per_sample.numel() < B * C48_IMG:
        raise RuntimeError("conv3d_k3_c48_per_sample needs bf16 activations with 48 channels and B weight images")
    if out is None:
        out = torch.empty((B, D, H, W, 48), dtype=X.dtype, device=X.device)
    ev = _prof(("conv3d_k3_c48", B, D, 48, 48))
    lib().call("nmh_conv3d_k3_c48_per_sample", X, Wk_per_sample, out, B, D, H, W, stats_acc, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


def conv3d_k3_c48_bwd_reduce_centered(dY, Wkd, z, stats1, sums, out=None, slope=0.01):
    """conv3d_k3_c48_bwd_reduce reading z = lrelu(y1 - mean) for y1"""
    _chk(dY, Wkd, z, stats1, sums, out)
    B, D, H, W, Cin = dY.shape
    if Cin != 48 or dY.dtype != torch.bfloat16 or z.dtype != torch.bfloat16 or z.numel() != dY.numel() or sums.dtype != torch.float64:
        raise RuntimeError("conv3d_k3_c48_bwd_reduce_centered needs bf16 tensors with 48 channels and fp64 sums")
    if out is None:
        out = torch.empty((B, D, H, W, 48), dtype=dY.dtype, device=dY.device)
    ev = _prof(("conv3d_k3_c48_bwd_reduce", B, D, 48, 48))
    lib().call("nmh_conv3d_k3_c48_bwd_reduce_centered", dY, Wkd, out, B, D, H, W, z, stats1, float(slope), sums, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


CCONV = __import__("os").environ.get("NMH_CCONV", "1") != "0"   # decoder1 forward: ConvTranspose(k = s = 4) composed with conv1 (csrc/cconv.hip)


def cconv_pack(Wt, W1, bt, Wcp, delta, ws=None):
    """composed decoder1 weights (fragment order, bf16) + [27,48] border table from transp_conv.weight [96,48,4,4,4], conv1.weight [48,48,3,3,3],
    transp_conv.bias; ws: cconv_pack_ws_floats() floats of scratch (allocated here when not given)"""
    _chk(Wt, W1, bt, Wcp, delta, ws)
    if tuple(Wt.shape) != (96, 48, 4, 4, 4) or tuple(W1.shape) != (48, 48, 3, 3, 3) or Wcp.numel() < cconv_pack_numel() or delta.numel() < 27 * 48:
        raise ValueError("cconv_pack: shapes")
    if ws is None:
        ws = torch.empty(cconv_pack_ws_floats(), dtype=torch.float32, device=Wcp.device)
    lib().call("nmh_cconv_pack", Wt, W1, bt, Wcp, delta, ws, _st())


def cconv_pack_centered(Wt, W1, bt, Wcp, delta, ws, mean_table):
    """cconv_pack + the [27,96,48] fp32 table of the composed blocks summed over the phases (include/nerfmae_hip.h: nmh_cconv_pack_centered)"""
    _chk(Wt, W1, bt, Wcp, delta, ws, mean_table)
    if mean_table.numel() < 27 * 96 * 48 or mean_table.dtype != torch.float32:
        raise ValueError("cconv_pack_centered: mean_table must hold 27 x 96 x 48 floats")
    lib().call("nmh_cconv_pack_centered", Wt, W1, bt, Wcp, delta, ws, mean_table, _st())


def cconv_output_mean(x, mean_table, delta, B, v):
    """per-(sample, channel) mean of cconv_fwd's output from the COARSE tensor (nmh_cconv_output_mean) -> fp32 [B,48]"""
    _chk(x, mean_table, delta)
    cls = torch.empty(B * (27 * 96 + 48), dtype=torch.float64, device=x.device)   # class sums + dot-product accumulators
    mean = torch.empty((B, 48), dtype=torch.float32, device=x.device)
    lib().call("nmh_cconv_output_mean", x, mean_table, delta, cls, mean, B, v, _st())
    return mean


def cconv_fwd_centered(x, Wcp, delta, mean, B, v, out=None, stats_acc=None, slope=0.01):
    """z = lrelu(cconv_fwd(x) - mean) + the statistics of cconv_fwd(x) - mean (nmh_cconv_fwd_centered)"""
    _chk(x, Wcp, delta, mean, out, stats_acc)
    if x.dtype != torch.bfloat16 or x.shape[-1] != 96 or v % 8:
        raise RuntimeError("cconv_fwd_centered needs bf16 activations with 96 channels on a coarse grid whose edge is a multiple of 8")
    if out is None:
        out = torch.empty((B, 4 * v, 4 * v, 4 * v, 48), dtype=x.dtype, device=x.device)
    ev = _prof(("cconv_fwd", B, 4 * v, 96, 48))
    lib().call("nmh_cconv_fwd_centered", x, Wcp, delta, mean, float(slope), out, B, v, stats_acc, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


def cconv_pack_ws_floats() -> int:
    return int(lib().call("nmh_cconv_pack_ws_floats"))


def cconv_pack_numel() -> int:
    return int(lib().call("nmh_cconv_pack_numel"))


def cconv_fwd(x, Wcp, delta, B, v, out=None, stats_acc=None):
    """y1 [B,(4v)^3,48] = conv3x3x3(ConvTranspose_{k=s=4}(x [B,v^3,96])) minus the interior bias constant (see include/nerfmae_hip.h)"""
    _chk(x, Wcp, delta, out, stats_acc)
    if x.dtype != torch.bfloat16 or x.shape[-1] != 96 or v % 8:
        raise RuntimeError("cconv_fwd needs bf16 activations with 96 channels on a coarse grid whose edge is a multiple of 8")
    if out is None:
        out = torch.empty((B, 4 * v, 4 * v, 4 * v, 48), dtype=x.dtype, device=x.device)
    ev = _prof(("cconv_fwd", B, 4 * v, 96, 48))
    lib().call("nmh_cconv_fwd", x, Wcp, delta, out, B, v, stats_acc, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


CCONV_DGRAD = __import__("os").environ.get("NMH_CCONV_DGRAD", "1") != "0"   # decoder1: input gradient through the composition (with NMH_CCONV_WGRAD)


def cconv_dgrad_pack_numel() -> int:
    return int(lib().call("nmh_cconv_dgrad_pack_numel"))


def cconv_dgrad_pack(Wcp, Wdp):
    """the composed weights in the input-gradient kernel's fragment order, gathered from the forward's (include/nerfmae_hip.h: nmh_cconv_dgrad_pack)"""
    _chk(Wcp, Wdp)
    if Wdp.numel() < cconv_dgrad_pack_numel() or Wdp.dtype != torch.bfloat16 or Wcp.numel() < cconv_pack_numel():
        raise ValueError("cconv_dgrad_pack: shapes")
    lib().call("nmh_cconv_dgrad_pack", Wcp, Wdp, _st())


def cconv_dgrad(dy1, Wdp, B, v, add=None, out=None):
    """dx [B,v^3,96] = (add or 0) + ConvT^T(conv1^T(dy1 [B,(4v)^3,48])) (include/nerfmae_hip.h: nmh_cconv_dgrad); add may be out"""
    _chk(dy1, Wdp, add, out)
    if dy1.dtype != torch.bfloat16 or v % 8 or dy1.numel() != B * (4 * v) ** 3 * 48:
        raise RuntimeError("cconv_dgrad needs a bf16 gradient with 48 channels on a fine grid whose edge is a multiple of 32")
    if out is None:
        out = torch.empty((B * v ** 3, 96), dtype=dy1.dtype, device=dy1.device)
    if out.numel() != B * v ** 3 * 96 or (add is not None and (add.numel() != out.numel() or add.dtype != out.dtype)):
        raise ValueError("cconv_dgrad: shapes")
    ev = _prof(("cconv_dgrad", B, 4 * v, 48, 96))
    lib().call("nmh_cconv_dgrad", dy1, Wdp, add, out, B, v, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


UPCONV4 = __import__("os").environ.get("NMH_UPCONV4", "1") != "0"   # decoder1's transpose conv as the persistent kernel of csrc/cconv.hip (with NMH_CCONV)


def upconv4_pack_numel() -> int:
    return int(lib().call("nmh_upconv4_pack_numel"))


def upconv4_pack(pack_ws, Wup):
    """fragment-ordered bf16 weights of the k = s = 4 transpose conv from the scratch cconv_pack filled in this step"""
    _chk(pack_ws, Wup)
    if Wup.numel() < upconv4_pack_numel() or Wup.dtype != torch.bfloat16:
        raise ValueError("upconv4_pack: shapes")
    lib().call("nmh_upconv4_pack", pack_ws, Wup, _st())


def upconv4_fwd(x, Wup, bt, out, B, v):
    """u [B,(4v)^3,48] = ConvTranspose3d_{k=s=4}(x [B,v^3,96]) + bt (include/nerfmae_hip.h: nmh_upconv4_fwd)"""
    _chk(x, Wup, bt, out)
    if x.dtype != torch.bfloat16 or out.dtype != torch.bfloat16 or x.shape[-1] != 96 or v % 8 or out.numel() != B * (4 * v) ** 3 * 48:
        raise RuntimeError("upconv4_fwd needs bf16 activations with 96 channels on a coarse grid whose edge is a multiple of 8")
    ev = _prof(("upconv4_fwd", B, 4 * v, 96, 48))
    lib().call("nmh_upconv4_fwd", x, Wup, bt, out, B, v, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


CCONV_WGRAD = __import__("os").environ.get("NMH_CCONV_WGRAD", "1") != "0"   # conv1's weight gradient through the composition (with NMH_CCONV)
CCONV_WGRAD_SPLIT = __import__("os").environ.get("NMH_CCONV_WGRAD_SPLIT", "0") != "0"   # its small launches on the side stream (measured: 52.36 vs 52.34 ms at 8 grids, 29.70 vs 29.64 at 4 -- off)
_CCW_WS = {}


def cconv_wgrad(x, dy1, pack_ws, bt, dW1, B, v, dWt=None, dbt=None, phase=0):
    """conv1.weight gradient [48,48,3,3,3] += through the composed ConvTranspose o conv (include/nerfmae_hip.h: nmh_cconv_wgrad); dy1 must be the
    input gradient of the affine-free InstanceNorm behind conv1 (zero per-sample sums).  dWt [96,48,4,4,4] / dbt [48] (fp32, optional): += the
    transpose conv's own parameter gradients through conv1 (used with cconv_dgrad, when conv1's input gradient on the fine grid is not formed)"""
    _chk(x, dy1, pack_ws, bt, dW1, dWt, dbt)
    if (dWt is None) != (dbt is None) or (dWt is not None and (dWt.numel() != 96 * 48 * 64 or dbt.numel() != 48)):
        raise ValueError("cconv_wgrad: dWt / dbt")
    if x.dtype != torch.bfloat16 or dy1.dtype != torch.bfloat16 or v % 8 or v > 40:
        raise RuntimeError("cconv_wgrad needs bf16 operands on a coarse grid whose edge is a multiple of 8 (<= 40)")
    key = x.device.index          # (one workspace per device: phase 2 may run on a side stream behind phase 1)
    if key not in _CCW_WS:
        _CCW_WS[key] = torch.empty(int(lib().call("nmh_cconv_wgrad_ws_floats")), dtype=torch.float32, device=x.device)
    ev = _prof(("cconv_wgrad", B, 4 * v, 96, 48))
    lib().call("nmh_cconv_wgrad", x, dy1, pack_ws, bt, dW1, dWt, dbt, _CCW_WS[key], B, v, int(phase), _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return dW1


C48MB_MIN_VOXELS = int(__import__("os").environ.get("NMH_C48MB_MIN_VOXELS", "32768"))   # 0 disables the 48-channel-block multi-block kernel


def conv3d_k3_c48mb(X, Wk, Cout, out=None, accumulate=False):
    """3x3x3 conv on 48-channel blocks (bf16, Cin and Cout multiples of 48; one fragment-ordered image per (output block, input block), pack
    modes 6/7 on the [Cout][Cin][27] weight): the decoder1 LDS-halo kernel with (tile, output block, input block) work items"""
    _chk(X, Wk, out)
    B, D, H, W, Cin = X.shape
    if Cin % 48 or Cout % 48 or X.dtype != torch.bfloat16:
        raise RuntimeError("conv3d_k3_c48mb needs bf16 activations with channel counts that are multiples of 48")
    if out is None:
        out = torch.empty((B, D, H, W, Cout), dtype=X.dtype, device=X.device)
    ev = _prof(("conv3d_k3_halo", B, D, Cin, Cout))
    lib().call("nmh_conv3d_k3_c48mb", X, Wk, out, B, D, H, W, Cin, Cout, int(accumulate), _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


def conv3d_k3_c64(X, Wk, Cout, out=None, accumulate=False, stats_acc=None, bias=None):
    """3x3x3 conv on 64-channel blocks (bf16, Cin and Cout multiples of 64; fragment-ordered weights Wk, pack modes 8/9); stats_acc:
    optional fp64 [B,Cout,2] buffer receiving the fused InstanceNorm statistics"""
    _chk(X, Wk, out, stats_acc, bias)
    B, D, H, W, Cin = X.shape
    if Cin % 64 or Cout % 64 or X.dtype != torch.bfloat16:
        raise RuntimeError("conv3d_k3_c64 needs bf16 activations with channel counts that are multiples of 64")
    if out is None:
        out = torch.empty((B, D, H, W, Cout), dtype=X.dtype, device=X.device)
    ev = _prof(("conv3d_k3_halo", B, D, Cin, Cout))
    lib().call("nmh_conv3d_k3_c64", X, Wk, out, B, D, H, W, Cin, Cout, int(accumulate), stats_acc, bias, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


C64_MIN_VOXELS = int(__import__("os").environ.get("NMH_C64_MIN_VOXELS", "32768"))   # 0 disables the 64-channel-block kernels


def use_conv64(X, Cout):
    """dispatch rule of the 64-channel-block LDS-halo conv: bf16, both channel counts multiples of 64 and a volume whose 4x4x16 tiles are
    mostly full (measured: 0.43 vs 0.24 of peak at 160^3 64->64, 0.37 vs 0.29 at 40^3 256->256, break-even at 20^3)"""
    B, D, H, W, Cin = X.shape
    return (C64_MIN_VOXELS > 0 and X.dtype == torch.bfloat16 and Cin % 64 == 0 and Cout % 64 == 0 and D * H * W >= C64_MIN_VOXELS
            and D * H * W * Cin * 2 < 2 ** 32)


def conv64_pack_numel(Cin, Cout):
    return (Cin // 64) * (Cout // 64) * 54 * 4 * 64 * 8


_C48_WS = {}


def conv3d_k3_c48_wgrad(dY, X, dW):
    """dW[48][48][3][3][3] += weight gradient (specialised LDS-halo kernel + per-workgroup partial reduce)"""
    _chk(dY, X, dW)
    B, D, H, W, Cin = X.shape
    key = X.device.index
    if key not in _C48_WS:
        _C48_WS[key] = torch.empty(lib().call("nmh_conv3d_k3_c48_wgrad_ws_floats"), dtype=torch.float32, device=X.device)
    ev = _prof(("conv3d_k3_c48_wgrad", B, D, 48, 48))
    lib().call("nmh_conv3d_k3_c48_wgrad", dY, X, dW, _C48_WS[key], B, D, H, W, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return dW


def conv3d_k3_c48_wgrad_scaled(dY, Z, stats, dW):
    """conv3d_k3_c48_wgrad on (dY, Z) with every workgroup's partial scaled by rstd[sample][ci] (stats [B,48,2]) in the reduce: the weight gradient of a conv whose
    input is rstd * Z (include/nerfmae_hip.h: nmh_conv3d_k3_c48_wgrad_scaled; B in {1, 2, 4, 8})"""
    _chk(dY, Z, stats, dW)
    B, D, H, W, Cin = Z.shape
    key = Z.device.index
    if key not in _C48_WS:
        _C48_WS[key] = torch.empty(lib().call("nmh_conv3d_k3_c48_wgrad_ws_floats"), dtype=torch.float32, device=Z.device)
    ev = _prof(("conv3d_k3_c48_wgrad", B, D, 48, 48))
    lib().call("nmh_conv3d_k3_c48_wgrad_scaled", dY, Z, stats, dW, _C48_WS[key], B, D, H, W, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return dW


_HALO_WS = {}
HALO_WGRAD = __import__("os").environ.get("NMH_HALO_WGRAD", "1") != "0"


def conv3d_k3_wgrad(dY, X, dW):
    """dW[Cout][Cin][3][3][3] += weight gradient.  bf16 with Cin, Cout multiples of 48: LDS-halo kernel on 48x48 channel blocks
    (own workspace: these launches run on the side stream); otherwise the implicit-GEMM kernel."""
    _chk(dY, X, dW)
    B, D, H, W, Cin = X.shape
    Cout = dY.shape[-1]
    ev = _prof(("conv3d_k3_wgrad", B, D, Cin, Cout))
    # (measured at 4 grids: 40^3 192->96 842 -> 354 us, 20^3 384->192 515 -> 243 us; at 10^3 the 4x4x16 tiles are mostly padding
    # and the implicit GEMM wins: 229 vs 267 us)
    if (HALO_WGRAD and X.dtype == torch.bfloat16 and Cin % 48 == 0 and Cout % 48 == 0 and (Cin // 48) * (Cout // 48) <= 256
            and D * H * W >= 4096):
        key = X.device.index
        if key not in _HALO_WS:
            _HALO_WS[key] = torch.empty(lib().call("nmh_conv3d_k3_c48_wgrad_ws_floats"), dtype=torch.float32, device=X.device)
        lib().call("nmh_conv3d_k3_wgrad_halo", dY, X, dW, _HALO_WS[key], B, D, H, W, Cin, Cout, _st())
    elif (HALO_WGRAD and X.dtype == torch.bfloat16 and Cin % 64 == 0 and Cout % 64 == 0 and (Cin // 64) * (Cout // 64) <= 128
          and D * H * W >= 4096):
        key = ("c64", X.device.index)
        if key not in _HALO_WS:
            _HALO_WS[key] = torch.empty(lib().call("nmh_conv3d_k3_c64_wgrad_ws_floats"), dtype=torch.float32, device=X.device)
        lib().call("nmh_conv3d_k3_c64_wgrad", dY, X, dW, _HALO_WS[key], B, D, H, W, Cin, Cout, _st())
    else:
        lib().call("nmh_conv3d_k3_wgrad", dt_of(X), dY, X, dW, B, D, H, W, Cin, Cout, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return dW


def layernorm_fwd(x, gamma, beta, out, mean, rstd, rows, C, src_mode=0, geom: Optional[WinGeom] = None, eps=1e-5,
                  pos=None, mask=None, mask_token=None, tokens_per_sample=1):
    _chk(x, gamma, beta, out, mean, rstd, pos, mask, mask_token)
    lib().call("nmh_layernorm_fwd", dt_of(x), src_mode, x, out, gamma, beta, eps, mean, rstd, rows, C,
               geom.carr if geom is not None else None, pos, mask, mask_token, tokens_per_sample, _st())
    return out


LN_DEFER_PARAM_GRADS = __import__("os").environ.get("NMH_LN_DEFER", "1") != "0"
EMBED_KEPT = __import__("os").environ.get("NMH_EMBED_KEPT", "1") != "0"   # patch embed (im2row, GEMM, LayerNorm, weight gradient) on the kept tokens only


def embed_kept_rows(mask, cap_rows):
    """token mask [n] uint8 (1 = removed) -> int32 rowmap [n + 2] (include/nerfmae_hip.h: nmh_patch_embed_kept_rows)"""
    _chk(mask)
    n = mask.numel()
    rowmap = torch.empty(n + 2, dtype=torch.int32, device=mask.device)
    lib().call("nmh_patch_embed_kept_rows", mask, n, int(cap_rows), rowmap, _st())
    return rowmap


def patch_embed_gather_kept(x, A, B, R, rowmap, cap_rows):
    _chk(x, A, rowmap)
    ev = _prof(("patch_embed_gather", B, R))
    lib().call("nmh_patch_embed_gather_kept", dt_of(A), x, A, B, R, rowmap, int(cap_rows), _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return A


def embed_norm_fwd_kept(y0, gamma, beta, tok, mean, rstd, rows, C, pos, mask, mask_token, tokens_per_sample, rowmap, cap_rows, eps=1e-5):
    _chk(y0, gamma, beta, tok, mean, rstd, pos, mask, mask_token, rowmap)
    lib().call("nmh_patch_embed_norm_fwd_kept", dt_of(y0), y0, tok, gamma, beta, eps, mean, rstd, rows, C, pos, mask, mask_token, tokens_per_sample, rowmap, int(cap_rows), _st())
    return tok


def embed_norm_bwd_kept(dtok, y0, gamma, mean, rstd, dy0, dgamma, dbeta, rows, C, mask, dmask_token, tokens_per_sample, rowmap, cap_rows):
    _chk(dtok, y0, gamma, mean, rstd, dy0, dgamma, dbeta, mask, dmask_token, rowmap)
    lib().call("nmh_patch_embed_norm_bwd_kept", dt_of(y0), dtok, y0, gamma, mean, rstd, dy0, dgamma, dbeta, rows, C, mask, dmask_token, tokens_per_sample, rowmap, int(cap_rows), _st())
    return dy0


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, src_mode=0, geom: Optional[WinGeom] = None, dres=None,
                  mask=None, dmask_token=None, tokens_per_sample=1, dyw=None, dyw_scale=None, wq=None):
    """dyw (mode 0, with geom): second output = dx in window order times dyw_scale[sample] (fused window gather).
    wq (a WgradQueue): dgamma / dbeta leave the launch as per-workgroup partial sums and are added by a reduce queued with the stage's weight gradients"""
    _chk(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dres, mask, dmask_token, dyw, dyw_scale)
    if wq is not None and LN_DEFER_PARAM_GRADS and mask is None and src_mode != 2:
        nb = int(lib().call("nmh_layernorm_bwd_partial_rows", rows, C))
        part = torch.empty((nb, 2 * C), dtype=torch.float32, device=x.device)
        lib().call("nmh_layernorm_bwd_deferred", dt_of(x), src_mode, dy, x, gamma, mean, rstd, dres, dx, part, rows, C,
                   geom.carr if geom is not None else None, dyw, dyw_scale, tokens_per_sample, _st())
        wq.add_ln_partials(part, nb, C, dgamma, dbeta)
        return dx
    lib().call("nmh_layernorm_bwd", dt_of(x), src_mode, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, C,
               geom.carr if geom is not None else None, mask, dmask_token, tokens_per_sample, dyw, dyw_scale, _st())
    return dx


def gemm_nt_window_scatter(A, W, out, resid, bias, rowscale, tokens_per_sample, geom: WinGeom):
    """out[tok] = resid[tok] + rowscale[sample] * (A[m] @ W^T + bias) for window-ordered rows m (proj + window reverse + residual)"""
    _chk(A, W, out, resid, bias, rowscale)
    M, K = A.shape
    N = W.shape[0]
    lib().call("nmh_gemm_nt_window_scatter", dt_of(A), A, A.stride(0), W, W.stride(0), M, N, K, out, resid, bias, rowscale, tokens_per_sample,
               geom.carr, _st())
    return out


TOKEN_BWD_UNFUSED = __import__("os").environ.get("NMH_TOKEN_BWD_UNFUSED", "1") != "0"   # ... also behind the unfused attention forward (small launches, stage 3, fp32)
TOKEN_BWD = __import__("os").environ.get("NMH_TOKEN_BWD", "1") != "0"   # padded stages behind a fused attention forward: token-ordered backward (model._BlockFn)
MLP_FUSED = __import__("os").environ.get("NMH_MLP_FUSED", "1") != "0"
MLP_FUSED_MIN_ROWS = int(__import__("os").environ.get("NMH_MLP_FUSED_MIN_ROWS", "4096"))   # fewer rows: too few 64-row workgroups to fill the chip, the unfused chain wins


# widths that take the fused kernels by default.  Measured (tools/bench_mlp_fused.py, 8 grids per GPU, forward + backward): C = 96 (persistent
# kernels, both weight matrices resident in LDS) 187 + 478 us against 457 + 555 us for the unfused chain; the chunk-ring kernels win the
# forward at C = 192 (77 vs 136 us) but lose it back in the backward (249 vs 204 us) and lose both at C = 384 (8000 rows = 125 workgroups,
# each a serial chain of 48 weight chunks) -- those widths stay on the unfused chain unless listed here
MLP_FUSED_WIDTHS = tuple(int(v) for v in __import__("os").environ.get("NMH_MLP_FUSED_WIDTHS", "96").split(",") if v)


def mlp_fused_ok(x, C: int, rows: int) -> bool:
    """dispatch rule of the fused MLP kernels: bf16, a width that profits (MLP_FUSED_WIDTHS) and enough rows to give every CU work"""
    return (MLP_FUSED and x.dtype == torch.bfloat16 and C in MLP_FUSED_WIDTHS and rows >= MLP_FUSED_MIN_ROWS
            and bool(lib().call("nmh_mlp_fused_supported", C)))


def mlp_fused_fwd(x1, gamma, beta, W1, b1, W2T, b2, rowscale=None, rows_per_scale=1, out=None, mean=None, rstd=None, eps=1e-5):
    """x2 = x1 + rowscale[row / rows_per_scale] * (gelu(LN(x1) @ W1^T + b1) @ W2 + b2) in one launch (bf16); W1 [4C,C], W2T = fc2.weight^T [4C,C]"""
    _chk(x1, gamma, beta, W1, b1, W2T, b2, rowscale, out, mean, rstd)
    M, C = x1.shape
    if out is None:
        out = torch.empty_like(x1)
    ev = _prof(("mlp_fused_fwd", M, C))
    lib().call("nmh_mlp_fused_fwd", x1, gamma, beta, W1, b1, W2T, b2, rowscale, rows_per_scale, out, mean, rstd, M, C, eps, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


def mlp_fused_bwd(x1, dx2, gamma, beta, W1, b1, W2T, dgamma, dbeta, rowscale=None, rows_per_scale=1, dyw=None, dyw_scale=None, geom: Optional[WinGeom] = None, eps=1e-5):
    """-> (dx1, x1n, hact, dh); dgamma / dbeta accumulated; dyw (optional, with geom): dx1 in window order times dyw_scale[sample]"""
    _chk(x1, dx2, gamma, beta, W1, b1, W2T, dgamma, dbeta, rowscale, dyw, dyw_scale)
    M, C = x1.shape
    dx1, x1n = torch.empty_like(x1), torch.empty_like(x1)
    hact = torch.empty((M, 4 * C), dtype=x1.dtype, device=x1.device)
    dh = torch.empty_like(hact)
    ev = _prof(("mlp_fused_bwd", M, C))
    lib().call("nmh_mlp_fused_bwd", x1, dx2, gamma, beta, W1, b1, W2T, rowscale, rows_per_scale, dx1, x1n, hact, dh, dgamma, dbeta, dyw, dyw_scale,
               geom.carr if geom is not None else None, M, C, eps, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return dx1, x1n, hact, dh


# ---- fused Swin-block kernels (csrc/swin_block.hip): bf16, C = 96 * {1, 2, 4} ----
SWIN_ATTN_FWD, SWIN_MLP_FWD = range(2)   # weight-stream kinds (nmh_swin_pack)


class _SwinPackItem(ctypes.Structure):   # include/nerfmae_hip.h: nmh_swin_pack_item
    _fields_ = [("w0", ctypes.c_void_p), ("w1", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("type", ctypes.c_int), ("C", ctypes.c_int)]


SWIN_FUSED = __import__("os").environ.get("NMH_SWIN", "1") != "0"
# dispatch thresholds (tools/bench_swin_block.py): a fused workgroup is one dependent chain of ~45 us whatever the launch size, so launches that
# cannot give most CUs a workgroup stay on the unfused kernels (one grid per GPU: 27 windows / 16 row tiles at stage 2)
SWIN_ATTN_MIN_WINDOWS = int(__import__("os").environ.get("NMH_SWIN_ATTN_MIN_WIN", "100"))
SWIN_MLP_MIN_ROWS = int(__import__("os").environ.get("NMH_SWIN_MLP_MIN_ROWS", "6000"))
SWIN_MLP_WIDTHS = tuple(int(v) for v in __import__("os").environ.get("NMH_SWIN_MLP_WIDTHS", "192,384").split(",") if v)   # C = 96 keeps csrc/mlp_fused.hip (weights resident in LDS)
SWIN_ATTN_WIDTHS = tuple(int(v) for v in __import__("os").environ.get("NMH_SWIN_ATTN_WIDTHS", "96,192,384").split(",") if v)


def swin_attn_ok(x, C: int, geom) -> bool:
    return SWIN_FUSED and x.dtype == torch.bfloat16 and C in SWIN_ATTN_WIDTHS and geom.rows // 64 >= SWIN_ATTN_MIN_WINDOWS


def swin_mlp_ok(x, C: int, rows: int) -> bool:
    return SWIN_FUSED and x.dtype == torch.bfloat16 and C in SWIN_MLP_WIDTHS and rows >= SWIN_MLP_MIN_ROWS


def swin_supported(C: int) -> bool:
    return bool(lib().call("nmh_swin_supported", C))


def swin_stream_numel(kind: int, C: int) -> int:
    return int(lib().call("nmh_swin_stream_numel", kind, C))


def swin_pack_items(items):
    """items: [(w0, w1 | None, dst, kind, C)] -> ctypes array for swin_pack (built once; the pointers are stable: flat parameter buffer, packed buffer)"""
    arr = (_SwinPackItem * len(items))()
    for i, (w0, w1, dst, kind, C) in enumerate(items):
        _chk(w0, w1, dst)
        arr[i] = _SwinPackItem(w0.data_ptr(), w1.data_ptr() if w1 is not None else None, dst.data_ptr(), kind, C)
    return arr


def swin_pack(arr):
    lib().call("nmh_swin_pack", arr, len(arr), _st())


def swin_attn_fwd(x, gamma, beta, wstream, bqkv, table, bproj, geom: WinGeom, rowscale=None, rows_per_scale=1, eps=1e-5, token_saves=False):
    """-> (x1, xnw, mean, rstd, qkv, o, lse): the whole attention branch of a Swin block in one launch (+ what its backward / weight gradients read).
    token_saves: xnw and o come back in token order, [T, C] (for the token-ordered backward: window_attn_bwd_tokens)"""
    _chk(x, gamma, beta, wstream, bqkv, table, bproj, rowscale)
    T, C = x.shape
    heads = C // 32
    dev, dt = x.device, x.dtype
    x1 = torch.empty_like(x)
    xnw = torch.empty((T if token_saves else geom.rows, C), dtype=dt, device=dev)
    mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
    qkv = torch.empty((geom.rows, 3 * C), dtype=dt, device=dev)
    o = torch.empty((T if token_saves else geom.rows, C), dtype=dt, device=dev)
    lse = torch.empty(geom.rows * heads, device=dev)
    ev = _prof(("swin_attn_fwd", geom.rows, C))
    lib().call("nmh_swin_attn_fwd", x, gamma, beta, wstream, bqkv, table, bproj, rowscale, rows_per_scale, xnw, mean, rstd, qkv, o, lse, x1, geom.carr, C, eps,
               int(token_saves), _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return x1, xnw, mean, rstd, qkv, o, lse


SWIN_MLP_SPLIT = __import__("os").environ.get("NMH_SWIN_SPLIT", "1") != "0"
_SPLIT_WS = {}


def swin_mlp_split_ws(M, C, device):
    """the zero-initialised workspace of the two-workgroups-per-tile MLP forward (None: this shape runs one workgroup per tile); one per shape, device AND
    stream: the blocks of a stage share it (their launches are ordered on the stream and each leaves the arrival counters zero), launches of the same shape on
    another stream -- a second model instance, a standalone stage -- get their own, so that they cannot corrupt each other's counters and partial sums.
    A workspace is never allocated inside a stream capture (it would come from the graph's private pool and outlive the graph in this table): during a
    capture the entry of the capture stream must already exist -- GraphedTrainStep warms up on the stream it then captures on -- and a missing one raises.
    (A launch that aborts half-way leaves non-zero counters behind: swin_mlp_split_ws_reset() clears them.)"""
    key = (device.index, M, C, torch.cuda.current_stream(device).cuda_stream)
    if key not in _SPLIT_WS:
        n = int(lib().call("nmh_swin_mlp_split_ws_bytes", M, C))
        if n > 0 and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("split-MLP workspace requested for the first time inside a stream capture: run one eager step on the capture stream first")
        _SPLIT_WS[key] = torch.zeros(n, dtype=torch.uint8, device=device) if n > 0 else None
    return _SPLIT_WS[key]


def swin_mlp_split_ws_reset(drop: bool = False):
    """zero every split-MLP workspace (after a failed / aborted launch, and at the start of every graph capture: the arrival counters must be zero before the
    next launch); drop=True releases them instead (trainer reset / a stream that is gone)"""
    if drop:
        _SPLIT_WS.clear()
        return
    for ws in _SPLIT_WS.values():
        if ws is not None:
            ws.zero_()


def swin_mlp_split_ws_drop_stream(stream_handle: int):
    """release the workspaces keyed by a stream that is going away (a GraphedTrainStep being deleted)"""
    for k in [k for k in _SPLIT_WS if k[3] == stream_handle]:
        del _SPLIT_WS[k]


def swin_mlp_fwd(x1, gamma, beta, wstream, b1, b2, rowscale=None, rows_per_scale=1, eps=1e-5, want_hact=False, split=None):
    """-> (x2, x1n, hp, mean, rstd[, hact]): the MLP branch in one launch; hp = fc1 pre-activation [M, 4C], hact = gelu(hp) on request.
    split (default SWIN_MLP_SPLIT): two workgroups per 64-row tile where the tiles alone leave half of the chip idle (C = 384, <= 8192 rows)"""
    _chk(x1, gamma, beta, wstream, b1, b2, rowscale)
    M, C = x1.shape
    dev = x1.device
    ws = swin_mlp_split_ws(M, C, dev) if (SWIN_MLP_SPLIT if split is None else split) else None
    x2, x1n = torch.empty_like(x1), torch.empty_like(x1)
    hp = torch.empty((M, 4 * C), dtype=x1.dtype, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    hact = torch.empty_like(hp) if want_hact else None
    ev = _prof(("swin_mlp_fwd", M, C))
    lib().call("nmh_swin_mlp_fwd", x1, gamma, beta, wstream, b1, b2, rowscale, rows_per_scale, x2, x1n, hp, hact, mean, rstd, M, C, eps,
               ws, ws.numel() if ws is not None else 0, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return (x2, x1n, hp, mean, rstd, hact) if want_hact else (x2, x1n, hp, mean, rstd)


def window_scatter_residual(yw, x, out, rowscale, C, geom: WinGeom):
    _chk(yw, x, out, rowscale)
    lib().call("nmh_window_scatter_residual", dt_of(x), yw, x, out, rowscale, C, geom.carr, _st())
    return out


def window_gather_scale(dx, dyw, rowscale, C, geom: WinGeom):
    _chk(dx, dyw, rowscale)
    lib().call("nmh_window_gather_scale", dt_of(dx), dx, dyw, rowscale, C, geom.carr, _st())
    return dyw


def window_attn_fwd(qkv, bias_table, out, lse, heads, C, geom: WinGeom):
    _chk(qkv, bias_table, out, lse)
    lib().call("nmh_window_attn_fwd", dt_of(qkv), qkv, bias_table, out, lse, heads, C, geom.carr, _st())
    return out


def window_attn_bwd(qkv, bias_table, dout, lse, dqkv, dbias_table, heads, C, geom: WinGeom):
    _chk(qkv, bias_table, dout, lse, dqkv, dbias_table)
    lib().call("nmh_window_attn_bwd", dt_of(qkv), qkv, bias_table, dout, lse, dqkv, dbias_table, heads, C, geom.carr, _st())
    return dqkv


def layernorm_fwd_window_tokens(x, gamma, beta, out_window, out_tokens, mean, rstd, C, geom: WinGeom, eps=1e-5):
    """LN1 with the pad -> roll -> partition gather (out_window [geom.rows, C]) plus a token-ordered copy (out_tokens [T, C])"""
    _chk(x, gamma, beta, out_window, out_tokens, mean, rstd)
    lib().call("nmh_layernorm_fwd_window_tokens", dt_of(x), x, out_window, out_tokens, gamma, beta, eps, mean, rstd, geom.rows, C, geom.carr, _st())


def window_attn_fwd_tokens(qkv, bias_table, out_tok, lse, heads, C, geom: WinGeom):
    """window attention core with its output in token order (out_tok [T, C])"""
    _chk(qkv, bias_table, out_tok, lse)
    lib().call("nmh_window_attn_fwd_tokens", dt_of(qkv), qkv, bias_table, out_tok, lse, heads, C, geom.carr, _st())
    return out_tok


def window_attn_bwd_tokens(qkv, bias_table, dout_tok, lse, dqkv_tok, dqkv_pad, dbias_table, heads, C, geom: WinGeom):
    """attention-core backward with token-ordered dO in / d(qkv) out (include/nerfmae_hip.h: nmh_window_attn_bwd_tokens); dqkv_pad [geom.rows, 3C] receives the pad rows"""
    _chk(qkv, bias_table, dout_tok, lse, dqkv_tok, dqkv_pad, dbias_table)
    lib().call("nmh_window_attn_bwd_tokens", dt_of(qkv), qkv, bias_table, dout_tok, lse, dqkv_tok, dqkv_pad, dbias_table, heads, C, geom.carr, _st())
    return dqkv_tok


def window_pad_rows_colsum(x, out, geom: WinGeom):
    """out[n] += sum of x[row][n] over the window rows that hold no token"""
    _chk(x, out)
    lib().call("nmh_window_pad_rows_colsum", dt_of(x), x, x.shape[1], geom.carr, out, _st())


def window_pad_rows_colsum_grouped(items, geom: WinGeom):
    """items: [(x, out)] of one geometry and width -> one launch (nmh_window_pad_rows_colsum_grouped)"""
    if not items:
        return
    n = len(items)
    xs, outs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    N, dt = items[0][0].shape[1], dt_of(items[0][0])
    for i, (x, out) in enumerate(items):
        _chk(x, out)
        if x.shape[1] != N or dt_of(x) != dt:
            raise ValueError("window_pad_rows_colsum_grouped: one width / dtype per call")
        xs[i], outs[i] = x.data_ptr(), out.data_ptr()
    lib().call("nmh_window_pad_rows_colsum_grouped", dt, xs, outs, n, N, geom.carr, _st())


def instnorm_stats(x, stats, scratch, B, V, C, eps=1e-5):
    _chk(x, stats, scratch)
    lib().call("nmh_instnorm_stats", dt_of(x), x, stats, scratch, B, V, C, eps, _st())
    return stats


def instnorm_finalize(acc, stats, B, V, C, eps=1e-5):
    _chk(acc, stats)
    lib().call("nmh_instnorm_finalize", acc, stats, B, V, C, eps, _st())
    return stats


def instnorm_apply(x, stats, out, B, V, C, r=None, stats_r=None, rmode=0, slope=0.01):
    _chk(x, stats, out, r, stats_r)
    ev = _prof(("instnorm_apply", B, V, C, rmode), (2 + (rmode != 0)) * x.numel() * x.element_size())
    lib().call("nmh_instnorm_apply", dt_of(x), x, stats, r, stats_r, rmode, out, B, V, C, slope, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return out


def instnorm_bwd_reduce(dout, out, x, stats, sums, B, V, C, r=None, stats_r=None, sums_r=None, rmode=0, slope=0.01):
    _chk(dout, out, x, stats, sums, r, stats_r, sums_r)
    ev = _prof(("instnorm_bwd_reduce", B, V, C, rmode), (2 + (out is not None) + (rmode == 2)) * x.numel() * x.element_size())
    lib().call("nmh_instnorm_bwd_reduce", dt_of(x), dout, out, x, stats, r, stats_r, rmode, sums, sums_r, B, V, C, slope, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())


def instnorm_bwd_apply(dout, out, x, stats, sums, dx, B, V, C, r=None, stats_r=None, sums_r=None, rmode=0, dr=None,
                       dr_accumulate=False, slope=0.01):
    _chk(dout, out, x, stats, sums, dx, r, stats_r, sums_r, dr)
    ev = _prof(("instnorm_bwd_apply", B, V, C, rmode),
               (3 + (out is not None) + (rmode == 2) + (dr is not None) * (1 + int(dr_accumulate))) * x.numel() * x.element_size())
    lib().call("nmh_instnorm_bwd_apply", dt_of(x), dout, out, x, stats, sums, r, stats_r, sums_r, rmode, dx, dr, int(dr_accumulate),
               B, V, C, slope, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())


INBWD_BG = __import__("os").environ.get("NMH_INBWD_BG", "1") != "0"   # decoder-1: IN backward on a forked stream beside conv2's weight gradient (same-box A/B: -0.7 ms at 8 grids)


def instnorm_bwd_apply_bg(dout, x, stats, sums, dx, B, V, C, slope=0.01):
    """nmh_instnorm_bwd_apply(rmode 0, out = None) as a one-workgroup-per-CU background launch (bf16, C = 48)"""
    _chk(dout, x, stats, sums, dx)
    ev = _prof(("instnorm_bwd_apply_bg", B, V, C), 3 * x.numel() * x.element_size())
    lib().call("nmh_instnorm_bwd_apply_bg", dt_of(x), dout, x, stats, sums, dx, B, V, C, slope, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())


def instnorm_bwd_apply_bg_centered(dout, z, stats, sums, dx, B, V, C, slope=0.01):
    """instnorm_bwd_apply_bg reading z = lrelu(y - mean) for y (centered decoder1)"""
    _chk(dout, z, stats, sums, dx)
    ev = _prof(("instnorm_bwd_apply_bg", B, V, C), 3 * z.numel() * z.element_size())
    lib().call("nmh_instnorm_bwd_apply_bg_centered", dt_of(z), dout, z, stats, sums, dx, B, V, C, slope, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())


def patch_embed_gather(x, A, B, R):
    _chk(x, A)
    lib().call("nmh_patch_embed_gather", dt_of(A), x, A, B, R, _st())
    return A


def upconv_fwd(x, Wt, bias, cat, B, v, k, Cin, Cout):
    """ConvTranspose3d(k=stride): cat[fine voxel][0:Cout] = x[coarse] @ Wt^T + bias (pixel shuffle in the GEMM epilogue)"""
    _chk(x, Wt, bias, cat)
    lib().call("nmh_upconv_fwd", dt_of(x), x, Wt, bias, cat, cat.stride(0), B, v, k, Cin, Cout, _st())
    return cat


def upconv_dgrad(dcat, Wd, dx, B, v, k, Cin, Cout):
    _chk(dcat, Wd, dx)
    lib().call("nmh_upconv_dgrad", dt_of(dcat), dcat, dcat.stride(0), Wd, dx, B, v, k, Cin, Cout, _st())
    return dx


def upconv_wgrad(dcat, x, dW, dbias, B, v, k, Cin, Cout):
    _chk(dcat, x, dW, dbias)
    if GROUPED_UPCONV_WGRAD and dcat.dtype == torch.bfloat16 and Cin % 8 == 0 and Cout % 8 == 0 and B * v ** 3 < 2 ** 31 and k * Cout <= 65535:
        return upconv_wgrad_grouped(dcat, x, dW, dbias, B, v, k, Cin, Cout)
    lib().call("nmh_upconv_wgrad", dt_of(dcat), dcat, dcat.stride(0), x, dW, dbias, B, v, k, Cin, Cout, _st())


# One grouped call for the whole gradient: the k taps along x of a (tz, ty) tap row are one problem -- its A rows are the k*Cout-element
# runs of the fine gradient (contiguous without a skip half; with one, k pieces of Cout at the row stride), folded back to (tx, co) by the
# kernel's two-level column map -- i.e. k^2 problems with full tiles, dcat read exactly once (1 grid/GPU, 40^3 -> 160^3: 416 -> 244 us
# against the split shuffled-view gemm_tn).  NMH_TNG_UP=0 keeps the shuffled-view kernel (fp32 always uses it).
GROUPED_UPCONV_WGRAD = int(__import__("os").environ.get("NMH_TNG_UP", "1"))


def upconv_wgrad_grouped(dcat, x, dW, dbias, B, v, k, Cin, Cout):
    """ConvTranspose3d(kernel = stride = k) weight / bias gradient as one nmh_gemm_tn_grouped call (bf16): dW[ci][co][tap] written
    through output strides, the bias gradient by atomic adds from every problem"""
    _chk(dcat, x, dW, dbias)
    k3, ldc, Vf = k ** 3, dcat.stride(0), v * k
    esz = dcat.element_size()
    arr = (_TnProblem * (k * k))()
    for t in range(k * k):   # problem (tz, ty): N = k*Cout columns (tx, co), tap index (tz*k + ty)*k + tx
        tz, ty = t // k, t % k
        off = ((tz * Vf + ty) * Vf) * ldc
        arr[t] = _TnProblem(dcat.data_ptr() + off * esz, ldc, x.data_ptr(), x.stride(0), dW.data_ptr() + (tz * k + ty) * k * 4, k3, dbias.data_ptr(), 0,
                            B * v ** 3, k * Cout, Cin, v ** 3, Cout * k3, k, v, 1, Cout, 1)
    ws = _tn_workspace(dcat.device)
    lib().call("nmh_gemm_tn_grouped", BF16, arr, k * k, ws, 0 if ws is None else ws.numel(), _st())


def upconv_shuffle_fwd(upre, bias, skip, out, B, v, k, Cout):
    _chk(upre, bias, skip, out)
    lib().call("nmh_upconv_shuffle_fwd", dt_of(upre), upre, bias, skip, out, B, v, k, Cout, _st())
    return out


def upconv_shuffle_bwd(dcat, dupre, dskip, dbias, B, v, k, Cout, has_skip):
    _chk(dcat, dupre, dskip, dbias)
    lib().call("nmh_upconv_shuffle_bwd", dt_of(dcat), dcat, dupre, dskip, dbias, B, v, k, Cout, int(has_skip), _st())


def mae_loss_fwd(d0, Wout, bout, target, extents, tokmask, B, R, Cd, sums, losses, pred=None, dpred=None):
    """sums: fp64[4] (fp64[8] when dpred [B*R^3, 4] fp32 is requested for mae_tail_bwd)"""
    _chk(d0, Wout, bout, target, extents, tokmask, sums, losses, pred, dpred)
    if dpred is not None and sums.numel() < 8:
        raise ValueError("mae_loss_fwd: sums needs 8 entries when dpred is requested")
    lib().call("nmh_mae_loss_fwd", dt_of(d0), d0, Wout, bout, target, extents, tokmask, B, R, Cd, sums, losses, pred, dpred, _st())
    return losses


def mae_loss_bwd(d0, Wout, bout, target, extents, tokmask, B, R, Cd, sums, dd0, dpred8, dWout, dbout):
    _chk(d0, Wout, bout, target, extents, tokmask, sums, dd0, dpred8, dWout, dbout)
    lib().call("nmh_mae_loss_bwd", dt_of(d0), d0, Wout, bout, target, extents, tokmask, B, R, Cd, sums, dd0, dpred8, dWout, dbout, _st())


def mae_tail_fwd(y, stats, r, d0, Wout, bout, target, extents, tokmask, B, R, C, sums, losses, pred=None, dpred=None, slope=0.01, bwd_sums=None, sign_mask=None):
    """d0 = lrelu(IN(y) + r), 1x1 head and loss terms in one pass (instnorm_apply rmode 1 + mae_loss_fwd); d0 = None: not stored"""
    _chk(y, stats, r, d0, Wout, bout, target, extents, tokmask, sums, losses, pred, dpred, bwd_sums, sign_mask)
    if sign_mask is not None and (bwd_sums is None or sign_mask.dtype != torch.uint8 or sign_mask.numel() < B * R ** 3 * 8 or C != 48):
        raise ValueError("mae_tail_fwd: sign_mask needs bwd_sums, 48 channels and B*R^3*8 bytes")
    if dpred is not None and sums.numel() < 8:
        raise ValueError("mae_tail_fwd: sums needs 8 entries when dpred is requested")
    if bwd_sums is not None and (dpred is None or bwd_sums.numel() < B * C * 4 + 4 * C):
        raise ValueError("mae_tail_fwd: bwd_sums needs dpred and B*C*4 + 4*C entries")
    # algorithmic bytes: y and r read once, the fp32 target (4 channels) read once, d(pred) (16 B / voxel) and d0 written if requested
    ev = _prof(("mae_tail_fwd", B, R, C), (2 + (d0 is not None)) * y.numel() * y.element_size() + B * R ** 3 * 16 * (1 + (dpred is not None) + (pred is not None))
               + (B * R ** 3 * 8 if sign_mask is not None else 0))
    lib().call("nmh_mae_tail_fwd", dt_of(y), y, stats, r, d0, Wout, bout, target, extents, tokmask, B, R, C, sums, losses, pred, dpred, slope, bwd_sums, sign_mask, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return losses


TAIL_FROM_COARSE = __import__("os").environ.get("NMH_TAIL_FROM_COARSE", "1") != "0"   # decoder1's tail forms the residual ConvT(x) from the coarse tensor (no 160^3 residual tensor)


def tail_residual_pack_numel() -> int:
    return int(lib().call("nmh_tail_residual_pack_numel"))


def tail_residual_pack(pack_ws, Wr):
    """the 64 phase weights of the k = s = 4 transpose conv as the MFMA fragments of mae_tail_fwd_from_coarse, from the scratch cconv_pack filled in this step"""
    _chk(pack_ws, Wr)
    if Wr.numel() < tail_residual_pack_numel() or Wr.dtype != torch.bfloat16:
        raise ValueError("tail_residual_pack: shapes")
    lib().call("nmh_tail_residual_pack", pack_ws, Wr, _st())


def tail_from_coarse_ok(R: int, C: int, Cin: int, dtype) -> bool:
    """shapes mae_tail_fwd_from_coarse takes (include/nerfmae_hip.h)"""
    return dtype == torch.bfloat16 and C == 48 and Cin == 96 and R % 4 == 0 and R <= 256 and ((R // 4) ** 3) % 16 == 0


def mae_tail_fwd_from_coarse(y, stats, xcoarse, Wr, bt, Wout, bout, target, extents, tokmask, B, R, C, sums, losses, dpred, bwd_sums, sign_mask, pred=None, slope=0.01):
    """mae_tail_fwd (training form: d(pred), the backward's sums and the sign mask are all written) with the residual ConvT_{k=s=4}(xcoarse) + bt formed inside"""
    _chk(y, stats, xcoarse, Wr, bt, Wout, bout, target, extents, tokmask, sums, losses, pred, dpred, bwd_sums, sign_mask)
    if not tail_from_coarse_ok(R, C, xcoarse.shape[-1], y.dtype) or xcoarse.dtype != torch.bfloat16 or xcoarse.numel() != B * (R // 4) ** 3 * 96:
        raise ValueError("mae_tail_fwd_from_coarse: bf16, 48 channels from a 96-channel coarse tensor of edge R / 4, (R / 4)^3 a multiple of 16")
    if sums.numel() < 8 or bwd_sums.numel() < B * C * 4 + 4 * C or sign_mask.dtype != torch.uint8 or sign_mask.numel() < B * R ** 3 * 8 or dpred.numel() < B * R ** 3 * 4:
        raise ValueError("mae_tail_fwd_from_coarse: sums[8], bwd_sums[B*C*4 + 4*C], sign_mask[B*R^3*8], dpred[B*R^3*4]")
    ev = _prof(("mae_tail_fwd", B, R, C), y.numel() * y.element_size() + xcoarse.numel() * 2 + B * R ** 3 * (16 * (2 + (pred is not None)) + 8))
    lib().call("nmh_mae_tail_fwd_from_coarse", dt_of(y), y, stats, xcoarse, Wr, bt, Wout, bout, target, extents, tokmask, B, R, C, sums, losses, pred, dpred, slope,
               bwd_sums, sign_mask, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())
    return losses


def mae_tail_bwd(d0, y, stats, dpred, loss_sums, Wout, in_sums, dy, dr, dWout, dbout, B, V, C, slope=0.01, r=None, bwd_sums=None, sign_mask=None):
    """d0 may be None when r (the forward's residual input) is given: the kernel rebuilds d0 from y, stats and r"""
    _chk(d0, r, y, stats, dpred, loss_sums, Wout, in_sums, dy, dr, dWout, dbout, bwd_sums, sign_mask)
    if d0 is None and r is None and (sign_mask is None or bwd_sums is None):
        raise ValueError("mae_tail_bwd needs d0, r, or the forward's sign mask together with its bwd_sums")
    # two passes (sums, then apply): y and r (or d0) read twice, d(pred) read twice, dy and dr written once
    # (with the forward's sign mask the apply pass reads y, 8 mask bytes and d(pred) per voxel and writes dy and dr: 3 tensor passes, not 4)
    nt = (3 if sign_mask is not None else 4) if bwd_sums is not None else 6
    ev = _prof(("mae_tail_bwd", B, V, C), nt * y.numel() * y.element_size() + (1 if bwd_sums is not None else 2) * B * V * 16
               + (B * V * 8 if sign_mask is not None and bwd_sums is not None else 0))
    lib().call("nmh_mae_tail_bwd", dt_of(y), d0, r, y, stats, dpred, loss_sums, Wout, in_sums, dy, dr, slope, dWout, dbout, B, V, C, bwd_sums, sign_mask, _st())
    if ev is not None:
        ev.record(torch.cuda.current_stream())


GRID_ROT, GRID_FLIP0, GRID_FLIP1, GRID_DENSITY = 1, 2, 4, 8


def grid_prepare(src, dst, R, flags=0):
    """src: one stored scene (W,L,H,4) fp32|uint8 on the device; dst: (4,R,R,R) fp32 slot; returns the valid extents"""
    _chk(src, dst)
    if src.dim() != 4 or src.shape[3] != 4 or src.dtype not in (torch.float32, torch.uint8):
        raise ValueError("grid_prepare: src must be (W,L,H,4) float32 or uint8")
    W, L, H, _ = src.shape
    lib().call("nmh_grid_prepare", int(src.dtype == torch.uint8), src, W, L, H, dst, R, flags, _st())
    return (L, W, H) if flags & GRID_ROT else (W, L, H)


def bias_grad(dY, db, M, N, rowscale=None, rows_per_scale=1):
    _chk(dY, db, rowscale)
    lib().call("nmh_bias_grad", dt_of(dY), dY, db, M, N, rowscale, rows_per_scale, _st())


def step_params(tokmask=None, block_bits=None, nb=0, g=0, hyper=None, hyper_dev=None, extents=None, extents_dev=None):
    """per-step host parameters as kernel arguments of one launch (no H2D copies): block_bits = numpy uint8 array of nb^3 {0,1} in raster
    order -> token mask [g^3]; hyper = 8 floats -> hyper_dev; extents = [B][3] ints -> extents_dev"""
    import numpy as np
    _chk(tokmask, hyper_dev, extents_dev)
    bits = hy = ex = None
    if tokmask is not None:
        packed = np.packbits(np.asarray(block_bits, dtype=np.uint8).reshape(-1), bitorder="little")
        packed = np.concatenate([packed, np.zeros((-len(packed)) % 4, dtype=np.uint8)])
        bits = (ctypes.c_uint32 * max(1, len(packed) // 4)).from_buffer_copy(packed.tobytes() or b"\0\0\0\0")
    if hyper is not None:
        hy = (ctypes.c_float * 8)(*[float(v) for v in hyper])
    n_ext = 0
    if extents is not None:
        flat = [int(v) for row in extents for v in row]
        n_ext = len(flat)
        ex = (ctypes.c_int * max(1, n_ext))(*flat)
    lib().call("nmh_step_params", bits, nb, g, tokmask, hy, hyper_dev, ex, n_ext, extents_dev, _st())


def grad_to_bf16(g, bucket):
    """fp32 gradient range -> bf16 bucket (data-parallel exchange at half the bytes)"""
    _chk(g, bucket)
    lib().call("nmh_grad_to_bf16", g, bucket, g.numel(), _st())
    return bucket


def grad_from_bf16(bucket, g, scale=1.0):
    _chk(g, bucket)
    lib().call("nmh_grad_from_bf16", bucket, g, g.numel(), scale, _st())
    return g


class AccArena:
    """Pre-zeroed accumulator arena (include/nerfmae_hip.h: nmh_set_prezeroed_arena).  `take` hands out zeroed fp64 slices that are used
    ONCE (reduce into, read back) before the next `begin`; `begin` -- called where a step starts to use them -- clears the WHOLE arena
    with one launch (4 MB: a microsecond of HBM time; clearing only the used part would make the launch depend on the history of
    previous steps, which a captured graph does not replay).  Slices that do not fit come from `torch.empty` (outside the registered
    range, so the library clears them itself): the arena is an optimisation, never a correctness dependency.  One per process."""

    BYTES = 4 << 20
    enabled = __import__("os").environ.get("NMH_ACC_ARENA", "1") != "0"
    _inst = None

    def __init__(self, device):
        self.buf = torch.zeros(self.BYTES // 8, dtype=torch.float64, device=device)
        self.off = 0        # doubles handed out since the last begin()
        lib().call("nmh_set_prezeroed_arena", self.buf, self.BYTES)

    @classmethod
    def get(cls, device):
        if not cls.enabled:
            return None
        if cls._inst is None:
            if torch.cuda.is_current_stream_capturing():   # never born inside a graph's private memory pool
                return None
            cls._inst = cls(device)
        return cls._inst if cls._inst.buf.device == torch.device(device) else None

    def begin(self):
        lib().call("nmh_fill_f32", self.buf, 0.0, self.BYTES // 4, _st())
        self.off = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 15) // 16 * 16   # 128-byte slices
        if self.off + n_al > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n].view(*shape)
        self.off += n_al
        return t


def acc_zeros(shape, device):
    """an fp64 accumulator for ONE reduce-and-read use: a zeroed arena slice when the arena is active, else an uninitialised tensor (the
    entry point it is passed to clears accumulators outside the arena itself)"""
    ar = AccArena.get(device)
    t = ar.take(shape) if ar is not None else None
    return t if t is not None else torch.empty(shape, dtype=torch.float64, device=device)


def add(a, b):
    """a + b into a fresh tensor (HIP kernel; used for the two gradients of a feature map with two consumers)"""
    _chk(a, b)
    out = torch.empty_like(a)
    lib().call("nmh_add", dt_of(a), a, b, out, a.numel(), _st())
    return out


def add_inplace(a, b):
    _chk(a, b)
    lib().call("nmh_add_inplace", dt_of(a), a, b, a.numel(), _st())
    return a


def pack_weights(dt, descs_dev, blk2desc_dev, blkstart_dev, nblocks):
    lib().call("nmh_pack_weights", dt, descs_dev, blk2desc_dev, blkstart_dev, nblocks, _st())


def grad_sqnorm(g, acc):
    lib().call("nmh_grad_sqnorm", g, g.numel(), acc, _st())


def clip_coef(acc, max_norm, coef, norm_out=None):
    lib().call("nmh_clip_coef", acc, max_norm, coef, norm_out, _st())


def adamw_step(p, g, m, v, hyper, coef=None):
    lib().call("nmh_adamw_step", p, g, m, v, p.numel(), hyper, coef, _st())


# ---- FPN neck (nerf_rpn/model/fpn.py) -----------------------------------------------------------------------
def conv3d_k3_bias(X, Wp, bias, Cout, out=None):
    """conv3d_k3 with a per-output-channel fp32 bias in the epilogue: X (B,D,H,W,Cin) channels-last, Wp packed [Cout][27][Cin]."""
    _chk(X, Wp, bias, out)
    B, D, H, W, Cin = X.shape
    if out is None:
        out = torch.empty((B, D, H, W, Cout), dtype=X.dtype, device=X.device)
    lib().call("nmh_conv3d_k3_bias", dt_of(X), X, Wp, bias, out, B, D, H, W, Cin, Cout, _st())
    return out


def nearest_upsample_add(coarse, fine):
    """fine (B,Df,Hf,Wf,C) += F.interpolate(coarse (B,Dc,Hc,Wc,C), size=fine, mode='nearest'), channels-last, in place."""
    _chk(coarse, fine)
    B, Dc, Hc, Wc, C = coarse.shape
    _, Df, Hf, Wf, _ = fine.shape
    lib().call("nmh_nearest_upsample_add", dt_of(fine), coarse, fine, B, Dc, Hc, Wc, Df, Hf, Wf, C, _st())
    return fine


def nearest_upsample_add_bwd(dfine, dcoarse):
    """adjoint of nearest_upsample_add: dcoarse += sum of dfine over the fine voxels that read each coarse voxel, in place."""
    _chk(dfine, dcoarse)
    B, Dc, Hc, Wc, C = dcoarse.shape
    _, Df, Hf, Wf, _ = dfine.shape
    lib().call("nmh_nearest_upsample_add_bwd", dt_of(dfine), dfine, dcoarse, B, Dc, Hc, Wc, Df, Hf, Wf, C, _st())
    return dcoarse


def ndhwc_to_ncdhw(x):
    """channels-last compute tensor (B,D,H,W,C) -> fp32 NCDHW (B,C,D,H,W)"""
    _chk(x)
    B, D, H, W, C = x.shape
    out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=x.device)
    lib().call("nmh_ndhwc_to_ncdhw", dt_of(x), x, out, B, D * H * W, C, _st())
    return out


def ncdhw_to_ndhwc(g, dtype):
    """fp32 NCDHW gradient (B,C,D,H,W) -> channels-last (B,D,H,W,C) in the compute dtype"""
    g = g.float().contiguous()
    _chk(g)
    B, C, D, H, W = g.shape
    out = torch.empty((B, D, H, W, C), dtype=dtype, device=g.device)
    lib().call("nmh_ncdhw_to_ndhwc", dt_of(out), g, out, B, D * H * W, C, _st())
    return out


def copy_cols(src, dst):
    """dst[m, :C] = src[m, :C] for 2-D views with arbitrary row strides (unit column stride): skip connection <-> concatenated tensor"""
    if src.dim() != 2 or dst.dim() != 2 or src.shape != dst.shape or src.stride(1) != 1 or dst.stride(1) != 1 or src.dtype != dst.dtype:
        raise RuntimeError("copy_cols needs two 2-D views of equal shape and dtype with unit column stride")
    if not (src.is_cuda and dst.is_cuda):
        raise RuntimeError("nerf_mae_amd ops need HIP device tensors (no CPU fallback)")
    M, C = src.shape
    lib().call("nmh_copy_cols", dt_of(src), src, src.stride(0), dst, dst.stride(0), M, C, _st())
    return dst


# ---- voxel super-resolution / semantics heads (csrc/heads.hip) -----------------------------------------------------
def grid_to_cl8(xb, dtype):
    """(B,4,R,R,R) fp32 grid -> (B,R,R,R,8) channels-last in `dtype`, channels 4..7 zero"""
    _chk(xb)
    B, _, D, H, W = xb.shape
    out = torch.empty((B, D, H, W, 8), dtype=dtype, device=xb.device)
    lib().call("nmh_grid_to_cl8", dt_of(out), xb, out, B, D * H * W, _st())
    return out


def head_upsample_fwd(y, Co, B, R, Ro, inv_scale):
    """y [B*R^3][Cp] -> pred (B,Co,Ro,Ro,Ro) fp32 NCDHW, nearest upsampling with ATen's explicit-scale source index"""
    _chk(y)
    pred = torch.empty((B, Co, Ro, Ro, Ro), dtype=torch.float32, device=y.device)
    lib().call("nmh_head_upsample_fwd", dt_of(y), y, pred, B, Co, y.shape[1], R, Ro, inv_scale, _st())
    return pred


def head_upsample_bwd(dpred, Cp, R, dtype, inv_scale):
    _chk(dpred)
    B, Co, Ro = dpred.shape[0], dpred.shape[1], dpred.shape[2]
    g = torch.empty((B * R ** 3, Cp), dtype=dtype, device=dpred.device)
    lib().call("nmh_head_upsample_bwd", dt_of(g), dpred, g, B, Co, Cp, R, Ro, inv_scale, _st())
    return g


def add_cols_f32(src, dst, C):
    """dst[m][:C] += src[m][:C] for fp32 2-D views (rows = leading dim)"""
    _chk(src, dst)
    lib().call("nmh_add_cols_f32", src, src.stride(0), dst, dst.stride(0), src.shape[0], C, _st())
    return dst


def voxel_sr_loss_fwd(pred, target, sums, loss):
    _chk(pred, target, sums, loss)
    B = pred.shape[0]
    lib().call("nmh_voxel_sr_loss_fwd", pred, target, B, pred[0, 0].numel(), sums, loss, _st())


def voxel_sr_loss_bwd(pred, target, sums, gscale, dpred):
    _chk(pred, target, sums, dpred)
    lib().call("nmh_voxel_sr_loss_bwd", pred, target, pred.shape[0], pred[0, 0].numel(), sums, gscale, dpred, _st())


def masked_ce_fwd(logits, labels, class_weights, sums, iou_sums, out):
    _chk(logits, labels, class_weights, sums, iou_sums, out)
    B, K = logits.shape[:2]
    lib().call("nmh_masked_ce_fwd", logits, labels, class_weights, B, K, logits[0, 0].numel(), sums, iou_sums, out, _st())


def masked_ce_bwd(logits, labels, class_weights, sums, gscale, dlogits):
    _chk(logits, labels, class_weights, sums, dlogits)
    B, K = logits.shape[:2]
    lib().call("nmh_masked_ce_bwd", logits, labels, class_weights, B, K, logits[0, 0].numel(), sums, gscale, dlogits, _st())
