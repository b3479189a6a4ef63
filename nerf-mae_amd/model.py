"""Host-side mirror of the reference's `SwinTransformer_MAE3D_New` (nerf_mae/model/mae/swin_mae3d.py:1067-1599;
imported there under the alias `SwinTransformer_MAE3D`, run_swin_mae3d.py:22).

Same constructor kwargs, `forward(list_of_grids, is_eval)` tuple contract, attribute contract used by nerf_rpn
(`patch_partition`, `pos_embed`, `stages`, deletable `decoder*`/`out`/`mask_token`; feature_extractor.py:1127-1187)
and `state_dict()` keys/shapes, so reference checkpoints load with strict=True and checkpoints written here load
into the reference.  The nn.Linear/Conv3d/LayerNorm children are *parameter holders only*: every forward and
backward op is a hand-written HIP kernel called through the C ABI (ops.py); activations stay channels-last
(B,A0,A1,A2,C) token-major matrices end to end (the reference's NCDHW round trips, swin_mae3d.py:1470, are gone).

Gradients: each block is one torch.autograd.Function with an explicit backward that launches the dgrad/wgrad
kernels; parameter gradients are accumulated by the kernels (fp32 atomics) straight into `param.grad`, which are
views of one flat fp32 buffer (likewise the parameters), so the optimizer and the data-parallel all-reduce work on
two flat buffers."""
from __future__ import annotations

import math
import random
import os
import struct
from functools import partial
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
from torch import Tensor, nn

from . import ops
from .ops import WinGeom

WS = 4


# --------------------------------------------------------------------------------------------------
# fixed tables (restated from the reference's formulas; validated against golden vectors in tests)
# --------------------------------------------------------------------------------------------------
def _sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    ang = pos.reshape(-1).astype(np.float64)[:, None] * omega[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_pos_embed_3d(embed_dim: int, g: int) -> np.ndarray:
    """torch_utils.py:5-31: channel thirds encode (a1, a0, a2) of token (a0,a1,a2); zero-padded when 3*(C//3) < C
    (the defined swin_b deviation)."""
    third = embed_dim // 3
    third -= third % 2
    i, j, k = np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij")
    emb = np.concatenate([_sincos_1d(third, j), _sincos_1d(third, i), _sincos_1d(third, k)], axis=1)
    if emb.shape[1] < embed_dim:
        emb = np.concatenate([emb, np.zeros((emb.shape[0], embed_dim - emb.shape[1]))], axis=1)
    return emb.reshape(1, g, g, g, embed_dim)


def relative_position_index(ws: int = WS) -> Tensor:
    c = torch.stack(torch.meshgrid(*[torch.arange(ws)] * 3, indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]) + (ws - 1)
    m = 2 * ws - 1
    return (rel[0] * m * m + rel[1] * m + rel[2]).flatten()


def embed_capacity_rows(g: int, kept: Optional[int] = None, p_remove: Optional[float] = None, block: int = 4) -> int:
    """rows per sample of the compact patch embed (_EmbedFn) on a g^3 token grid: the kept count rounded up to 64-row granules when the mask is known (eager
    forward), or -- for a captured step, whose shapes are fixed before the masks are drawn -- the mean kept count of the block-Bernoulli mask
    (swin_mae3d.py:1366-1373) plus eight standard deviations (GraphedTrainStep checks every mask against it and re-captures with full capacity should one
    ever exceed it)"""
    tps = g ** 3
    if kept is None:
        nb = max(0, (g - block) // block + 1) ** 3
        keep = 1.0 - float(p_remove)
        kept = int(math.ceil(block ** 3 * (nb * keep + 8.0 * math.sqrt(max(nb * keep * (1.0 - keep), 0.0)) + 1.0))) + (tps - nb * block ** 3)
    return int(min(tps, max(64, -(-kept // 64) * 64)))


def draw_block_bits(g: Sequence[int], p_remove: float, block: int = 4, rng=random) -> np.ndarray:
    """the draws of `draw_block_mask` only: uint8 array (nb0, nb1, nb2), 1 = block removed (same RNG stream, same order)"""
    nb = [max(0, (gi - block) // block + 1) for gi in g]
    return np.fromiter((rng.random() < p_remove for _ in range(nb[0] * nb[1] * nb[2])), dtype=np.uint8, count=nb[0] * nb[1] * nb[2]).reshape(nb)


def block_bits_of_mask(m, block: int = 4):
    """inverse of the block fill: (g,g,g) {0,1} mask -> its (nb,nb,nb) block bits, or None if the mask is not block-structured on a cube"""
    m = np.asarray(m, dtype=np.uint8)
    if m.ndim != 3 or len(set(m.shape)) != 1:
        return None
    nb = max(0, (m.shape[0] - block) // block + 1)
    bits = m[::block, ::block, ::block][:nb, :nb, :nb]
    full = np.zeros_like(m)
    if nb:
        full[:nb * block, :nb * block, :nb * block] = bits.repeat(block, 0).repeat(block, 1).repeat(block, 2)
    return bits if np.array_equal(full, m) else None


def draw_block_mask(g: Sequence[int], p_remove: float, block: int = 4, rng=random) -> Tensor:
    """swin_mae3d.py:1366-1373: one python-`random` draw per 4x4x4-token block in raster order -> uint8 (g0,g1,g2), 1 = removed.
    The draws keep the reference's order (same RNG stream, same pattern); only the block fill is vectorised (the per-block tensor
    slicing of a literal restatement costs 20-60 ms of host time per step and caps small-batch throughput)."""
    nb = [max(0, (gi - block) // block + 1) for gi in g]
    bits = np.fromiter((rng.random() < p_remove for _ in range(nb[0] * nb[1] * nb[2])), dtype=np.uint8, count=nb[0] * nb[1] * nb[2])
    m = np.zeros(tuple(g), dtype=np.uint8)
    if bits.size:
        blk = bits.reshape(nb).repeat(block, 0).repeat(block, 1).repeat(block, 2)
        m[:blk.shape[0], :blk.shape[1], :blk.shape[2]] = blk
    return torch.from_numpy(m)


# --------------------------------------------------------------------------------------------------
# packed compute-dtype weights
# --------------------------------------------------------------------------------------------------
class _Packer:
    """Owns the flat compute-dtype buffer holding every GEMM operand layout derived from the fp32 master
    parameters, and the device descriptor table for the single-launch pack kernel."""

    CAST, TRANS, CONV_F, CONV_D, CONVT_F, CONVT_D, C48_F, C48_D, C64_F, C64_D, PAD_ROWS, PAD_ROWS_T, PAD_CIN8 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12
    C48_NUMEL = 41 * 3 * 64 * 8  # conv48.hip fragment order [step][ntile][lane][8]

    def __init__(self):
        self.items = []  # (key, param, mode, dims, numel)
        self.views = {}
        self.cconv = None   # (decoder1 UpBlock3D, composed weight buffer, border table) when the composed forward applies
        self.swin_early = self.swin_late = None   # ctypes item arrays of the fused Swin-block weight streams (csrc/swin_block.hip), stages < / >= LATE_STAGE

    def add(self, key: str, p: nn.Parameter, mode: int):
        sh = tuple(p.shape)
        if mode in (self.CAST,):
            dims = (0, 0, 0)
        elif mode == self.TRANS:
            dims = (sh[0], int(np.prod(sh[1:])), 0)
        elif mode in (self.CONV_F, self.CONV_D, self.C48_F, self.C48_D, self.C64_F, self.C64_D):
            dims = (sh[0], sh[1], 27)
        else:
            dims = (sh[0], sh[1], int(np.prod(sh[2:])))
        numel = (self.C48_NUMEL * (sh[0] // 48) * (sh[1] // 48) if mode in (self.C48_F, self.C48_D)
                 else ops.conv64_pack_numel(sh[1], sh[0]) if mode in (self.C64_F, self.C64_D) else p.numel())
        if mode in (self.PAD_ROWS, self.PAD_ROWS_T):     # [Co][C] -> [Cop][C] / [C][Cop], Cop = Co rounded up to 8 (GEMM N granule)
            cop = (sh[0] + 7) // 8 * 8
            dims, numel = (sh[0], int(np.prod(sh[1:])), cop), cop * int(np.prod(sh[1:]))
        elif mode == self.PAD_CIN8:                      # conv weights [Co][Ci <= 8][taps] -> [Co][taps][8]
            dims, numel = (sh[0], sh[1], int(np.prod(sh[2:]))), sh[0] * int(np.prod(sh[2:])) * 8
        self.items.append((key, p, mode, dims, numel))

    def build(self, dtype: torch.dtype, device):
        total = sum((n + 63) // 64 * 64 for *_, n in self.items)
        self.buf = torch.empty(total, dtype=dtype, device=device)
        esz = self.buf.element_size()
        descs, blk2desc, blkstart = b"", [], []
        off = 0
        self.split = None   # first block of the decoder packs (they can be produced while the encoder runs)
        self.late = None    # first block of the encoder packs of stages >= LATE_STAGE (96 % of the encoder's elements; produced while stages 0/1 run)
        self._late_ev, self._late_pending = None, False
        for i, (key, p, mode, dims, n) in enumerate(self.items):
            if self.split is None and key.startswith("decoder"):
                self.split = len(blk2desc)
            if self.late is None and self.split is None and key.startswith(f"s{self.LATE_STAGE}."):
                self.late = len(blk2desc)
            self.views[key] = self.buf[off:off + n]
            descs += struct.pack("<QQiiiiq", p.data_ptr(), self.buf.data_ptr() + off * esz, mode, dims[0], dims[1], dims[2], n)
            if mode in (self.TRANS, self.CONV_F, self.CONV_D):   # tiled modes: blkstart = block id (nmh_pack_weights)
                nblk = {self.TRANS: ((dims[0] + 31) // 32) * ((dims[1] + 31) // 32), self.CONV_F: dims[0] * ((dims[1] + 31) // 32),
                        self.CONV_D: dims[1] * ((dims[0] + 31) // 32)}[mode]
                for k in range(nblk):
                    blk2desc.append(i)
                    blkstart.append(k)
            else:
                for s in range(0, n, 1024):
                    blk2desc.append(i)
                    blkstart.append(s)
            off += (n + 63) // 64 * 64
        self.descs = torch.frombuffer(bytearray(descs), dtype=torch.uint8).to(device)
        self.blk2desc = torch.tensor(blk2desc, dtype=torch.int32, device=device)
        self.blkstart = torch.tensor(blkstart, dtype=torch.int64, device=device)
        self.dt = ops.BF16 if dtype == torch.bfloat16 else ops.F32

    LATE_STAGE = 2

    def run(self, late_split: bool = True):
        """encoder layouts on the current stream; the decoder layouts (60 % of the elements, the stride-27 conv gathers) on the
        forked side stream, where they overlap the latency-bound encoder -- `join()` before the decoder reads them.  late_split: the
        encoder layouts of stages >= LATE_STAGE go to the side stream as well (first in its order), and `wait_late()` -- called by the
        model before such a stage runs -- makes the current stream wait for just that launch: the step starts 0.15 ms earlier at every
        batch size.  (False for callers that capture the stages into separate graphs: an event does not cross captures.)"""
        n = self.blk2desc.numel()
        side = ops.side_stream.enabled and os.environ.get("NMH_PACK_SPLIT", "1") != "0"
        sp = self.split if (self.split and side) else n
        lt = self.late if (self.late and side and late_split and sp < n and os.environ.get("NMH_PACK_LATE", "1") != "0") else sp
        self._late_pending = False
        if lt > 0:
            ops.pack_weights(self.dt, self.descs, self.blk2desc[:lt], self.blkstart[:lt], lt)
        if self.swin_early is not None:
            ops.swin_pack(self.swin_early)
        if self.swin_late is not None and not lt < sp:
            ops.swin_pack(self.swin_late)
        with ops.side_stream(enable=sp < n):
            if lt < sp:
                ops.pack_weights(self.dt, self.descs, self.blk2desc[lt:sp], self.blkstart[lt:sp], sp - lt)
                if self.swin_late is not None:
                    ops.swin_pack(self.swin_late)
                if self._late_ev is None:
                    self._late_ev = torch.cuda.Event()
                self._late_ev.record(torch.cuda.current_stream())
                self._late_pending = True
            if sp < n:
                ops.pack_weights(self.dt, self.descs, self.blk2desc[sp:], self.blkstart[sp:], n - sp)
            if self.cconv is not None:    # decoder1: ConvTranspose o conv1 composed weights (csrc/cconv.hip), from the fp32 masters
                up, Wcp, delta, ws, Wup, Wdp, mtab, Wres = self.cconv
                if mtab is not None:   # + the table the centered forward's mean comes from
                    ops.cconv_pack_centered(up.transp_conv.weight, up.conv_block.conv1.weight, up.transp_conv.bias, Wcp, delta, ws, mtab)
                else:
                    ops.cconv_pack(up.transp_conv.weight, up.conv_block.conv1.weight, up.transp_conv.bias, Wcp, delta, ws)
                if Wup is not None:
                    ops.upconv4_pack(ws, Wup)
                if Wres is not None:   # the same weights as the fragments of the tail that forms the residual itself
                    ops.tail_residual_pack(ws, Wres)
                if Wdp is not None:
                    ops.cconv_dgrad_pack(Wcp, Wdp)

    @staticmethod
    def join():
        ops.join_side()

    def wait_late(self):
        if self._late_pending:
            torch.cuda.current_stream().wait_event(self._late_ev)
            self._late_pending = False

    def __getitem__(self, key):
        return self.views[key]


def _gradbuf(p: nn.Parameter) -> Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


# --------------------------------------------------------------------------------------------------
# autograd Functions (one per block; explicit backward launching the HIP dgrad/wgrad kernels)
# --------------------------------------------------------------------------------------------------
class _EmbedFn(torch.autograd.Function):
    """patch conv (im2row + GEMM) -> LayerNorm -> + pos_embed -> masked tokens <- mask_token (swin_mae3d.py:1455-1463).
    With a mask (training) the embedding of a removed token is never read -- it is replaced by mask_token -- so the im2row, the GEMM rows, the LayerNorm rows and
    the weight-gradient contraction run on the KEPT tokens only, in compact rows [B][cap][.] (ops.EMBED_KEPT; cap = model._embed_cap rows per sample, sized by
    the caller that drew the mask).  Same values for every token that is read: the gradients differ from the full pass only by the order of fp32 sums."""

    @staticmethod
    def forward(ctx, anchor, mod, xb, mask_dev):
        m: "SwinTransformer_MAE3D_New" = mod
        m._wq.reset()   # first op of every forward pass: drops whatever a failed backward left queued
        B, R = xb.shape[0], xb.shape[2]
        g = R // 4
        tps = g ** 3
        T, C, dtype = B * tps, m.embed_dim, m.compute_dtype
        use_mask = mask_dev is not None
        kept = use_mask and ops.EMBED_KEPT and m._add_pos
        cap = min(tps, int(m._embed_cap)) if (kept and m._embed_cap) else tps
        rowmap = ops.embed_kept_rows(mask_dev, cap) if kept else None
        rows = B * cap if kept else T
        A = torch.empty((rows, 256), dtype=dtype, device=xb.device)
        if kept:
            ops.patch_embed_gather_kept(xb, A, B, R, rowmap, cap)
        else:
            ops.patch_embed_gather(xb, A, B, R)
        conv, ln = m.patch_partition[0], m.patch_partition[2]
        y0 = ops.gemm_nt(A, m._pk["pe.w"].view(C, 256), bias=conv.bias)
        tok = torch.empty((T, C), dtype=dtype, device=xb.device)
        mean, rstd = torch.empty(T, device=xb.device), torch.empty(T, device=xb.device)
        if kept:
            ops.embed_norm_fwd_kept(y0, ln.weight, ln.bias, tok, mean, rstd, T, C, m.pos_embed.view(-1, C), mask_dev, m.mask_token, tps, rowmap, cap)
        else:
            ops.layernorm_fwd(y0, ln.weight, ln.bias, tok, mean, rstd, T, C, pos=m.pos_embed.view(-1, C) if m._add_pos else None,
                              mask=mask_dev, mask_token=m.mask_token if use_mask else None, tokens_per_sample=tps)
        ctx.m, ctx.saved, ctx.dims = m, (A, y0, mean, rstd, mask_dev, rowmap), (T, C, g, cap)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        m = ctx.m
        A, y0, mean, rstd, mask_dev, rowmap = ctx.saved
        T, C, g, cap = ctx.dims
        conv, ln = m.patch_partition[0], m.patch_partition[2]
        dy0 = torch.empty_like(y0)
        if rowmap is not None:
            ops.embed_norm_bwd_kept(dtok.contiguous(), y0, ln.weight, mean, rstd, dy0, _gradbuf(ln.weight), _gradbuf(ln.bias), T, C, mask_dev, _gradbuf(m.mask_token),
                                    g ** 3, rowmap, cap)
        else:
            ops.layernorm_bwd(dtok.contiguous(), y0, ln.weight, mean, rstd, dy0, _gradbuf(ln.weight), _gradbuf(ln.bias), T, C,
                              mask=mask_dev, dmask_token=_gradbuf(m.mask_token) if mask_dev is not None else None, tokens_per_sample=g ** 3)
        q = m._wq if (ops.GROUPED_WGRAD and dy0.dtype == torch.bfloat16) else None
        if q is not None:
            q.add(dy0, A, _gradbuf(conv.weight).view(C, 256), dbias=_gradbuf(conv.bias), rows_per_sample=cap if rowmap is not None else g ** 3)
            q.flush(foreground=True)
            q.join()      # the backward pass ends here: every deferred weight gradient has been issued and joined
        else:
            ops.gemm_tn(dy0, A, _gradbuf(conv.weight), dbias=_gradbuf(conv.bias))
        return None, None, None, None


class _BlockFn(torch.autograd.Function):
    """x + SD(attn(LN1 x)); x + SD(MLP(LN2 x))  (swin_mae3d.py:366-369)."""

    @staticmethod
    def forward(ctx, x, blk, geom, sd1, sd2):
        b: "SwinBlock3D" = blk
        pk, key = b._pk, b._key
        T, C, heads = geom.tokens, b.dim, b.num_heads
        dev, dtype = x.device, x.dtype
        tps = T // geom.B
        sw = getattr(b, "_sw", None)
        # token-ordered backward (padded stages: 10^3 tokens in 12^3 window rows): window order stays inside the attention kernels, the Linear layers of the
        # attention branch see the real tokens only in the backward pass -- the forward then saves LN1(x) and o in token order
        # (the fused kernels are built for heads = C / 32 and a hidden width of 4 C: anything else keeps the unfused chain)
        swin_shape = heads * 32 == C and b.mlp[0].out_features == 4 * C
        fused_attn = sw is not None and swin_shape and ops.swin_attn_ok(x, C, geom)
        # the dispatch decisions are taken ONCE, here, and kept on ctx: the layout of the saved tensors depends on them, and the thresholds / switches they
        # are derived from are module globals that a test (or a caller) may change between forward and backward
        ctx.tok_bwd = bool(ops.TOKEN_BWD and geom.rows != geom.tokens and ctx.needs_input_grad[0] and not ops.mlp_fused_ok(x, C, T)
                           and (fused_attn or ops.TOKEN_BWD_UNFUSED))
        if fused_attn:
            # LN1 -> QKV -> window attention -> proj -> row scale -> + residual in ONE launch (csrc/swin_block.hip); saves the same tensors
            x1, xnw, mean1, rstd1, qkv, o, lse = ops.swin_attn_fwd(x, b.norm1.weight, b.norm1.bias, sw[ops.SWIN_ATTN_FWD], b.attn.qkv.bias,
                                                                   b.attn.relative_position_bias_table, b.attn.proj.bias, geom, rowscale=sd1, rows_per_scale=tps,
                                                                   token_saves=ctx.tok_bwd)
        elif ctx.tok_bwd:
            # unfused chain with token-ordered saves: LN1 writes its window-ordered output (operand of the qkv product: pad tokens are keys / values) and a
            # token-ordered copy (operand of the qkv weight gradient); the attention core scatters o to token order, proj is a plain T-row GEMM
            xw = torch.empty((geom.rows, C), dtype=dtype, device=dev)
            xnw = torch.empty((T, C), dtype=dtype, device=dev)
            mean1, rstd1 = torch.empty(T, device=dev), torch.empty(T, device=dev)
            ops.layernorm_fwd_window_tokens(x, b.norm1.weight, b.norm1.bias, xw, xnw, mean1, rstd1, C, geom)
            qkv = ops.gemm_nt(xw, pk[key + "qkv.w"].view(3 * C, C), bias=b.attn.qkv.bias)
            o = torch.empty((T, C), dtype=dtype, device=dev)
            lse = torch.empty(geom.rows * heads, device=dev)
            ops.window_attn_fwd_tokens(qkv, b.attn.relative_position_bias_table, o, lse, heads, C, geom)
            x1 = ops.gemm_nt(o, pk[key + "proj.w"].view(C, C), bias=b.attn.proj.bias, resid=x, rowscale=sd1, rows_per_scale=tps)
        else:
            xnw = torch.empty((geom.rows, C), dtype=dtype, device=dev)
            mean1, rstd1 = torch.empty(T, device=dev), torch.empty(T, device=dev)
            ops.layernorm_fwd(x, b.norm1.weight, b.norm1.bias, xnw, mean1, rstd1, geom.rows, C, src_mode=1, geom=geom)
            qkv = ops.gemm_nt(xnw, pk[key + "qkv.w"].view(3 * C, C), bias=b.attn.qkv.bias)
            o = torch.empty((geom.rows, C), dtype=dtype, device=dev)
            lse = torch.empty(geom.rows * heads, device=dev)
            ops.window_attn_fwd(qkv, b.attn.relative_position_bias_table, o, lse, heads, C, geom)
            x1 = torch.empty_like(x)   # x1 = x + sd1 * window_reverse(proj(o)): the reverse + residual are the GEMM's store
            ops.gemm_nt_window_scatter(o, pk[key + "proj.w"].view(C, C), x1, x, b.attn.proj.bias, sd1, tps, geom)
        ctx.mlp_fused = ops.mlp_fused_ok(x, C, T)
        if not ctx.mlp_fused and sw is not None and swin_shape and ops.swin_mlp_ok(x, C, T):
            # LN2 -> fc1 -> GELU -> fc2 -> row scale -> + residual in one launch; keeps what the (unfused) backward reads, gelu(hp) included
            x2, x1n, h_pre, mean2, rstd2, h_act = ops.swin_mlp_fwd(x1, b.norm2.weight, b.norm2.bias, sw[ops.SWIN_MLP_FWD], b.mlp[0].bias, b.mlp[3].bias,
                                                                  rowscale=sd2, rows_per_scale=tps, want_hact=True)
        elif ctx.mlp_fused:
            # LN2 -> fc1 -> GELU -> fc2 -> row scale -> + residual in one launch; nothing but x1 is kept for the backward (csrc/mlp_fused.hip)
            x2 = ops.mlp_fused_fwd(x1, b.norm2.weight, b.norm2.bias, pk[key + "fc1.w"].view(4 * C, C), b.mlp[0].bias, pk[key + "fc2.wT"].view(4 * C, C),
                                   b.mlp[3].bias, rowscale=sd2, rows_per_scale=tps)
            x1n = mean2 = rstd2 = h_pre = h_act = None
        else:
            x1n = torch.empty_like(x)
            mean2, rstd2 = torch.empty(T, device=dev), torch.empty(T, device=dev)
            ops.layernorm_fwd(x1, b.norm2.weight, b.norm2.bias, x1n, mean2, rstd2, T, C)
            h_pre = torch.empty((T, 4 * C), dtype=dtype, device=dev)
            h_act = ops.gemm_nt(x1n, pk[key + "fc1.w"].view(4 * C, C), bias=b.mlp[0].bias, act=1, C2=h_pre)
            x2 = ops.gemm_nt(h_act, pk[key + "fc2.w"].view(C, 4 * C), bias=b.mlp[3].bias, resid=x1, rowscale=sd2, rows_per_scale=tps)
        ctx.b, ctx.geom = b, geom
        ctx.saved = (x, xnw, mean1, rstd1, qkv, o, lse, x1, x1n, mean2, rstd2, h_pre, h_act, sd1, sd2)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        b, geom = ctx.b, ctx.geom
        x, xnw, mean1, rstd1, qkv, o, lse, x1, x1n, mean2, rstd2, h_pre, h_act, sd1, sd2 = ctx.saved
        pk, key = b._pk, b._key
        T, C, heads = geom.tokens, b.dim, b.num_heads
        tps = T // geom.B
        dx2 = dx2.contiguous()
        # weight/bias gradients are off the critical path.  bf16: they are queued (the operands stay referenced by the queue) and issued
        # per stage as grouped launches (ops.WgradQueue); fp32 parity mode: one launch each on the forked side stream.
        q = b._wq if (ops.GROUPED_WGRAD and dx2.dtype == torch.bfloat16) else None

        def wgrad(A, Bm, lin, rowscale=None, rps=tps):
            if q is not None:
                q.add(A, Bm, _gradbuf(lin.weight), dbias=_gradbuf(lin.bias), rowscale=rowscale, rows_per_sample=rps)
            else:
                with ops.side_stream(enable=T >= ops.side_stream.min_rows):
                    ops.gemm_tn(A, Bm, _gradbuf(lin.weight), rowscale=rowscale, rows_per_scale=rps, dbias=_gradbuf(lin.bias))
        # ---- MLP branch
        tok_bwd = ctx.tok_bwd
        # = sd1 * dx1 in window order (adjoint of the window reverse), second output of the LN backward; not formed by the token-ordered backward
        dyw = None if tok_bwd else torch.empty_like(xnw)
        if ctx.mlp_fused:
            # one launch: recomputes LN2 / the hidden activations, writes the operands of the two weight gradients and dx1 (+ its window-ordered copy)
            dx1, x1n, h_act, dh = ops.mlp_fused_bwd(x1, dx2, b.norm2.weight, b.norm2.bias, pk[key + "fc1.w"].view(4 * C, C), b.mlp[0].bias,
                                                    pk[key + "fc2.wT"].view(4 * C, C), _gradbuf(b.norm2.weight), _gradbuf(b.norm2.bias),
                                                    rowscale=sd2, rows_per_scale=tps, dyw=dyw, dyw_scale=sd1, geom=geom)
            if q is not None:
                q.flush_due()   # the previous stage's weight gradients fork off BEHIND this stage's first input-gradient kernel (ops.WgradQueue.request_flush)
            wgrad(dx2, h_act, b.mlp[3], rowscale=sd2)
            wgrad(dh, x1n, b.mlp[0])
            if q is not None and ops.EARLY_MLP_WGRAD:
                # (NMH_EARLY_MLP_WGRAD=1, off by default) every block that takes the fused MLP backward -- with the default MLP_FUSED_WIDTHS = {96} that is
                # stage 0, the end of the backward pass: the MLP pair's gradients -- 10 of the block's 16 operand passes -- start under the block's own
                # attention-branch chain instead of waiting for the stage flush.  The flush is a background flush of EVERYTHING queued so far (LayerNorm /
                # pad items included): it splits the stage's grouped launches per block
                q.flush()
        else:
            dh = ops.gemm_nt(dx2, pk[key + "fc2.wT"].view(4 * C, C), act=2, C2=h_pre, rowscale=sd2, rows_per_scale=tps)
            if q is not None:
                q.flush_due()
            wgrad(dx2, h_act, b.mlp[3], rowscale=sd2)
            dx1n = ops.gemm_nt(dh, pk[key + "fc1.wT"].view(C, 4 * C))
            wgrad(dh, x1n, b.mlp[0])
            dx1 = torch.empty_like(x)
            ops.layernorm_bwd(dx1n, x1, b.norm2.weight, mean2, rstd2, dx1, _gradbuf(b.norm2.weight), _gradbuf(b.norm2.bias), T, C, dres=dx2,
                              geom=None if tok_bwd else geom, tokens_per_sample=tps, dyw=dyw, dyw_scale=None if tok_bwd else sd1, wq=q)   # (dgamma / dbeta: partial sums now, reduced with the stage's weight gradients)
        # ---- attention branch
        if tok_bwd:
            # xnw / o hold LN1(x) / the attention output in TOKEN order here; every GEMM below has T rows
            dtab = _gradbuf(b.attn.relative_position_bias_table)
            do = ops.gemm_nt(dx1, pk[key + "proj.wT"].view(C, C), rowscale=sd1, rows_per_scale=tps)           # gradient of the attention output, token order
            wgrad(dx1, o, b.attn.proj, rowscale=sd1)
            dqkv = torch.empty((T, 3 * C), dtype=x.dtype, device=x.device)
            dq_pad = torch.empty_like(qkv)                                                                     # only its pad rows are written (and read)
            ops.window_attn_bwd_tokens(qkv, b.attn.relative_position_bias_table, do, lse, dqkv, dq_pad, dtab, heads, C, geom)
            dxn = ops.gemm_nt(dqkv, pk[key + "qkv.wT"].view(C, 3 * C))
            wgrad(dqkv, xnw, b.attn.qkv)
            g_qb = _gradbuf(b.attn.qkv.bias)
            if q is not None:   # the pad rows' share of the qkv bias gradient, with the stage's weight gradients (one launch for all blocks of the flush)
                q.add_pad_colsum(dq_pad, g_qb, geom)
            else:
                with ops.side_stream():
                    ops.window_pad_rows_colsum(dq_pad, g_qb, geom)
            dx = torch.empty_like(x)
            ops.layernorm_bwd(dxn, x, b.norm1.weight, mean1, rstd1, dx, _gradbuf(b.norm1.weight), _gradbuf(b.norm1.bias), T, C, dres=dx1, wq=q)
        else:
            do = ops.gemm_nt(dyw, pk[key + "proj.wT"].view(C, C))
            wgrad(dyw, o, b.attn.proj, rps=geom.rows // geom.B)
            dqkv = torch.empty_like(qkv)
            ops.window_attn_bwd(qkv, b.attn.relative_position_bias_table, do, lse, dqkv, _gradbuf(b.attn.relative_position_bias_table), heads, C, geom)
        if not tok_bwd:
            dxnw = ops.gemm_nt(dqkv, pk[key + "qkv.wT"].view(C, 3 * C))
            wgrad(dqkv, xnw, b.attn.qkv, rps=geom.rows // geom.B)
            dx = torch.empty_like(x)
            ops.layernorm_bwd(dxnw, x, b.norm1.weight, mean1, rstd1, dx, _gradbuf(b.norm1.weight), _gradbuf(b.norm1.bias), T, C, src_mode=1, geom=geom, dres=dx1, wq=q)
        # the block's own side-stream launches (fp32 parity mode) read temporaries of this block: join before they are released.  With the queue
        # (bf16) the operands stay referenced by it and NOTHING may be joined here: a join makes this block's successor wait for every weight
        # gradient the last flush put on the side stream -- the trace showed the two queues taking turns (>= 2 kernels in flight for 4 of 52 ms)
        if q is None or not ops.WQ_LATE_JOIN:
            ops.join_side()
        return dx, None, None, None, None


class _StageFlushFn(torch.autograd.Function):
    """identity at the input of an encoder stage; its backward runs when the stage's input-gradient chain is complete and issues the
    weight gradients queued by the stage's blocks (and its patch merging) as grouped launches"""

    @staticmethod
    def forward(ctx, x, wq, last=False):
        ctx.wq, ctx.last = wq, last    # last: the first stage -- its flush ends the backward pass and has the chip to itself
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # Joining the previous stage's launch here makes the next input-gradient kernel a successor of the side chain: the HIP graph executor then puts
        # the whole next main segment on the side chain's queue, BEHIND the weight gradients just flushed -- the trace showed the two queues alternating
        # (>= 2 kernels in flight for 4 of 52 ms).  With the late join the side chain only hangs off the main chain (fork edges), the operands stay
        # referenced by the queue until the end-of-backward join.
        if not ops.WQ_LATE_JOIN or ctx.wq.sync_after_flush:
            ctx.wq.join()     # the previous stage's launch (if any) has had a whole stage of input-gradient work to finish under
        if ops.WQ_FLUSH_AFTER_FIRST and not ctx.last and not ctx.wq.sync_after_flush:
            ctx.wq.request_flush()   # issued by the next block's backward behind its first kernel
        else:
            ctx.wq.flush(foreground=ctx.last)
        if ctx.wq.sync_after_flush:
            ctx.wq.join()
        return g, None, None


class _Fork2Fn(torch.autograd.Function):
    """a feature map with two consumers (the next encoder stage and a decoder skip connection / the FPN neck): identity forward, the two
    incoming gradients are summed by a HIP kernel (autograd's own accumulation would be an ATen add on the dependent chain)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return g1 if g2 is None else g2
        g1, g2 = g1.contiguous(), g2.contiguous()
        if g1.numel() % 8:
            return g1 + g2
        return ops.add(g1, g2)


class _MergeFn(torch.autograd.Function):
    """2x2x2 gather -> LayerNorm(8C) -> Linear(8C,2C,no bias)  (swin_mae3d.py:390-414)."""

    @staticmethod
    def forward(ctx, x, mod, geom):
        m: "PatchMerging3D" = mod
        Cin = m.dim
        H2, W2, D2 = (geom.H + 1) // 2, (geom.W + 1) // 2, (geom.D + 1) // 2
        rows = geom.B * H2 * W2 * D2
        xg = torch.empty((rows, 8 * Cin), dtype=x.dtype, device=x.device)
        mean, rstd = torch.empty(rows, device=x.device), torch.empty(rows, device=x.device)
        ops.layernorm_fwd(x, m.norm.weight, m.norm.bias, xg, mean, rstd, rows, 8 * Cin, src_mode=2, geom=geom)
        y = ops.gemm_nt(xg, m._pk[m._key + "red.w"].view(2 * Cin, 8 * Cin))
        ctx.m, ctx.geom, ctx.saved, ctx.rows = m, geom, (x, xg, mean, rstd), rows
        return y

    @staticmethod
    def backward(ctx, dy):
        m, geom, rows = ctx.m, ctx.geom, ctx.rows
        x, xg, mean, rstd = ctx.saved
        Cin = m.dim
        dy = dy.contiguous()
        dxg = ops.gemm_nt(dy, m._pk[m._key + "red.wT"].view(8 * Cin, 2 * Cin))
        if ops.GROUPED_WGRAD and dy.dtype == torch.bfloat16:
            m._wq.add(dy, xg, _gradbuf(m.reduction.weight), rows_per_sample=rows // geom.B)
        else:
            ops.gemm_tn(dy, xg, _gradbuf(m.reduction.weight))
        dx = torch.empty_like(x)
        ops.layernorm_bwd(dxg, x, m.norm.weight, mean, rstd, dx, _gradbuf(m.norm.weight), _gradbuf(m.norm.bias), rows, 8 * Cin, src_mode=2, geom=geom)
        return dx, None, None


class _UpBlockFn(torch.autograd.Function):
    """ConvTranspose3d(k=s) -> cat(skip) -> UnetResBlock (unetr_block.py:57-71,193-200), channels-last."""

    @staticmethod
    def forward(ctx, x, skip, mod, B, v, tail=None):
        """tail = (model, xb, extents, tokmask, pred_out): the block is the last decoder level and the 1x1 head + loss are
        evaluated here, so that backward can fuse the loss gradient with the backward of the last InstanceNorm; returns the
        loss triple instead of the feature map."""
        m: "UpBlock3D" = mod
        pk, key = m._pk, m._key
        k, Cin, Cout = m.k, m.cin, m.cout
        k3, V = k ** 3, (v * k) ** 3
        dev, dtype = x.device, x.dtype
        has_skip = skip is not None
        Cc = 2 * Cout if has_skip else Cout
        cc = pk.cconv if ((key + "c1.wk") in pk.views and pk.cconv is not None and pk.cconv[0] is m and v % 8 == 0 and not has_skip) else None
        # decoder1 in training (round 6): the residual u = ConvT(x) is read by the tail forward alone (the composed kernels take conv1's passes through the coarse
        # tensor, the tail backward reads a sign mask), and the tail forms it from x on the matrix cores -- u is never stored (csrc/norm.hip: tail_fwd_coarse_kernel)
        rx = (cc is not None and ops.TAIL_FROM_COARSE and cc[7] is not None and tail is not None and not m.has_proj and ctx.needs_input_grad[0]
              and tail[0].fuse_tail_sums and ops.TAIL_SIGN_MASK and ops.CCONV_WGRAD and v <= 40 and cc[5] is not None and k == 4 and (key + "c2.wk") in pk.views
              and ops.tail_from_coarse_ok(v * k, Cout, Cin, dtype))
        ctx.rx = rx
        cat = None if rx else torch.empty((B * V, Cc), dtype=dtype, device=dev)
        if rx:
            pass
        elif cc is not None and cc[4] is not None:   # decoder1: persistent kernel, coarse fragments in registers (csrc/cconv.hip)
            ops.upconv4_fwd(x.view(B, v, v, v, Cin), cc[4], m.transp_conv.bias, cat, B, v)
        else:
            ops.upconv_fwd(x, pk[key + "t.w"].view(k3 * Cout, Cin), m.transp_conv.bias, cat, B, v, k, Cin, Cout)   # pixel shuffle in the epilogue
        if has_skip:
            ops.copy_cols(skip.reshape(B * V, Cout), cat[:, Cout:])
        S = v * k
        new_acc = lambda: ops.acc_zeros((B, Cout, 2), dev)   # noqa: E731  (one slice per reduction: arena slices are single-use)
        scratch = new_acc()
        ctx.c64 = False
        c48 = (key + "c1.wk") in pk.views and (key + "c2.wk") in pk.views
        c48mb = (not c48 and (key + "c1.wkm") in pk.views and (key + "c2.wkm") in pk.views and S ** 3 >= ops.C48MB_MIN_VOXELS > 0
                 and S ** 3 * Cc * 2 < 2 ** 32)
        c64 = (not c48 and not c48mb and (key + "c1.w64") in pk.views and (key + "c2.w64") in pk.views and S ** 3 >= ops.C64_MIN_VOXELS > 0
               and S ** 3 * Cc * 2 < 2 ** 32)
        if c48:   # Cin = Cout = 48 in bf16: LDS-halo kernel with fragment-ordered weights ("c1.w" -> "c1.wk", "c1.wd" -> "c1.wkd")
            conv = lambda X, nm, co, **kw: ops.conv3d_k3_c48(X, pk[key + nm.replace(".w", ".wk")], **kw)  # noqa: E731
        elif c48mb:  # channel counts multiples of 48 at a volume that fills the chip (40^3, 96 / 192 channels): the same kernel on 48-channel blocks
            conv = lambda X, nm, co, **kw: ops.conv3d_k3_c48mb(X, pk[key + nm.replace(".w", ".wkm")], co, **kw)  # noqa: E731
        elif c64:  # channel counts multiples of 64 (swin_b): LDS-halo kernel on 64-channel blocks ("c1.w" -> "c1.w64", "c1.wd" -> "c1.w64d")
            conv = lambda X, nm, co, **kw: ops.conv3d_k3_c64(X, pk[key + nm.replace(".w", ".w64")], co, **kw)  # noqa: E731
        else:
            conv = lambda X, nm, co, **kw: ops.conv3d_k3(X, pk[key + nm], co, **kw)  # noqa: E731
        ctx.c64 = c64
        halo_stats = c48 or c64    # InstanceNorm statistics come out of the conv epilogue
        st1 = torch.empty((B, Cout, 2), device=dev)
        cc = cc if c48 else None
        # centered form (csrc/cconv.hip, round 6): the mean of conv1's output comes from the coarse tensor, conv1 stores z = lrelu(y1 - mean), 1 / std goes into conv2's
        # weights per sample; the backward reads z.  Needs the launches that read it (fused sums in conv2's input gradient, the background InstanceNorm backward, the
        # scaled weight-gradient reduce: B in {1, 2, 4, 8}) -- otherwise the classic form below
        cz = (cc is not None and ops.CCONV_CENTERED and cc[6] is not None and B in (1, 2, 4, 8) and S % 16 == 0 and ((S // 4) ** 2 * (S // 16)) % (8 // B) == 0
              and ops.C48_BWD_REDUCE and ops.INBWD_BG
              and ops.side_stream.enabled and ops.CCONV_WGRAD and v <= 40 and cc[5] is not None)
        ctx.cz = cz
        if cz:
            mhat = ops.cconv_output_mean(x.view(B, v, v, v, Cin), cc[6], cc[2], B, v)
            y1 = ops.cconv_fwd_centered(x.view(B, v, v, v, Cin), cc[1], cc[2], mhat, B, v, stats_acc=scratch).view(B * V, Cout)   # (y1 holds z)
            ops.instnorm_finalize(scratch, st1, B, V, Cout)    # (mean of y1 - mhat: ~ 0; rstd)
        elif cc is not None:   # decoder1: y1 straight from the coarse map with the composed weights (4x fewer FLOPs; cat is only the residual)
            y1 = ops.cconv_fwd(x.view(B, v, v, v, Cin), cc[1], cc[2], B, v, stats_acc=scratch).view(B * V, Cout)
            ops.instnorm_finalize(scratch, st1, B, V, Cout)
        elif halo_stats:   # InstanceNorm statistics come out of the conv epilogue (no extra pass over the 160^3 tensor)
            y1 = conv(cat.view(B, S, S, S, Cc), "c1.w", Cout, stats_acc=scratch).view(B * V, Cout)
            ops.instnorm_finalize(scratch, st1, B, V, Cout)
        else:
            y1 = conv(cat.view(B, S, S, S, Cc), "c1.w", Cout).view(B * V, Cout)
            ops.instnorm_stats(y1, st1, scratch, B, V, Cout)
        a1 = None
        if not cz:
            a1 = torch.empty_like(y1)
            ops.instnorm_apply(y1, st1, a1, B, V, Cout)
        st2 = torch.empty((B, Cout, 2), device=dev)
        scratch = new_acc()
        if cz:
            wk2 = torch.empty(B * ops.C48_IMG, dtype=dtype, device=dev)
            ops.conv48_pack_scaled(m.conv_block.conv2.weight, st1, wk2, B)
            y2 = ops.conv3d_k3_c48_per_sample(y1.view(B, S, S, S, Cout), wk2, stats_acc=scratch).view(B * V, Cout)
            ops.instnorm_finalize(scratch, st2, B, V, Cout)
        elif halo_stats:
            y2 = conv(a1.view(B, S, S, S, Cout), "c2.w", Cout, stats_acc=scratch).view(B * V, Cout)
            ops.instnorm_finalize(scratch, st2, B, V, Cout)
        else:
            y2 = conv(a1.view(B, S, S, S, Cout), "c2.w", Cout).view(B * V, Cout)
            ops.instnorm_stats(y2, st2, scratch, B, V, Cout)
        y3 = st3 = None
        fused_tail = tail is not None and not m.has_proj
        # fused tail: d0 = lrelu(IN(y2) + cat) is consumed inside the tail kernels and rebuilt from (y2, cat) in the backward -- never stored
        out = None if fused_tail else torch.empty_like(y2)
        if m.has_proj:
            y3 = ops.gemm_nt(cat, pk[key + "c3.w"].view(Cout, Cc))
            st3 = torch.empty((B, Cout, 2), device=dev)
            ops.instnorm_stats(y3, st3, new_acc(), B, V, Cout)
            ops.instnorm_apply(y2, st2, out, B, V, Cout, r=y3, stats_r=st3, rmode=2)
        elif not fused_tail:
            ops.instnorm_apply(y2, st2, out, B, V, Cout, r=cat, rmode=1)
        ctx.m, ctx.dims, ctx.conv, ctx.c48 = m, (B, v, has_skip), conv, c48
        ctx.cc = cc is not None and ops.CCONV_WGRAD and v <= 40
        ctx.saved = (x, cat, y1, st1, a1, y2, st2, y3, st3, out)
        ctx.tail = None
        if tail is not None:
            assert not m.has_proj
            model, xb, extents, tokmask, pred_out = tail
            lsums = torch.empty(8, dtype=torch.float64, device=dev)
            losses = torch.empty(3, device=dev)
            dpred = torch.empty((B * V, 4), device=dev) if ctx.needs_input_grad[0] else None
            # the reductions of the tail BACKWARD (InstanceNorm-backward sums, head weight gradient) are taken by the same pass
            bsum = torch.empty(B * Cout * 4 + 4 * Cout, dtype=torch.float64, device=dev) if (dpred is not None and model.fuse_tail_sums) else None
            # last InstanceNorm + residual + LeakyReLU, the 1x1 head and the loss terms in one pass over (y2, cat)
            # with the backward's sums taken here, the sign of d0 is all the tail backward still needs of it: 6 bytes per voxel instead of the residual row
            smask = (torch.empty((B * V, 8), dtype=torch.uint8, device=dev)
                     if (bsum is not None and ops.TAIL_SIGN_MASK and dtype == torch.bfloat16 and Cout == 48 and out is None and S % 4 == 0) else None)
            if rx:
                assert smask is not None and bsum is not None and dpred is not None
                ops.mae_tail_fwd_from_coarse(y2, st2, x.view(B, v, v, v, Cin), cc[7], m.transp_conv.bias, model.out.conv.weight, model.out.conv.bias, xb, extents, tokmask,
                                             B, S, Cout, lsums, losses, dpred, bsum, smask, pred=pred_out)
            else:
                ops.mae_tail_fwd(y2, st2, cat, out, model.out.conv.weight, model.out.conv.bias, xb, extents, tokmask, B, S, Cout, lsums, losses,
                                 pred_out, dpred, bwd_sums=bsum, sign_mask=smask)
            ctx.tail = (model, lsums, dpred, bsum, smask)
            return losses
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.m
        B, v, has_skip = ctx.dims
        x, cat, y1, st1, a1, y2, st2, y3, st3, out = ctx.saved
        pk, key = m._pk, m._key
        k, Cin, Cout = m.k, m.cin, m.cout
        k3, S = k ** 3, v * k
        V = S ** 3
        Cc = cat.shape[1] if cat is not None else Cout
        dev, dtype = x.device, x.dtype
        sums2 = ops.acc_zeros((B, Cout, 2), dev)
        dy2 = torch.empty_like(y2)
        dcat = torch.empty((B * V, Cc), dtype=dtype, device=dev)
        if ctx.tail is not None:   # d(loss)/d(losses[0]) == 1 (the reference calls loss.backward()); d(d0) is never materialised
            model, lsums, dpred, bsum, smask = ctx.tail
            ops.mae_tail_bwd(None, y2, st2, dpred, lsums, model.out.conv.weight, sums2, dy2, dcat,
                             _gradbuf(model.out.conv.weight), _gradbuf(model.out.conv.bias), B, V, Cout, r=None if smask is not None else cat, bwd_sums=bsum,
                             sign_mask=smask)
        elif m.has_proj:
            dout = dout.contiguous()
            sums3 = ops.acc_zeros((B, Cout, 2), dev)
            dy3 = torch.empty_like(y3)
            ops.instnorm_bwd_reduce(dout, out, y2, st2, sums2, B, V, Cout, r=y3, stats_r=st3, sums_r=sums3, rmode=2)
            ops.instnorm_bwd_apply(dout, out, y2, st2, sums2, dy2, B, V, Cout, r=y3, stats_r=st3, sums_r=sums3, rmode=2, dr=dy3)
        else:
            dout = dout.contiguous()
            ops.instnorm_bwd_reduce(dout, out, y2, st2, sums2, B, V, Cout, rmode=1)
            ops.instnorm_bwd_apply(dout, out, y2, st2, sums2, dy2, B, V, Cout, rmode=1, dr=dcat)  # dcat <- g (plain residual)
        conv = ctx.conv
        wgrad = ops.conv3d_k3_c48_wgrad if ctx.c48 else ops.conv3d_k3_wgrad
        sums1 = ops.acc_zeros((B, Cout, 2), dev)
        fused_red = ctx.c48 and ops.C48_BWD_REDUCE and S % 16 == 0
        cz = getattr(ctx, "cz", False)
        if cz:          # centered form: y1 holds z = lrelu(conv1's output - mean)
            da1 = ops.conv3d_k3_c48_bwd_reduce_centered(dy2.view(B, S, S, S, Cout), pk[key + "c2.wkd"], y1, st1, sums1).view(B * V, Cout)
        elif fused_red:   # the InstanceNorm-backward sums of (da1, y1) come out of the conv epilogue (no separate pass over both 160^3 tensors)
            da1 = ops.conv3d_k3_c48_bwd_reduce(dy2.view(B, S, S, S, Cout), pk[key + "c2.wkd"], y1, st1, sums1).view(B * V, Cout)
        else:
            da1 = conv(dy2.view(B, S, S, S, Cout), "c2.wd", Cout).view(B * V, Cout)
        # Weight gradients run on the forked side stream -- except the persistent 160^3 kernels, which own every CU: overlapping
        # them with the next MFMA kernel OR with the HBM-bound InstanceNorm passes AS THEY ARE (100 k workgroups of 120 VGPRs) measured slower
        # (51.2 vs 50.4 ms at 4 grids, 34.8 vs 30.8 ms at 1), so they stay on the main stream.  Round 5: the pass as ONE small workgroup per CU that
        # fits beside the weight gradient's (`bg` below) does overlap.
        # small-level weight gradients: forked to the side stream -- or (NMH_DEFER_DEC) queued on the encoder's weight-gradient queue and issued
        # with its next flush, one fork for all of them
        defer = ops.DEFER_DECODER_WGRAD and getattr(m, "_wq", None) is not None and not (ctx.c48 or ctx.c64)

        def side(fn):
            if defer and ops.DEC_WGRAD_NOW:
                # issued right away on the forked side stream (no join before the end of the backward pass): the small decoder levels' input-gradient chain
                # is a string of latency-bound launches that leaves most of the chip idle, the side stream has nothing else to do there
                m._wq.launch_now(fn)
            elif defer:
                m._wq.defer(fn)
            else:
                with ops.side_stream(enable=not (ctx.c48 or ctx.c64)):
                    fn()
        g_c2, g_c1 = _gradbuf(m.conv_block.conv2.weight), _gradbuf(m.conv_block.conv1.weight)
        dy1 = torch.empty_like(dy2)  # (dy2 is still being read by the side-stream wgrad)
        bg = cz or (ops.INBWD_BG and ctx.c48 and fused_red and Cout == 48 and dtype == torch.bfloat16 and ops.side_stream.enabled)
        if cz:
            with ops.side_stream():
                ops.instnorm_bwd_apply_bg_centered(da1, y1, st1, sums1, dy1, B, V, Cout)
            ops.conv3d_k3_c48_wgrad_scaled(dy2.view(B, S, S, S, Cout), y1.view(B, S, S, S, Cout), st1, g_c2)
        elif bg:
            # decoder1: the HBM-bound InstanceNorm backward (needs conv2's input gradient only) streams on a forked stream UNDER conv2's weight gradient, as one
            # small workgroup per CU beside the persistent kernel's (csrc/norm.hip: in_bwd_apply_bg_kernel); joined in front of its first consumer
            # (the transpose conv's input gradient, a 4000-workgroup GEMM, does not ride along: one of its workgroups per CU at a time crawls -- 2.7 ms for 0.64)
            with ops.side_stream():
                ops.instnorm_bwd_apply_bg(da1, y1, st1, sums1, dy1, B, V, Cout)
        if not cz:
            side(lambda: wgrad(dy2.view(B, S, S, S, Cout), a1.view(B, S, S, S, Cout), g_c2))
        if not fused_red and not cz:
            ops.instnorm_bwd_reduce(da1, None, y1, st1, sums1, B, V, Cout, rmode=0)   # sign(a1) == sign(y1 - mean): a1 is not re-read
        if bg:
            ops.join_side()
        else:
            ops.instnorm_bwd_apply(da1, None, y1, st1, sums1, dy1, B, V, Cout, rmode=0)
        # decoder1 with the composed kernels: conv1's input gradient on the fine grid is never formed -- dx and the transpose conv's parameter gradients
        # take their conv1 part through the composition (cconv_dgrad below, cconv_wgrad's G blocks), dcat keeps the residual branch's gradient alone
        cdg = ctx.cc and pk.cconv[5] is not None
        if not cdg:
            conv(dy1.view(B, S, S, S, Cout), "c1.wd", Cc, out=dcat.view(B, S, S, S, Cc), accumulate=not m.has_proj)
        if ctx.cc:   # decoder1: conv1's weight gradient through the composition (a quarter of the FLOPs, contraction over the coarse cells; csrc/cconv.hip)
            # the persistent G kernel on the main stream; reduce / border sums / chain rules (0.5 ms of small launches that only feed weight gradients) behind
            # it on the side stream, or with the encoder's next weight-gradient flush
            cw = lambda ph: ops.cconv_wgrad(x.view(B, v, v, v, Cin), dy1.view(B, S, S, S, Cout), pk.cconv[3], m.transp_conv.bias, g_c1, B, v,  # noqa: E731
                                            dWt=_gradbuf(m.transp_conv.weight) if cdg else None, dbt=_gradbuf(m.transp_conv.bias) if cdg else None, phase=ph)
            if not ops.side_stream.enabled or not ops.CCONV_WGRAD_SPLIT:
                cw(0)
            else:
                cw(1)
                if ops.DEFER_DECODER_WGRAD and getattr(m, "_wq", None) is not None:
                    m._wq.defer(lambda: cw(2))
                else:
                    with ops.side_stream():
                        cw(2)
        else:
            side(lambda: wgrad(dy1.view(B, S, S, S, Cout), cat.view(B, S, S, S, Cc), g_c1))
        if m.has_proj:
            ops.gemm_nt(dy3, pk[key + "c3.wT"].view(Cc, Cout), out=dcat, accumulate=True)
            g_c3 = _gradbuf(m.conv_block.conv3.weight)
            if defer:
                (m._wq.launch_now if ops.DEC_WGRAD_NOW else m._wq.defer)(lambda: ops.gemm_tn(dy3, cat, g_c3))
            else:
                with ops.side_stream():
                    ops.gemm_tn(dy3, cat, g_c3)
        dskip = None
        if has_skip:
            dskip = torch.empty((B * V, Cout), dtype=dtype, device=dev)
            ops.copy_cols(dcat[:, Cout:], dskip)
        dx = torch.empty((B * v ** 3, Cin), dtype=dtype, device=dev)
        ops.upconv_dgrad(dcat, pk[key + "t.wd"].view(Cin, k3 * Cout), dx, B, v, k, Cin, Cout)   # reads dcat through the pixel shuffle
        if cdg:
            ops.cconv_dgrad(dy1, pk.cconv[5], B, v, add=dx, out=dx)
        g_tw, g_tb = _gradbuf(m.transp_conv.weight), _gradbuf(m.transp_conv.bias)
        if (ctx.c48 or ctx.c64) and ops.UPW_EARLY and ops.WQ_LATE_JOIN and ops.side_stream.enabled and getattr(m, "_wq", None) is not None:
            # decoder1 (the FIRST block of the backward pass): queued, its 1.5 ms would join the weight gradients of every later level in front of the
            # encoder's on the side stream, which is busy from the stage-3 flush to the end of the step; issued here it runs under the small decoder levels
            m._wq.launch_now(lambda: ops.upconv_wgrad(dcat, x, g_tw, g_tb, B, v, k, Cin, Cout))
        elif defer or (ops.DEFER_DECODER_WGRAD and getattr(m, "_wq", None) is not None):
            (m._wq.launch_now if ops.DEC_WGRAD_NOW else m._wq.defer)(lambda: ops.upconv_wgrad(dcat, x, g_tw, g_tb, B, v, k, Cin, Cout))
        else:
            with ops.side_stream():
                ops.upconv_wgrad(dcat, x, g_tw, g_tb, B, v, k, Cin, Cout)
            ops.join_side()
        return dx, dskip, None, None, None, None


class _LossFn(torch.autograd.Function):
    """UnetOutBlock 1x1 conv + forward_loss (swin_mae3d.py:1513-1549) fused; outputs (loss, loss_rgb, loss_alpha)."""

    @staticmethod
    def forward(ctx, d0, mod, xb, extents, tokmask, pred_out):
        m = mod
        B, R, Cd = xb.shape[0], xb.shape[2], d0.shape[1]
        sums = torch.empty(4, dtype=torch.float64, device=d0.device)
        losses = torch.empty(3, device=d0.device)
        ops.mae_loss_fwd(d0, m.out.conv.weight, m.out.conv.bias, xb, extents, tokmask, B, R, Cd, sums, losses, pred_out)
        ctx.m, ctx.saved, ctx.dims = m, (d0, xb, extents, tokmask, sums), (B, R, Cd)
        return losses

    @staticmethod
    def backward(ctx, dl):
        m = ctx.m
        d0, xb, extents, tokmask, sums = ctx.saved
        B, R, Cd = ctx.dims
        dd0 = torch.empty_like(d0)
        dp8 = torch.empty((d0.shape[0], 8), dtype=d0.dtype, device=d0.device)
        ops.mae_loss_bwd(d0, m.out.conv.weight, m.out.conv.bias, xb, extents, tokmask, B, R, Cd, sums, dd0, dp8,
                         _gradbuf(m.out.conv.weight), _gradbuf(m.out.conv.bias))
        return dd0, None, None, None, None, None


# --------------------------------------------------------------------------------------------------
# modules (parameter names/shapes == reference)
# --------------------------------------------------------------------------------------------------
class _Permute(nn.Module):  # placeholder for torchvision Permute (index 1 of patch_partition; no parameters)
    def __init__(self, dims):
        super().__init__()
        self.dims = dims


class PatchPartition(nn.Sequential):
    """Conv3d(4,C,k=4,s=4) -> channels-last -> LayerNorm (swin_mae3d.py:1120-1129); callable on (B,4,R,R,R) -> (B,g,g,g,C)
    as nerf_rpn uses it (feature_extractor.py:1179)."""

    def forward(self, x):
        m = self._owner()
        m._ensure_ready(x.device)
        m._add_pos = False
        try:
            tok = _EmbedFn.apply(m._anchor, m, x.float().contiguous(), None)
        finally:
            m._add_pos = True
        g = x.shape[2] // 4
        return tok.view(x.shape[0], g, g, g, -1)


class WindowAttention3D(nn.Module):
    def __init__(self, dim, window_size, shift_size, num_heads):
        super().__init__()
        self.window_size, self.shift_size, self.num_heads = window_size, shift_size, num_heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WS - 1) ** 3, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", relative_position_index(WS))


class SwinBlock3D(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, stochastic_depth_prob=0.0, eps=1e-5):
        super().__init__()
        self.dim, self.num_heads, self.shift_size, self.sd_prob = dim, num_heads, list(shift_size), stochastic_depth_prob
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = WindowAttention3D(dim, window_size, shift_size, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        hid = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hid), nn.GELU(), nn.Dropout(0.0), nn.Linear(hid, dim), nn.Dropout(0.0))

    def _sd_noise(self, B, device):
        if not self.training or self.sd_prob == 0.0:
            return None
        keep = 1.0 - self.sd_prob
        return torch.empty(B, device=device).bernoulli_(keep).div_(keep)

    def forward(self, x, sd_noise=None):
        """x: (B,H,W,D,C) channels-last."""
        B, H, W, D, C = x.shape
        geom = WinGeom(B, H, W, D, self.shift_size)
        if sd_noise is None:
            sd_noise = (self._sd_noise(B, x.device), self._sd_noise(B, x.device))
        y = _BlockFn.apply(x.reshape(-1, C), self, geom, sd_noise[0], sd_noise[1])
        return y.view(B, H, W, D, C)


class PatchMerging3D(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim, eps=eps)

    def forward(self, x):
        B, H, W, D, C = x.shape
        geom = WinGeom(B, H, W, D, [0, 0, 0])
        y = _MergeFn.apply(x.reshape(-1, C), self, geom)
        return y.view(B, (H + 1) // 2, (W + 1) // 2, (D + 1) // 2, 2 * C)


class _Stage(nn.Sequential):
    def forward(self, x):
        m = self._owner()
        m._ensure_ready(x.device)
        if x.dtype != m.compute_dtype:
            x = x.to(m.compute_dtype)
        if ops.GROUPED_WGRAD and m.compute_dtype == torch.bfloat16 and torch.is_grad_enabled() and x.requires_grad:
            x = _StageFlushFn.apply(x, m._wq)
        m._packer.wait_late()   # the layouts of stages >= LATE_STAGE are produced on the side stream: a stage called on its own must wait for them too
        for mod in self:
            x = mod(x.contiguous())
        return x


class ResBlock3D(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, 3, 1, 1)
        self.conv2 = nn.Conv3d(cout, cout, 3, 1, 1)
        if cin != cout:
            self.conv3 = nn.Conv3d(cin, cout, 1, 1)


class UpBlock3D(nn.Module):
    def __init__(self, cin, cout, k, use_skip=True):
        super().__init__()
        self.cin, self.cout, self.k, self.use_skip = cin, cout, k, use_skip
        self.transp_conv = nn.ConvTranspose3d(cin, cout, k, stride=k)
        self.conv_block = ResBlock3D(2 * cout if use_skip else cout, cout)
        self.has_proj = use_skip

    def forward(self, x, skip=None, tail=None):
        """channels-last (B,v,v,v,Cin) [+ skip (B,kv,kv,kv,Cout)] -> (B,kv,kv,kv,Cout), or the loss triple when `tail` is given"""
        B, v = x.shape[0], x.shape[1]
        S = v * self.k
        out = _UpBlockFn.apply(x.reshape(-1, self.cin), skip.reshape(-1, self.cout) if self.use_skip else None, self, B, v, tail)
        return out if tail is not None else out.view(B, S, S, S, self.cout)


class OutBlock3D(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 1)


class SwinTransformer_MAE3D_New(nn.Module):
    def __init__(self, patch_size: List[int], embed_dim: int, depths: List[int], num_heads: List[int], window_size: List[int],
                 mlp_ratio: float = 4.0, dropout: float = 0.0, attention_dropout: float = 0.0, stochastic_depth_prob: float = 0.1,
                 norm_layer: Optional[Callable[..., nn.Module]] = None, block: Optional[Callable[..., nn.Module]] = None,
                 downsample_layer: Optional[Callable[..., nn.Module]] = None, expand_dim: bool = True, out_channels: int = 4,
                 input_ch_dim: int = 4, decoder_embed_dim: int = 768, masking_prob=0.50, resolution=160, drop_rate=0.10,
                 masking_strategy="random", compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if list(patch_size) != [4, 4, 4] or list(window_size) != [WS] * 3 or not expand_dim or input_ch_dim != 4 or out_channels != 4:
            raise ValueError("the HIP path implements the configuration the reference trains: patch 4^3, window 4^3, expand_dim, 4->4 channels")
        if dropout != 0.0 or attention_dropout != 0.0:
            raise ValueError("dropout/attention_dropout are 0 in every reference config (run_swin_mae3d.py:400-411)")
        if any(embed_dim * 2 ** s != num_heads[s] * 32 for s in range(len(depths))):
            raise ValueError("head_dim must be 32 (all reference backbones; swin_b uses heads [4,8,16,32], SURVEY 8(c))")
        if embed_dim % 16:
            raise ValueError("embed_dim must be a multiple of 16")
        self.out_channels, self.embed_dim, self.patch_size = out_channels, embed_dim, list(patch_size)
        self.masking_prob, self.resolution, self.compute_dtype = masking_prob, resolution, compute_dtype
        owner = lambda: self  # noqa: E731  (children reach the model without registering it as a submodule)
        self.patch_partition = PatchPartition(nn.Conv3d(4, embed_dim, 4, 4), _Permute([0, 2, 3, 4, 1]), nn.LayerNorm(embed_dim, eps=1e-5))
        self.patch_partition._owner = owner
        self.stages = nn.ModuleList()
        total, bid = sum(depths), 0
        for s, depth in enumerate(depths):
            dim = embed_dim * 2 ** s
            mods: List[nn.Module] = [PatchMerging3D(dim // 2)] if s > 0 else []
            for i in range(depth):
                sd = stochastic_depth_prob * float(bid) / (total - 1)
                mods.append(SwinBlock3D(dim, num_heads[s], list(window_size), [0] * 3 if i % 2 == 0 else [WS // 2] * 3, mlp_ratio, sd))
                bid += 1
            st = _Stage(*mods)
            st._owner = owner
            self.stages.append(st)
        E = embed_dim
        self.decoder4 = UpBlock3D(8 * E, 4 * E, 2)
        self.decoder3 = UpBlock3D(4 * E, 2 * E, 2)
        self.decoder2 = UpBlock3D(2 * E, E, 2)
        self.decoder1 = UpBlock3D(E, E // 2, 4, use_skip=False)
        self.out = OutBlock3D(E // 2, out_channels)
        self.num_patches = g = resolution // patch_size[0]
        self.pos_embed = nn.Parameter(torch.zeros(1, g, g, g, E), requires_grad=False)
        self.mask_token = nn.Parameter(torch.zeros(E))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        self.pos_embed.data.copy_(torch.from_numpy(sincos_pos_embed_3d(E, g)).float())
        nn.init.normal_(self.mask_token, std=0.02)
        self._flat = self._flat_grad = None
        self._packer = None
        self._add_pos = True
        self._anchor = None
        self._reducer = None  # dist.GradReducer when data-parallel
        self._embed_cap = None   # rows per sample of the compact patch embed of the kept tokens (_EmbedFn); None = every token has a row

    # ---- flat buffers + packed weights ------------------------------------------------------------
    def _trainable(self):
        """trainable parameters in *gradient-completion-friendly* order: [mask_token, patch embed | stage0..3 | decoders, head]
        (mask_token's gradient is produced by the embed backward, i.e. last, so it sits with the embed segment)"""
        ps = [self.mask_token] if "mask_token" in self._parameters else []
        seen = {id(p) for p in ps}
        for p in self.parameters():
            if p.requires_grad and id(p) not in seen:
                ps.append(p)
                seen.add(id(p))
        return ps

    def flatten_parameters(self, device=None):
        """Re-home every trainable parameter (and its .grad) as a view of one flat fp32 buffer."""
        ps = self._trainable()
        device = device or ps[0].device
        n = sum((p.numel() + 7) // 8 * 8 for p in ps)   # 8-element granules: every segment boundary is 16-byte aligned in fp32 AND in bf16 buckets
        flat = torch.zeros(n, device=device)
        fg = torch.zeros(n, device=device)
        off = 0
        self._offsets = {}
        for p in ps:
            k = p.numel()
            flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat[off:off + k].view(p.shape)
            p.grad = fg[off:off + k].view(p.shape)
            self._offsets[id(p)] = off
            off += (k + 7) // 8 * 8
        self._flat, self._flat_grad = flat, fg
        self._build_packer(device)
        return flat, fg

    def _build_packer(self, device):
        P = _Packer()
        P.add("pe.w", self.patch_partition[0].weight, P.CAST)
        for s, st in enumerate(self.stages):
            for i, mod in enumerate(st):
                key = f"s{s}.{i}."
                mod._key = key
                if isinstance(mod, PatchMerging3D):
                    P.add(key + "red.w", mod.reduction.weight, P.CAST)
                    P.add(key + "red.wT", mod.reduction.weight, P.TRANS)
                else:
                    for nm, lin in (("qkv", mod.attn.qkv), ("proj", mod.attn.proj), ("fc1", mod.mlp[0]), ("fc2", mod.mlp[3])):
                        P.add(key + nm + ".w", lin.weight, P.CAST)
                        P.add(key + nm + ".wT", lin.weight, P.TRANS)
        for name in ("decoder4", "decoder3", "decoder2", "decoder1"):
            if not hasattr(self, name):
                continue
            d = getattr(self, name)
            key = name + "."
            d._key = key
            P.add(key + "t.w", d.transp_conv.weight, P.CONVT_F)
            P.add(key + "t.wd", d.transp_conv.weight, P.CONVT_D)
            for cn in ("c1", "c2"):
                conv = getattr(d.conv_block, "conv" + cn[1])
                P.add(key + cn + ".w", conv.weight, P.CONV_F)
                P.add(key + cn + ".wd", conv.weight, P.CONV_D)
                if self.compute_dtype == torch.bfloat16 and tuple(conv.weight.shape[:2]) == (48, 48):
                    P.add(key + cn + ".wk", conv.weight, P.C48_F)     # specialised LDS-halo kernel (decoder1 @160^3)
                    P.add(key + cn + ".wkd", conv.weight, P.C48_D)
                elif (self.compute_dtype == torch.bfloat16 and conv.weight.shape[0] % 48 == 0 and conv.weight.shape[1] % 48 == 0
                      and (d.k * (self.resolution // 4) // {"decoder4": 8, "decoder3": 4, "decoder2": 2, "decoder1": 1}[name]) ** 3 >= ops.C48MB_MIN_VOXELS > 0):
                    P.add(key + cn + ".wkm", conv.weight, P.C48_F)    # the LDS-halo kernel on 48-channel blocks (swin_t/s decoder level 40^3)
                    P.add(key + cn + ".wkmd", conv.weight, P.C48_D)
                elif (self.compute_dtype == torch.bfloat16 and conv.weight.shape[0] % 64 == 0 and conv.weight.shape[1] % 64 == 0
                      and (d.k * (self.resolution // 4) // {"decoder4": 8, "decoder3": 4, "decoder2": 2, "decoder1": 1}[name]) ** 3 >= ops.C64_MIN_VOXELS > 0):
                    P.add(key + cn + ".w64", conv.weight, P.C64_F)    # 64-channel-block LDS-halo kernel (swin_b decoder levels >= 32^3)
                    P.add(key + cn + ".w64d", conv.weight, P.C64_D)
            if d.has_proj:
                P.add(key + "c3.w", d.conv_block.conv3.weight, P.CAST)
                P.add(key + "c3.wT", d.conv_block.conv3.weight, P.TRANS)
        P.build(self.compute_dtype, device)
        self._build_swin_streams(P, device)
        d1 = getattr(self, "decoder1", None)
        if (ops.CCONV and d1 is not None and self.compute_dtype == torch.bfloat16 and (d1.cin, d1.cout, d1.k) == (96, 48, 4) and not d1.has_proj
                and (self.resolution // 4) % 8 == 0):
            P.cconv = (d1, torch.empty(ops.cconv_pack_numel(), dtype=torch.bfloat16, device=device), torch.empty((27, 48), device=device),
                       torch.empty(ops.cconv_pack_ws_floats(), dtype=torch.float32, device=device),
                       torch.empty(ops.upconv4_pack_numel(), dtype=torch.bfloat16, device=device) if ops.UPCONV4 else None,
                       torch.empty(ops.cconv_dgrad_pack_numel(), dtype=torch.bfloat16, device=device) if (ops.CCONV_DGRAD and ops.CCONV_WGRAD) else None,
                       torch.empty((27, 96, 48), dtype=torch.float32, device=device) if ops.CCONV_CENTERED else None,
                       torch.empty(ops.tail_residual_pack_numel(), dtype=torch.bfloat16, device=device) if ops.TAIL_FROM_COARSE else None)
        self._packer = P
        self._pk = P
        self._wq = ops.WgradQueue()   # deferred encoder weight gradients (grouped launches, issued per stage)
        for mod in self.modules():
            if isinstance(mod, (SwinBlock3D, PatchMerging3D, UpBlock3D)):
                mod._pk = P
                mod._wq = self._wq
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

    def _build_swin_streams(self, P, device):
        """weight streams of the fused Swin-block kernels (csrc/swin_block.hip) for every block of a supported width: one flat bf16 buffer, packed from
        the fp32 masters by one launch per group (stages < / >= LATE_STAGE, like the other encoder layouts)"""
        if not (ops.SWIN_FUSED and self.compute_dtype == torch.bfloat16):
            return
        kinds = [ops.SWIN_ATTN_FWD, ops.SWIN_MLP_FWD]
        src = {ops.SWIN_ATTN_FWD: lambda b: (b.attn.qkv.weight, b.attn.proj.weight), ops.SWIN_MLP_FWD: lambda b: (b.mlp[0].weight, b.mlp[3].weight)}
        blocks = [(s, b) for s, st in enumerate(self.stages) for b in st if isinstance(b, SwinBlock3D) and ops.swin_supported(b.dim)]
        total = sum((ops.swin_stream_numel(k, b.dim) + 63) // 64 * 64 for _, b in blocks for k in kinds)
        if not total:
            return
        buf = torch.empty(total, dtype=torch.bfloat16, device=device)
        early, late, off = [], [], 0
        for s, b in blocks:
            b._sw = {}
            for k in kinds:
                n = ops.swin_stream_numel(k, b.dim)
                b._sw[k] = buf[off:off + n]
                w0, w1 = src[k](b)
                (late if s >= _Packer.LATE_STAGE else early).append((w0.data, w1.data if w1 is not None else None, b._sw[k], k, b.dim))
                off += (n + 63) // 64 * 64
        P.swin_buf = buf
        P.swin_early = ops.swin_pack_items(early) if early else None
        P.swin_late = ops.swin_pack_items(late) if late else None

    def _ensure_ready(self, device):
        ps = self._trainable()
        stale = (self._flat is None or self._flat.device != ps[0].device or
                 any(p.data_ptr() != self._flat.data_ptr() + 4 * self._offsets.get(id(p), -1) for p in ps))
        if stale:
            if not ps[0].is_cuda:
                raise RuntimeError("SwinTransformer_MAE3D (HIP) needs its parameters on a HIP device: call .cuda() first (no CPU fallback)")
            self.flatten_parameters()
        self._packer.run()

    def zero_grad(self, set_to_none: bool = False):
        if self._flat_grad is not None:
            self._flat_grad.zero_()
        else:
            super().zero_grad(set_to_none=set_to_none)

    # ---- reference API ----------------------------------------------------------------------------
    def transform(self, x: List[Tensor], device):
        """pad_tensor semantics (torch_utils.py:56-90) without materialising the ones-mask: returns the padded
        batch (B,4,R,R,R) fp32 and the valid extents [B,3] (the analytic mask)."""
        R = self.resolution
        xb = torch.zeros((len(x), 4, R, R, R), dtype=torch.float32, device=device)
        ext = []
        for i, t in enumerate(x):
            a0, a1, a2 = t.shape[1:]
            xb[i, :, :a0, :a1, :a2] = t.to(device=device, dtype=torch.float32, non_blocking=True)
            ext.append([a0, a1, a2])
        return xb, torch.tensor(ext, dtype=torch.int32).to(device, non_blocking=True)

    def _draw_sd_noise(self, B: int, device):
        """row-mode stochastic-depth factors of every block (attention and MLP branch) in TWO launches instead of four per block
        (torchvision StochasticDepth(p, "row"): bernoulli(1-p)/(1-p) per sample, swin_mae3d.py:390-414)"""
        blocks = [m for st in self.stages for m in st if isinstance(m, SwinBlock3D)]
        if not self.training or all(b.sd_prob == 0.0 for b in blocks):
            return None
        keep = getattr(self, "_sd_keep", None)
        if keep is None or keep.device != device or keep.shape[1] != B:
            k = torch.tensor([1.0 - b.sd_prob for b in blocks for _ in (0, 1)], device=device)
            keep = self._sd_keep = k[:, None].expand(-1, B).contiguous()
        noise = torch.bernoulli(keep).div_(keep)
        return [(noise[2 * i], noise[2 * i + 1]) for i in range(len(blocks))]

    def _run_stage(self, si: int, x: Tensor, sd_noise, bi: int, red=None):
        """stage `si` on channels-last x; `bi` = index of the stage's first block in `sd_noise`.  Returns (output, next block index).
        With a data-parallel reducer the stage (each block group of the chunked stage, dist.GradReducer) is preceded by the trigger that
        launches its gradient range's all-reduce -- backward order: blocks of the group, the flush that issues (and joins) their queued weight
        gradients, then the trigger."""
        if si >= _Packer.LATE_STAGE:
            self._packer.wait_late()
        grouped = ops.GROUPED_WGRAD and self.compute_dtype == torch.bfloat16 and torch.is_grad_enabled() and x.requires_grad
        groups = red.chunk_groups if (red is not None and si == getattr(red, "chunk_stage", -1) and red.chunk_groups) else [list(self.stages[si])]
        if red is None and grouped and ops.STAGE_FLUSH_BLOCKS > 0 and len(groups) == 1:
            # a long stage (stage 2: 18 blocks) flushes its queued weight gradients every STAGE_FLUSH_BLOCKS blocks: flushed once at the stage's end they all
            # run under stage 1 / stage 0, while the side stream idles for the last two thirds of the stage's own input-gradient chain (round-5 trace)
            nblk = sum(1 for mod in groups[0] if isinstance(mod, SwinBlock3D))
            if nblk > ops.STAGE_FLUSH_BLOCKS:
                from .dist import stage_chunk_groups
                groups = stage_chunk_groups(self, si, -(-nblk // ops.STAGE_FLUSH_BLOCKS))
        if red is None and si == 0 and grouped and ops.STAGE0_BLOCK_FLUSH and len(groups) == 1 and len(groups[0]) > 1:
            # stage 0 is the END of the backward pass: flushed as a whole, all of its weight gradients run exposed after the last input-gradient
            # kernel; per block, the second block's run under the first block's chain
            groups = [[mod] for mod in groups[0]]
        for k, grp in enumerate(groups):
            if red is not None:
                x = red.trigger(x, red.seg_stage(si, k))  # backward reaching here => this range's gradients are complete
            if grouped:
                x = _StageFlushFn.apply(x, self._wq, si == 0 and k == 0)
            for mod in grp:
                if isinstance(mod, SwinBlock3D):
                    x = mod(x, None if sd_noise is None else sd_noise[bi])
                    bi += 1
                else:
                    x = mod(x)
        return x, bi

    def forward_encoder(self, tok: Tensor, sd_noise=None):
        feats, x, bi = [], tok, 0
        red = self._reducer
        if sd_noise is None:
            sd_noise = self._draw_sd_noise(tok.shape[0], tok.device)
        self._wq.sync_after_flush = red is not None   # a gradient all-reduce of the range follows the flush: join before it
        for si in range(len(self.stages)):
            x, bi = self._run_stage(si, x, sd_noise, bi, red)
            if si + 1 < len(self.stages) and torch.is_grad_enabled() and x.requires_grad:
                x, skip = _Fork2Fn.apply(x)     # consumed by the next stage and by a decoder / neck
                feats.append(skip)
            else:
                feats.append(x)
        return feats

    def forward_decoder(self, feats: List[Tensor], tail=None) -> Tensor:
        self._packer.join()   # decoder weight layouts are packed on the side stream while the encoder runs
        arena = ops.AccArena.get(feats[3].device)
        if arena is not None:
            arena.begin()     # one clearing launch for every InstanceNorm accumulator of this forward + backward (ops.AccArena)
        f3 = feats[3]
        if self._reducer is not None:
            f3 = self._reducer.trigger(f3, self._reducer.seg_decoder())  # decoder4 is the last decoder op in backward order
            if torch.is_grad_enabled() and f3.requires_grad:
                # backward order: decoder4, THIS flush (the decoder's queued small-level weight gradients are issued and joined), then the
                # trigger that starts the decoder segment's all-reduce
                self._wq.sync_after_flush = True
                f3 = _StageFlushFn.apply(f3, self._wq)
        d = self.decoder4(f3, feats[2])
        d = self.decoder3(d, feats[1])
        d = self.decoder2(d, feats[0])
        return self.decoder1(d, tail=tail)

    fuse_tail = True   # loss backward fused with the last decoder level's InstanceNorm backward (False: separate kernels)
    fuse_tail_sums = __import__("os").environ.get("NMH_TAIL_SUMS", "1") != "0"   # the tail backward's reductions taken in the tail forward pass

    def _decode_and_loss(self, feats, xb, ext, mask_dev, pred):
        if self.fuse_tail:
            return self.forward_decoder(feats, tail=(self, xb, ext, mask_dev, pred))
        d0 = self.forward_decoder(feats)
        return _LossFn.apply(d0.reshape(-1, d0.shape[-1]), self, xb, ext, mask_dev, pred)

    def forward(self, x: List[Tensor], is_eval: bool = False, block_mask: Optional[Tensor] = None, sd_noise=None, return_pred: bool = False):
        device = self.mask_token.device
        self._ensure_ready(device)
        xb, ext = self.transform(x, device)
        B, R = xb.shape[0], self.resolution
        g = R // 4
        if block_mask is None:
            block_mask = draw_block_mask((g, g, g), self.masking_prob)
        mask_dev = block_mask.to(torch.uint8).contiguous().view(-1).to(device, non_blocking=True)
        # the mask was drawn on the host: the compact patch embed gets exactly the rows it needs (64-row granules)
        cap_was, self._embed_cap = self._embed_cap, embed_capacity_rows(g, kept=int(g ** 3 - int(block_mask.to(torch.int64).sum())))
        try:
            tok = _EmbedFn.apply(self._anchor, self, xb, mask_dev).view(B, g, g, g, self.embed_dim)
        finally:
            self._embed_cap = cap_was
        feats = self.forward_encoder(tok, sd_noise)
        want_pred = is_eval or return_pred
        pred = torch.empty((B, 4, R, R, R), device=device) if want_pred else None
        losses = self._decode_and_loss(feats, xb, ext, mask_dev, pred)
        loss, loss_rgb, loss_alpha = losses[0], losses[1], losses[2]
        if return_pred:
            return loss, loss_rgb, loss_alpha, pred
        if is_eval:
            patch = lambda t: t.reshape(B, 4, g, 4, g, 4, g, 4).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, g, g, g, 64, 4)  # noqa: E731
            tgt = patch(xb)
            return loss, loss_rgb, loss_alpha, patch(pred), tgt[..., 3:] > 0.01, tgt
        return loss, loss_rgb, loss_alpha

    def forward_static(self, xb: Tensor, ext: Tensor, mask_dev: Tensor):
        """graph-capturable training forward: everything already on the device (padded batch (B,4,R,R,R) fp32, extents [B,3]
        int32, token mask [g^3] uint8); no host<->device traffic, no python RNG."""
        B, R = xb.shape[0], self.resolution
        g = R // 4
        self._packer.run()
        tok = _EmbedFn.apply(self._anchor, self, xb, mask_dev).view(B, g, g, g, self.embed_dim)
        losses = self._decode_and_loss(self.forward_encoder(tok), xb, ext, mask_dev, None)
        return losses[0], losses[1], losses[2]

    def encoder_features(self, xb: Tensor) -> List[Tensor]:
        """nerf_rpn contract (feature_extractor.py:1176-1187): NCDHW feature list [C,2C,4C,8C]."""
        t = self.patch_partition(xb)
        t = t + self.pos_embed.to(t.dtype)
        feats = []
        for st in self.stages:
            t = st(t)
            feats.append(t.permute(0, 4, 1, 2, 3).contiguous())
        return feats


SwinTransformer_MAE3D = SwinTransformer_MAE3D_New  # the alias run_swin_mae3d.py:22 imports

SWIN_CONFIGS = {
    "swin_t": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24]),
    "swin_s": dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24]),
    "swin_b": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32]),  # defined deviation, SURVEY 8(c)
    "swin_l": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48]),
}


def build_model(backbone_type: str = "swin_s", resolution: int = 160, masking_prob: float = 0.75, stochastic_depth_prob: float = 0.1,
                compute_dtype: torch.dtype = torch.bfloat16) -> SwinTransformer_MAE3D_New:
    """The config dict of run_swin_mae3d.py:378-411."""
    cfg = SWIN_CONFIGS[backbone_type]
    return SwinTransformer_MAE3D_New(patch_size=[4, 4, 4], embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                                     window_size=[4, 4, 4], stochastic_depth_prob=stochastic_depth_prob, expand_dim=True,
                                     resolution=resolution, masking_prob=masking_prob, compute_dtype=compute_dtype)
