"""The native op surface of SURVEY.md section 8(b), by the names listed there, as plain functions on tensors:

    window_attn_fwd / window_attn_bwd       shifted_window_attention (swin_mae3d.py:27-197): pad, roll, window partition, qkv, attention with
                                            relative-position bias and shift mask, proj, window reverse, un-roll, un-pad
    ln_mlp_fwd / ln_mlp_bwd                 LayerNorm -> Linear -> GELU -> Linear (swin_mae3d.py:352-369, the MLP branch of a block)
    patch_merge_fwd / patch_merge_bwd       PatchMerging (swin_mae3d.py:372-414)
    patch_embed_fwd / patch_embed_bwd       Conv3d(4, C, k = s = 4) -> channels-last -> LayerNorm (swin_mae3d.py:1119-1131)
    convT_k_eq_s_fwd / _bwd                 ConvTranspose3d(kernel = stride) (unetr_block.py:151-158)
    conv3d_3x3x3_fwd / _dgrad / _wgrad      nn.Conv3d(k = 3, padding = 1, bias-free use) (unetr_block.py:35-44)
    instnorm_lrelu_add_fwd / _bwd           InstanceNorm3d (+ residual) -> LeakyReLU(0.01) (unetr_block.py:57-71)
    mae_loss_fwd / mae_loss_bwd             UnetOutBlock 1x1 conv + forward_loss (swin_mae3d.py:1513-1549)
    adamw_clip_step                         clip_grad_norm_ + AdamW on flat buffers (run_swin_mae3d.py:665-669)

Every function is a composition of the C-ABI entry points of include/nerfmae_hip.h through `ops` (the model's autograd Functions in
model.py fuse further: LayerNorm with the window gather, the window reverse with the proj GEMM's store, deferred grouped weight
gradients); this module is the per-operator face of the same kernels -- what a reference maintainer would call op by op, and what the
operator-level parity tests (tests/test_surface_gpu.py) exercise against the oracle's restatement of each reference function.
Activations: channels-last, compute dtype bf16 or fp32; parameters fp32 (cast to the compute dtype here); parameter gradients fp32.
No CPU fallback: every function raises on non-HIP tensors (ops._chk)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
from torch import Tensor

from . import ops


def _w(t: Tensor, dtype) -> Tensor:
    return t.detach().to(dtype).contiguous()


# ---- shifted-window attention ------------------------------------------------------------------------------------------------------------
def window_attn_fwd(x: Tensor, qkv_w: Tensor, qkv_b: Tensor, proj_w: Tensor, proj_b: Tensor, bias_table: Tensor, shift: Sequence[int], heads: int):
    """x (B,H,W,D,C) -> (y (B,H,W,D,C), saved).  head_dim = C / heads must be 32 (every Swin variant of the reference)."""
    B, H, W, D, C = x.shape
    geom = ops.WinGeom(B, H, W, D, shift)
    dt, dev = x.dtype, x.device
    xt = x.reshape(-1, C).contiguous()
    xw = torch.empty((geom.rows, C), dtype=dt, device=dev)
    ops.window_gather_scale(xt, xw, None, C, geom)                      # pad + roll + window partition (pad rows = 0)
    qkv = ops.gemm_nt(xw, _w(qkv_w, dt), bias=qkv_b)
    o = torch.empty((geom.rows, C), dtype=dt, device=dev)
    lse = torch.empty(geom.rows * heads, device=dev)
    ops.window_attn_fwd(qkv, bias_table, o, lse, heads, C, geom)
    yw = ops.gemm_nt(o, _w(proj_w, dt), bias=proj_b)
    y = torch.empty_like(xt)
    ops.window_scatter_residual(yw, torch.zeros_like(xt), y, None, C, geom)   # window reverse + un-roll + un-pad
    return y.view(B, H, W, D, C), (geom, heads, xw, qkv, o, lse, qkv_w, proj_w, bias_table)


def window_attn_bwd(dy: Tensor, saved) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (dx, dqkv_w, dqkv_b, dproj_w, dproj_b, dbias_table)"""
    geom, heads, xw, qkv, o, lse, qkv_w, proj_w, bias_table = saved
    C, dt, dev = xw.shape[1], xw.dtype, xw.device
    dyt = dy.reshape(-1, C).contiguous()
    dyw = torch.empty((geom.rows, C), dtype=dt, device=dev)
    ops.window_gather_scale(dyt, dyw, None, C, geom)
    dproj_w, dproj_b = torch.zeros(C, C, device=dev), torch.zeros(C, device=dev)
    ops.gemm_tn(dyw, o, dproj_w, dbias=dproj_b)
    do = ops.gemm_nt(dyw, _w(proj_w.t(), dt))
    dqkv = torch.empty_like(qkv)
    dtable = torch.zeros_like(bias_table)
    ops.window_attn_bwd(qkv, bias_table, do, lse, dqkv, dtable, heads, C, geom)
    dqkv_w, dqkv_b = torch.zeros(3 * C, C, device=dev), torch.zeros(3 * C, device=dev)
    ops.gemm_tn(dqkv, xw, dqkv_w, dbias=dqkv_b)
    dxw = ops.gemm_nt(dqkv, _w(qkv_w.t(), dt))
    dx = torch.empty_like(dyt)
    ops.window_scatter_residual(dxw, torch.zeros_like(dyt), dx, None, C, geom)
    return dx.view(geom.B, geom.H, geom.W, geom.D, C), dqkv_w, dqkv_b, dproj_w, dproj_b, dtable


# ---- LayerNorm + MLP ---------------------------------------------------------------------------------------------------------------------
def ln_mlp_fwd(x: Tensor, ln_g: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
    """x (..., C) -> (fc2(gelu(fc1(LN(x)))), saved)"""
    C = x.shape[-1]
    xt = x.reshape(-1, C).contiguous()
    T, dt, dev = xt.shape[0], x.dtype, x.device
    xn = torch.empty_like(xt)
    mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
    ops.layernorm_fwd(xt, ln_g, ln_b, xn, mean, rstd, T, C)
    h_pre = torch.empty((T, w1.shape[0]), dtype=dt, device=dev)
    h = ops.gemm_nt(xn, _w(w1, dt), bias=b1, act=1, C2=h_pre)
    y = ops.gemm_nt(h, _w(w2, dt), bias=b2)
    return y.view(x.shape), (xt, xn, mean, rstd, h_pre, h, ln_g, w1, w2)


def ln_mlp_bwd(dy: Tensor, saved):
    """-> (dx, dln_g, dln_b, dw1, db1, dw2, db2)"""
    xt, xn, mean, rstd, h_pre, h, ln_g, w1, w2 = saved
    T, C = xt.shape
    dt, dev = xt.dtype, xt.device
    dyt = dy.reshape(T, C).contiguous()
    dw2, db2 = torch.zeros_like(w2, dtype=torch.float32), torch.zeros(C, device=dev)
    ops.gemm_tn(dyt, h, dw2, dbias=db2)
    dh = ops.gemm_nt(dyt, _w(w2.t(), dt), act=2, C2=h_pre)              # (dy W2) * gelu'(pre)
    dw1, db1 = torch.zeros_like(w1, dtype=torch.float32), torch.zeros(w1.shape[0], device=dev)
    ops.gemm_tn(dh, xn, dw1, dbias=db1)
    dxn = ops.gemm_nt(dh, _w(w1.t(), dt))
    dx = torch.empty_like(xt)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_bwd(dxn, xt, ln_g, mean, rstd, dx, dg, db, T, C)
    return dx.view(dy.shape), dg, db, dw1, db1, dw2, db2


# ---- patch merging -----------------------------------------------------------------------------------------------------------------------
def patch_merge_fwd(x: Tensor, norm_g: Tensor, norm_b: Tensor, red_w: Tensor):
    """x (B,H,W,D,C) -> ((B,ceil(H/2),ceil(W/2),ceil(D/2),2C), saved)"""
    B, H, W, D, C = x.shape
    geom = ops.WinGeom(B, H, W, D, (0, 0, 0))
    H2, W2, D2 = (H + 1) // 2, (W + 1) // 2, (D + 1) // 2
    rows = B * H2 * W2 * D2
    xt = x.reshape(-1, C).contiguous()
    xg = torch.empty((rows, 8 * C), dtype=x.dtype, device=x.device)
    mean, rstd = torch.empty(rows, device=x.device), torch.empty(rows, device=x.device)
    ops.layernorm_fwd(xt, norm_g, norm_b, xg, mean, rstd, rows, 8 * C, src_mode=2, geom=geom)
    y = ops.gemm_nt(xg, _w(red_w, x.dtype))
    return y.view(B, H2, W2, D2, 2 * C), (geom, rows, xt, xg, mean, rstd, norm_g, red_w)


def patch_merge_bwd(dy: Tensor, saved):
    """-> (dx, dnorm_g, dnorm_b, dred_w)"""
    geom, rows, xt, xg, mean, rstd, norm_g, red_w = saved
    C, dt, dev = xt.shape[1], xt.dtype, xt.device
    dyt = dy.reshape(rows, 2 * C).contiguous()
    dred = torch.zeros_like(red_w, dtype=torch.float32)
    ops.gemm_tn(dyt, xg, dred)
    dxg = ops.gemm_nt(dyt, _w(red_w.t(), dt))
    dx = torch.empty_like(xt)
    dg, db = torch.zeros(8 * C, device=dev), torch.zeros(8 * C, device=dev)
    ops.layernorm_bwd(dxg, xt, norm_g, mean, rstd, dx, dg, db, rows, 8 * C, src_mode=2, geom=geom)
    return dx.view(geom.B, geom.H, geom.W, geom.D, C), dg, db, dred


# ---- patch embedding ---------------------------------------------------------------------------------------------------------------------
def patch_embed_fwd(xb: Tensor, conv_w: Tensor, conv_b: Tensor, ln_g: Tensor, ln_b: Tensor, dtype=torch.bfloat16):
    """xb fp32 (B,4,R,R,R) -> (tokens (B,g,g,g,C) in `dtype`, saved)"""
    B, R = xb.shape[0], xb.shape[2]
    g, C = R // 4, conv_w.shape[0]
    T = B * g ** 3
    A = torch.empty((T, 256), dtype=dtype, device=xb.device)
    ops.patch_embed_gather(xb.contiguous(), A, B, R)
    y0 = ops.gemm_nt(A, _w(conv_w.reshape(C, 256), dtype), bias=conv_b)
    tok = torch.empty((T, C), dtype=dtype, device=xb.device)
    mean, rstd = torch.empty(T, device=xb.device), torch.empty(T, device=xb.device)
    ops.layernorm_fwd(y0, ln_g, ln_b, tok, mean, rstd, T, C)
    return tok.view(B, g, g, g, C), (A, y0, mean, rstd, ln_g, conv_w.shape)


def patch_embed_bwd(dtok: Tensor, saved):
    """-> (dconv_w, dconv_b, dln_g, dln_b)   (the input grid has no gradient on this path)"""
    A, y0, mean, rstd, ln_g, wshape = saved
    T, C = y0.shape
    dev = y0.device
    dy0 = torch.empty_like(y0)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_bwd(dtok.reshape(T, C).contiguous(), y0, ln_g, mean, rstd, dy0, dg, db, T, C)
    dw, dbias = torch.zeros(C, 256, device=dev), torch.zeros(C, device=dev)
    ops.gemm_tn(dy0, A, dw, dbias=dbias)
    return dw.view(wshape), dbias, dg, db


# ---- ConvTranspose3d (kernel = stride) ------------------------------------------------------------------------------------------------------
def convT_k_eq_s_fwd(x: Tensor, w: Tensor, b: Tensor, k: int):
    """x (B,v,v,v,Cin), w (Cin,Cout,k,k,k) -> ((B,vk,vk,vk,Cout), saved)"""
    B, v, Cin = x.shape[0], x.shape[1], x.shape[-1]
    Cout, k3 = w.shape[1], k ** 3
    dt = x.dtype
    xt = x.reshape(-1, Cin).contiguous()
    wf = _w(w.reshape(Cin, Cout, k3).permute(2, 1, 0).reshape(k3 * Cout, Cin), dt)          # [(tap, co)][ci]
    y = torch.empty((B * (v * k) ** 3, Cout), dtype=dt, device=x.device)
    ops.upconv_fwd(xt, wf, b, y, B, v, k, Cin, Cout)
    return y.view(B, v * k, v * k, v * k, Cout), (xt, w, B, v, k)


def convT_k_eq_s_bwd(dy: Tensor, saved):
    """-> (dx, dw, db)"""
    xt, w, B, v, k = saved
    Cin, Cout, k3 = w.shape[0], w.shape[1], k ** 3
    dyt = dy.reshape(-1, Cout).contiguous()
    wd = _w(w.reshape(Cin, Cout, k3).permute(0, 2, 1).reshape(Cin, k3 * Cout), xt.dtype)      # [ci][(tap, co)]
    dx = torch.empty_like(xt)
    ops.upconv_dgrad(dyt, wd, dx, B, v, k, Cin, Cout)
    dw, db = torch.zeros_like(w, dtype=torch.float32), torch.zeros(Cout, device=xt.device)
    ops.upconv_wgrad(dyt, xt, dw, db, B, v, k, Cin, Cout)
    return dx.view(B, v, v, v, Cin), dw, db


# ---- 3x3x3 convolution -------------------------------------------------------------------------------------------------------------------
def conv3d_3x3x3_fwd(x: Tensor, w: Tensor) -> Tensor:
    """x (B,D,H,W,Cin), w (Cout,Cin,3,3,3) -> (B,D,H,W,Cout) (padding 1, no bias: the reference's use in front of InstanceNorm)"""
    Cout, Cin = w.shape[:2]
    wp = _w(w.reshape(Cout, Cin, 27).permute(0, 2, 1), x.dtype)                             # [Cout][27][Cin]
    return ops.conv3d_k3(x.contiguous(), wp, Cout)


def conv3d_3x3x3_dgrad(dy: Tensor, w: Tensor) -> Tensor:
    Cout, Cin = w.shape[:2]
    wd = _w(w.reshape(Cout, Cin, 27).flip(2).permute(1, 2, 0), dy.dtype)                    # [Cin][27 flipped][Cout]
    return ops.conv3d_k3(dy.contiguous(), wd, Cin)


def conv3d_3x3x3_wgrad(dy: Tensor, x: Tensor) -> Tensor:
    dw = torch.zeros(dy.shape[-1], x.shape[-1], 3, 3, 3, device=x.device)
    return ops.conv3d_k3_wgrad(dy.contiguous(), x.contiguous(), dw)


# ---- InstanceNorm (+ residual) + LeakyReLU -----------------------------------------------------------------------------------------------
def instnorm_lrelu_add_fwd(x: Tensor, residual: Tensor = None, slope: float = 0.01):
    """x (B, ..., C) channels-last -> (lrelu(IN(x) [+ residual]), saved)"""
    B, C = x.shape[0], x.shape[-1]
    V = x.numel() // (B * C)
    xt = x.reshape(B, V, C).contiguous()
    stats = torch.empty((B, C, 2), device=x.device)
    ops.instnorm_stats(xt, stats, torch.empty((B, C, 2), dtype=torch.float64, device=x.device), B, V, C)
    out = torch.empty_like(xt)
    r = None if residual is None else residual.reshape(B, V, C).contiguous()
    ops.instnorm_apply(xt, stats, out, B, V, C, r=r, rmode=0 if r is None else 1, slope=slope)
    return out.view(x.shape), (xt, stats, out, r is not None, slope)


def instnorm_lrelu_add_bwd(dout: Tensor, saved):
    """-> (dx, dresidual or None)"""
    xt, stats, out, has_r, slope = saved
    B, V, C = xt.shape
    d = dout.reshape(B, V, C).contiguous()
    sums = torch.empty((B, C, 2), dtype=torch.float64, device=xt.device)
    rmode = 1 if has_r else 0
    ops.instnorm_bwd_reduce(d, out, xt, stats, sums, B, V, C, rmode=rmode, slope=slope)
    dx = torch.empty_like(xt)
    dr = torch.empty_like(xt) if has_r else None
    ops.instnorm_bwd_apply(d, out, xt, stats, sums, dx, B, V, C, rmode=rmode, dr=dr, slope=slope)
    return dx.view(dout.shape), (dr.view(dout.shape) if has_r else None)


# ---- 1x1 head + masked MSE ---------------------------------------------------------------------------------------------------------------
def mae_loss_fwd(d0: Tensor, w_out: Tensor, b_out: Tensor, target: Tensor, extents: Tensor, tokmask: Tensor):
    """d0 (B,R,R,R,Cd), target fp32 (B,4,R,R,R), extents int32 (B,3), tokmask uint8 (g,g,g) -> ((loss, loss_rgb, loss_alpha), pred, saved)"""
    B, R, Cd = d0.shape[0], d0.shape[1], d0.shape[-1]
    dev = d0.device
    sums, losses = torch.empty(8, dtype=torch.float64, device=dev), torch.empty(3, device=dev)
    pred = torch.empty((B, 4, R, R, R), device=dev)
    dpred = torch.empty((B * R ** 3, 4), device=dev)
    d0t = d0.reshape(-1, Cd).contiguous()
    wo = w_out.reshape(4, Cd).contiguous()
    ops.mae_loss_fwd(d0t, wo, b_out, target, extents, tokmask, B, R, Cd, sums, losses, pred, dpred)
    return losses, pred, (d0t, wo, b_out, target, extents, tokmask, B, R, Cd, sums)


def mae_loss_bwd(saved):
    """gradient of loss (= losses[0]) -> (dd0, dw_out, db_out)"""
    d0t, wo, b_out, target, extents, tokmask, B, R, Cd, sums = saved
    dd0 = torch.empty_like(d0t)
    dW, db = torch.zeros(4, Cd, device=d0t.device), torch.zeros(4, device=d0t.device)
    ops.mae_loss_bwd(d0t, wo, b_out, target, extents, tokmask, B, R, Cd, sums, dd0, torch.empty((d0t.shape[0], 8), dtype=d0t.dtype, device=d0t.device), dW, db)
    return dd0.view(B, R, R, R, Cd), dW, db


# ---- clip + AdamW ------------------------------------------------------------------------------------------------------------------------
def adamw_clip_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                    weight_decay: float = 1e-2, max_grad_norm: float = 0.0) -> Tensor:
    """flat fp32 buffers, in place: clip_grad_norm_(max_grad_norm) (0 = off) then one torch.optim.AdamW step number `step` (1-based);
    returns the pre-clip gradient norm (device scalar)"""
    dev = p.device
    acc, coef, norm = torch.empty(1, dtype=torch.float64, device=dev), torch.empty(1, device=dev), torch.empty(1, device=dev)
    ops.grad_sqnorm(g, acc)
    ops.clip_coef(acc, max_grad_norm, coef, norm)
    hyper = torch.tensor([lr, beta1, beta2, eps, weight_decay, 1.0 - beta1 ** step, 1.0 - beta2 ** step, 0.0], device=dev)
    ops.adamw_step(p, g, m, v, hyper, coef)
    return norm
