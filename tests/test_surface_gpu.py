"""Operator-level parity of the SURVEY section 8(b) op surface (nerf_mae_amd.surface) against the oracle's restatement of each reference
function, forward and every gradient, on the same seeded inputs: fp32 (exact-MFMA mode) tight, bf16 to bf16 rounding.  These read like
the per-function checks the reference itself would hold for swin_mae3d.py / unetr_block.py."""
import random

import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import DTS, check, dev, q, rnd

pytestmark = pytest.mark.gpu


def _S():
    from nerf_mae_amd import surface
    return surface


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,shift,C,heads", [((2, 8, 8, 8), 0, 96, 3), ((1, 5, 6, 7), 2, 64, 2), ((1, 10, 10, 10), 2, 96, 3), ((1, 2, 2, 2), 2, 32, 1)])
def test_window_attn(dt, shape, shift, C, heads):
    """shifted_window_attention (swin_mae3d.py:27-197): padded / shifted / degenerate windows; all six gradients"""
    from oracle import mae3d_oracle as O
    S = _S()
    x = q(rnd(*shape, C), dt)
    qw, qb = q(rnd(3 * C, C, seed=1, scale=C ** -0.5), dt), rnd(3 * C, seed=2, scale=0.1)
    pw, pb = q(rnd(C, C, seed=3, scale=C ** -0.5), dt), rnd(C, seed=4, scale=0.1)
    tab = rnd(343, heads, seed=5, scale=0.5)
    dy = q(rnd(*shape, C, seed=6), dt)
    ref_in = [t.clone().requires_grad_(True) for t in (x, qw, qb, pw, pb, tab)]
    yr = O.window_attention(ref_in[0], ref_in[1], ref_in[2], ref_in[3], ref_in[4], ref_in[5], heads, [shift] * 3)
    yr.backward(dy)
    y, saved = S.window_attn_fwd(dev(x, dt), dev(qw), dev(qb), dev(pw), dev(pb), dev(tab), [shift] * 3, heads)
    check(y, yr, dt, "y")
    grads = S.window_attn_bwd(dev(dy, dt), saved)
    for gk, tr, nm in zip(grads, ref_in, ("dx", "dqkv_w", "dqkv_b", "dproj_w", "dproj_b", "dbias_table")):
        check(gk, tr.grad, dt, nm, 3)


@pytest.mark.parametrize("dt", DTS)
def test_ln_mlp(dt):
    """LayerNorm -> Linear -> GELU -> Linear (swin_mae3d.py:352-369)"""
    S = _S()
    T, C = 777, 96
    x = q(rnd(T, C) * 1.5 + 0.3, dt)
    g, b = rnd(C, seed=1, scale=0.2) + 1.0, rnd(C, seed=2, scale=0.1)
    w1, b1 = q(rnd(4 * C, C, seed=3, scale=C ** -0.5), dt), rnd(4 * C, seed=4, scale=0.1)
    w2, b2 = q(rnd(C, 4 * C, seed=5, scale=(4 * C) ** -0.5), dt), rnd(C, seed=6, scale=0.1)
    dy = q(rnd(T, C, seed=7), dt)
    ref_in = [t.clone().requires_grad_(True) for t in (x, g, b, w1, b1, w2, b2)]
    xn = F.layer_norm(ref_in[0], (C,), ref_in[1], ref_in[2], 1e-5)
    yr = F.linear(F.gelu(F.linear(xn, ref_in[3], ref_in[4])), ref_in[5], ref_in[6])
    yr.backward(dy)
    y, saved = S.ln_mlp_fwd(dev(x, dt), dev(g), dev(b), dev(w1), dev(b1), dev(w2), dev(b2))
    check(y, yr, dt, "y", 2)
    for gk, tr, nm in zip(S.ln_mlp_bwd(dev(dy, dt), saved), ref_in, ("dx", "dln_g", "dln_b", "dw1", "db1", "dw2", "db2")):
        check(gk, tr.grad, dt, nm, 4)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 8, 8, 8), (1, 5, 5, 5), (1, 6, 5, 4)])
def test_patch_merge(dt, shape):
    """PatchMerging (swin_mae3d.py:372-414) on even / odd / mixed extents"""
    from oracle import mae3d_oracle as O
    S = _S()
    C = 16
    x = q(rnd(*shape, C), dt)
    g, b = rnd(8 * C, seed=1, scale=0.2) + 1.0, rnd(8 * C, seed=2, scale=0.1)
    w = q(rnd(2 * C, 8 * C, seed=3, scale=(8 * C) ** -0.5), dt)
    ref_in = [t.clone().requires_grad_(True) for t in (x, g, b, w)]
    yr = F.linear(F.layer_norm(O.patch_merge_gather(ref_in[0]), (8 * C,), ref_in[1], ref_in[2], 1e-5), ref_in[3])
    dy = q(rnd(*yr.shape, seed=4), dt)
    yr.backward(dy)
    y, saved = S.patch_merge_fwd(dev(x, dt), dev(g), dev(b), dev(w))
    check(y, yr, dt, "y", 2)
    for gk, tr, nm in zip(S.patch_merge_bwd(dev(dy, dt), saved), ref_in, ("dx", "dnorm_g", "dnorm_b", "dred_w")):
        check(gk, tr.grad, dt, nm, 4)


@pytest.mark.parametrize("dt", DTS)
def test_patch_embed(dt):
    """Conv3d(4, C, k = s = 4) -> channels-last -> LayerNorm (swin_mae3d.py:1119-1131)"""
    S = _S()
    B, R, C = 2, 16, 96
    xb = rnd(B, 4, R, R, R)
    cw, cb = q(rnd(C, 4, 4, 4, 4, seed=1, scale=256 ** -0.5), dt), rnd(C, seed=2, scale=0.1)
    g, b = rnd(C, seed=3, scale=0.2) + 1.0, rnd(C, seed=4, scale=0.1)
    ref_in = [t.clone().requires_grad_(True) for t in (cw, cb, g, b)]
    xin = q(xb, dt)    # the gather kernel rounds the fp32 grid to the compute dtype
    yr = F.layer_norm(F.conv3d(xin, ref_in[0], ref_in[1], stride=4).permute(0, 2, 3, 4, 1), (C,), ref_in[2], ref_in[3], 1e-5)
    dtok = q(rnd(*yr.shape, seed=5), dt)
    yr.backward(dtok)
    tok, saved = S.patch_embed_fwd(dev(xb), dev(cw), dev(cb), dev(g), dev(b), dtype=dt)
    check(tok, yr, dt, "tokens", 2)
    for gk, tr, nm in zip(S.patch_embed_bwd(dev(dtok, dt), saved), ref_in, ("dconv_w", "dconv_b", "dln_g", "dln_b")):
        check(gk, tr.grad, dt, nm, 4)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("k,Cin,Cout,v", [(2, 32, 16, 5), (4, 96, 48, 3)])
def test_convT_k_eq_s(dt, k, Cin, Cout, v):
    """ConvTranspose3d(kernel = stride) (unetr_block.py:151-158)"""
    S = _S()
    B = 2
    x = q(rnd(B, v, v, v, Cin), dt)
    w, b = q(rnd(Cin, Cout, k, k, k, seed=1, scale=Cin ** -0.5), dt), rnd(Cout, seed=2, scale=0.1)
    ref_in = [t.clone().requires_grad_(True) for t in (x, w, b)]
    yr = F.conv_transpose3d(ref_in[0].permute(0, 4, 1, 2, 3), ref_in[1], ref_in[2], stride=k).permute(0, 2, 3, 4, 1)
    dy = q(rnd(*yr.shape, seed=3), dt)
    yr.backward(dy)
    y, saved = S.convT_k_eq_s_fwd(dev(x, dt), dev(w), dev(b), k)
    check(y, yr, dt, "y")
    for gk, tr, nm in zip(S.convT_k_eq_s_bwd(dev(dy, dt), saved), ref_in, ("dx", "dw", "db")):
        check(gk, tr.grad, dt, nm, 3)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Cin,Cout,shape", [(16, 24, (2, 6, 7, 8)), (48, 48, (1, 8, 16, 16))])
def test_conv3d_3x3x3(dt, Cin, Cout, shape):
    """nn.Conv3d(k = 3, padding = 1) forward, input gradient, weight gradient (unetr_block.py:35-44)"""
    S = _S()
    x = q(rnd(*shape, Cin), dt)
    w = q(rnd(Cout, Cin, 3, 3, 3, seed=1, scale=(27 * Cin) ** -0.5), dt)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr.permute(0, 4, 1, 2, 3), wr, padding=1).permute(0, 2, 3, 4, 1)
    dy = q(rnd(*yr.shape, seed=2), dt)
    yr.backward(dy)
    check(S.conv3d_3x3x3_fwd(dev(x, dt), dev(w)), yr, dt, "fwd")
    check(S.conv3d_3x3x3_dgrad(dev(dy, dt), dev(w)), xr.grad, dt, "dgrad", 2)
    check(S.conv3d_3x3x3_wgrad(dev(dy, dt), dev(x, dt)), wr.grad, dt, "wgrad", 2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("with_res", [False, True])
def test_instnorm_lrelu_add(dt, with_res):
    """InstanceNorm3d (+ residual) -> LeakyReLU(0.01) (unetr_block.py:57-71)"""
    S = _S()
    B, V, C = 2, 900, 48
    x = q(rnd(B, V, C) * 1.3 + 0.2, dt)
    r = q(rnd(B, V, C, seed=1), dt) if with_res else None
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if with_res else None
    yn = F.instance_norm(xr.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    yr = F.leaky_relu(yn + rr if with_res else yn, 0.01)
    dy = q(rnd(B, V, C, seed=2), dt)
    yr.backward(dy)
    y, saved = S.instnorm_lrelu_add_fwd(dev(x, dt), dev(r, dt) if with_res else None)
    check(y, yr, dt, "y")
    dx, dr = S.instnorm_lrelu_add_bwd(dev(dy, dt), saved)
    check(dx, xr.grad, dt, "dx", 3)
    if with_res:
        check(dr, rr.grad, dt, "dresidual", 3)


@pytest.mark.parametrize("dt", DTS)
def test_mae_loss(dt):
    """UnetOutBlock 1x1 conv + forward_loss (swin_mae3d.py:1513-1549) with a padded sample and a block mask"""
    from oracle import mae3d_oracle as O
    S = _S()
    B, R, Cd = 2, 32, 48
    x = torch.stack([O.synthetic_grid((R, R, R), 3), O.synthetic_grid((R, R, R), 4)])
    valid = torch.ones_like(x)
    valid[1, :, 28:] = 0
    x = x * valid
    ext = torch.tensor([[R, R, R], [28, R, R]], dtype=torch.int32)
    tm = O.draw_block_mask((R // 4,) * 3, 0.6, rng=random.Random(5))
    d0 = q(rnd(B, R, R, R, Cd), dt)
    wo, bo = rnd(4, Cd, seed=1, scale=0.2), rnd(4, seed=2, scale=0.1)
    ref_in = [t.clone().requires_grad_(True) for t in (d0, wo, bo)]
    pred_r = (ref_in[0] @ ref_in[1].T + ref_in[2]).permute(0, 4, 1, 2, 3)
    l, l_rgb, l_a, *_ = O.mae_loss(x, pred_r, valid, tm[None, ..., None].expand(B, -1, -1, -1, 1))
    l.backward()
    losses, pred, saved = S.mae_loss_fwd(dev(d0, dt), dev(wo), dev(bo), dev(x), dev(ext), dev(tm.to(torch.uint8)))
    check(pred, pred_r, dt, "pred")
    check(losses, torch.stack([l, l_rgb, l_a]), dt, "losses")
    for gk, tr, nm in zip(S.mae_loss_bwd(saved), ref_in, ("dd0", "dw_out", "db_out")):
        check(gk, tr.grad, dt, nm, 3)


def test_adamw_clip_step():
    """clip_grad_norm_ + torch.optim.AdamW on a flat buffer, three steps with changing lr / beta1 (run_swin_mae3d.py:665-669)"""
    S = _S()
    n = 10007
    p0, gs = rnd(n), [rnd(n, seed=s + 1) * (3.0 if s == 1 else 0.01) for s in range(3)]
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step, (g, lr, b1) in enumerate(zip(gs, (1e-3, 3e-3, 2e-3), (0.95, 0.9, 0.85)), 1):
        pr.grad = g.clone()
        nr = torch.nn.utils.clip_grad_norm_([pr], 0.1)
        for grp in opt.param_groups:
            grp["lr"], grp["betas"] = lr, (b1, 0.999)
        opt.step()
        nk = S.adamw_clip_step(p, g.clone().cuda(), m, v, step, lr, beta1=b1, max_grad_norm=0.1)
        assert abs(nk.item() - nr.item()) < 1e-5 * nr.item()
        assert torch.allclose(p.cpu(), pr.detach(), rtol=1e-5, atol=1e-7)
