"""GPU parity of every C-ABI kernel against the CPU oracle / plain fp32 math on the same seeded inputs.
fp32 mode (exact fp32 MFMA) must agree to 1e-3 relative (north_star tolerance; we assert much tighter),
bf16 mode to bf16 rounding.  Sizes are small enough that the CPU side finishes in seconds."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTS = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


def _ops():
    from nerf_mae_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 1000)
    return torch.randn(*shape, generator=g) * scale


def q(t, dt):
    """round to the storage dtype (so the reference sees the same inputs), back to fp32 on CPU"""
    return t.to(dt).float()


def dev(t, dt=None):
    return (t if dt is None else t.to(dt)).cuda().contiguous()


from tests._metrics import assert_close, rel_l2, relerr  # noqa: E402


def check(a, b, dt, name="", mult=1.0, elem_mult=1.0):
    """max-norm, relative-L2 and elementwise (atol = rtol * rms(ref)) against the same tolerance (tests/_metrics.py)"""
    assert_close(a, b, TOL[dt] * mult, f"{name} {dt}", elem_mult=elem_mult)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(300, 96, 96), (1000, 384, 96), (517, 288, 96), (2050, 48, 1296), (130, 768, 3072), (64, 64, 256), (4100, 3072, 96),
                                   (1000, 384, 1536), (200, 96, 776), (9000, 64, 384), (9000, 32, 512), (10000, 96, 384)])
def test_gemm_nt_plain_bias(dt, M, N, K):
    ops = _ops()
    A, W, b = q(rnd(M, K), dt), q(rnd(N, K, seed=1, scale=K ** -0.5), dt), rnd(N, seed=2)
    out = ops.gemm_nt(dev(A, dt), dev(W, dt), bias=dev(b))
    check(out, A @ W.T + b, dt, "gemm_nt")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K,rps", [(640, 384, 96, 320), (33130, 384, 96, 16565), (32800, 288, 128, 16400), (640, 384, 768, 320), (3200, 288, 512, 1600), (2500, 288, 512, 1250)],
                         ids=["small", "33k_rows_k96", "33k_rows_k128", "narrow_tiles_32col", "narrow_tiles_48col", "narrow_tiles_32col_ragged"])
def test_gemm_nt_epilogues(dt, M, N, K, rps):
    """fused epilogues of nmh_gemm_nt; the two large cases are short contractions on many rows (the 40^3-token Linears) with a ragged last
    row tile"""
    ops = _ops()
    A, W, b = q(rnd(M, K), dt), q(rnd(N, K, seed=1, scale=0.1), dt), rnd(N, seed=2, scale=0.1)
    # act 1: dual output gelu
    pre = torch.empty(M, N, dtype=dt, device="cuda")
    act = ops.gemm_nt(dev(A, dt), dev(W, dt), bias=dev(b), act=1, C2=pre)
    ref_pre = A @ W.T + b
    check(pre, ref_pre, dt, "pre")
    check(act, F.gelu(ref_pre), dt, "gelu")
    # act 2 + rowscale: v * gelu'(aux) * s
    aux = q(rnd(M, N, seed=5), dt)
    rs = torch.tensor([0.0, 1.0 / 0.9])
    out = ops.gemm_nt(dev(A, dt), dev(W, dt), act=2, C2=dev(aux, dt), rowscale=dev(rs), rows_per_scale=rps)
    xg = aux.clone().requires_grad_(True)
    F.gelu(xg).sum().backward()
    ref = (A @ W.T) * xg.grad * rs.repeat_interleave(rps)[:, None]
    check(out, ref, dt, "gelu_grad")
    # residual + rowscale + accumulate
    res = q(rnd(M, N, seed=6), dt)
    c0 = q(rnd(M, N, seed=7), dt)
    out = dev(c0, dt)
    ops.gemm_nt(dev(A, dt), dev(W, dt), bias=dev(b), resid=dev(res, dt), rowscale=dev(rs), rows_per_scale=rps, out=out, accumulate=True)
    ref = res + rs.repeat_interleave(rps)[:, None] * (A @ W.T + b) + c0
    check(out, ref, dt, "resid/accumulate")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(5000, 384, 96), (777, 96, 384), (3000, 48, 48), (900, 1536, 384), (2000, 4, 48)])
def test_gemm_tn(dt, M, N, K):
    ops = _ops()
    lda = (N + 7) // 8 * 8
    A = torch.zeros(M, lda)
    A[:, :N] = rnd(M, N)
    A, B = q(A, dt), q(rnd(M, K, seed=1), dt)
    rs = torch.tensor([0.5, 2.0, 1.0, 0.0, 1.5])
    rps = (M + 4) // 5
    dW = torch.full((N, K), 0.25, device="cuda")
    db = torch.full((N,), -0.5, device="cuda")
    ops.gemm_tn(dev(A, dt), dev(B, dt), dW, rowscale=dev(rs), rows_per_scale=rps, N=N, dbias=db)
    sc = rs.repeat_interleave(rps)[:M, None]
    As = q(A[:, :N] * sc, dt) if dt == torch.bfloat16 else A[:, :N] * sc
    ref = 0.25 + As.T @ B
    check(dW, ref, dt, "gemm_tn", mult=1.0 if dt == torch.float32 else 1.0)
    check(db, -0.5 + As.sum(0), dt, "gemm_tn fused bias grad")   # column sums of A from the same pass


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,D,H,W,Cin,Cout", [(2, 5, 6, 7, 16, 24), (1, 8, 8, 8, 48, 48), (1, 4, 9, 5, 96, 48), (1, 10, 10, 10, 24, 96), (2, 10, 10, 10, 192, 96), (1, 5, 6, 20, 96, 192), (2, 32, 32, 32, 96, 48), (2, 32, 32, 32, 48, 48)])
def test_conv3d_fwd_dgrad_wgrad(dt, B, D, H, W, Cin, Cout):
    ops = _ops()
    x = q(rnd(B, Cin, D, H, W), dt)
    w = q(rnd(Cout, Cin, 3, 3, 3, seed=1, scale=(27 * Cin) ** -0.5), dt)
    dy = q(rnd(B, Cout, D, H, W, seed=2), dt)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv3d(xr, wr, padding=1)
    y.backward(dy)
    xcl = dev(x.permute(0, 2, 3, 4, 1), dt)
    wp_f = dev(w.reshape(Cout, Cin, 27).permute(0, 2, 1), dt)                    # [Co][27][Ci]
    wp_d = dev(w.reshape(Cout, Cin, 27).flip(2).permute(1, 2, 0), dt)            # [Ci][27 flipped][Co]
    yk = ops.conv3d_k3(xcl, wp_f, Cout)
    check(yk.permute(0, 4, 1, 2, 3), y, dt, "conv fwd")
    dycl = dev(dy.permute(0, 2, 3, 4, 1), dt)
    dxk = ops.conv3d_k3(dycl, wp_d, Cin)
    check(dxk.permute(0, 4, 1, 2, 3), xr.grad, dt, "conv dgrad")
    dW = torch.zeros(Cout, Cin, 3, 3, 3, device="cuda")
    ops.conv3d_k3_wgrad(dycl, xcl, dW)
    check(dW, wr.grad, dt, "conv wgrad")


def _geom(ops, B, H, W, D, s):
    return ops.WinGeom(B, H, W, D, [s] * 3)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,shift", [((2, 8, 8, 8), 0), ((2, 8, 8, 8), 2), ((1, 5, 5, 5), 2), ((1, 10, 10, 10), 2), ((1, 2, 2, 2), 2), ((1, 6, 8, 4), 2)])
@pytest.mark.parametrize("C", [96, 192])
def test_layernorm_window_modes(dt, shape, shift, C):
    from oracle import mae3d_oracle as O
    ops = _ops()
    B, H, W, D = shape
    x = q(rnd(B, H, W, D, C), dt)
    gam, bet = 1 + 0.2 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    geom = _geom(ops, B, H, W, D, shift)
    T = geom.tokens
    # forward, window-ordered
    out = torch.empty(geom.rows, C, dtype=dt, device="cuda")
    mean, rstd = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    ops.layernorm_fwd(dev(x, dt), dev(gam), dev(bet), out, mean, rstd, geom.rows, C, src_mode=1, geom=geom)
    xr = x.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ln = F.layer_norm(xr, (C,), gr, br, 1e-5)
    pad = [g_ - s for g_, s in zip(geom.P, (H, W, D))]
    lp = F.pad(ln, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    if sum(geom.shift) > 0:
        lp = torch.roll(lp, shifts=[-s for s in geom.shift], dims=(1, 2, 3))
    ref = O.window_partition(lp).reshape(-1, C)
    check(out, ref, dt, "ln window fwd")
    # backward from a window-ordered gradient, with residual grad
    dyw = q(rnd(geom.rows, C, seed=3), dt)
    dres = q(rnd(T, C, seed=4), dt)
    (ref * dyw).sum().backward()
    dx = torch.empty(T, C, dtype=dt, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.layernorm_bwd(dev(dyw, dt), dev(x, dt), dev(gam), mean, rstd, dx, dg, db, T, C, src_mode=1, geom=geom, dres=dev(dres, dt))
    check(dx, xr.grad.reshape(T, C) + dres, dt, "ln window bwd dx", mult=2)
    check(dg, gr.grad, dt, "dgamma", mult=2)
    check(db, br.grad, dt, "dbeta", mult=2)
    # scatter/gather adjoint pair
    yw = q(rnd(geom.rows, C, seed=5), dt)
    rs = torch.tensor([1.0 / 0.9, 0.0][:B])
    o2 = torch.empty(T, C, dtype=dt, device="cuda")
    ops.window_scatter_residual(dev(yw, dt), dev(x.reshape(T, C), dt), o2, dev(rs), C, geom)
    yv = O.window_reverse(yw.view(-1, 64, C), B, *geom.P)
    if sum(geom.shift) > 0:
        yv = torch.roll(yv, shifts=list(geom.shift), dims=(1, 2, 3))
    ref2 = x + rs.view(B, 1, 1, 1, 1) * yv[:, :H, :W, :D]
    check(o2, ref2.reshape(T, C), dt, "window scatter")
    g2 = torch.empty(geom.rows, C, dtype=dt, device="cuda")
    ops.window_gather_scale(dev(x.reshape(T, C), dt), g2, dev(rs), C, geom)
    xs = x * rs.view(B, 1, 1, 1, 1)
    xp = F.pad(xs, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    if sum(geom.shift) > 0:
        xp = torch.roll(xp, shifts=[-s for s in geom.shift], dims=(1, 2, 3))
    check(g2, O.window_partition(xp).reshape(-1, C), dt, "window gather")
    # the product path: the window reverse is the store of the proj GEMM, the window gather a second output of the LN backward
    Wp, bp = q(rnd(C, C, seed=6, scale=C ** -0.5), dt), rnd(C, seed=7, scale=0.1)
    o3 = torch.empty(T, C, dtype=dt, device="cuda")
    ops.gemm_nt_window_scatter(dev(yw, dt), dev(Wp, dt), o3, dev(x.reshape(T, C), dt), dev(bp), dev(rs), T // B, geom)
    pv = O.window_reverse((yw @ Wp.T + bp).view(-1, 64, C), B, *geom.P)
    if sum(geom.shift) > 0:
        pv = torch.roll(pv, shifts=list(geom.shift), dims=(1, 2, 3))
    check(o3, (x + rs.view(B, 1, 1, 1, 1) * pv[:, :H, :W, :D]).reshape(T, C), dt, "proj + window scatter + residual", 2)
    dy0 = q(rnd(T, C, seed=8), dt)
    dx0, dx1 = torch.empty(T, C, dtype=dt, device="cuda"), torch.empty(T, C, dtype=dt, device="cuda")
    dgs = [torch.zeros(C, device="cuda") for _ in range(4)]
    mean0, rstd0 = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    ln0 = torch.empty(T, C, dtype=dt, device="cuda")
    ops.layernorm_fwd(dev(x.reshape(T, C), dt), dev(gam), dev(bet), ln0, mean0, rstd0, T, C)
    ops.layernorm_bwd(dev(dy0, dt), dev(x.reshape(T, C), dt), dev(gam), mean0, rstd0, dx0, dgs[0], dgs[1], T, C, dres=dev(dres, dt))
    g_ref = torch.empty(geom.rows, C, dtype=dt, device="cuda")
    ops.window_gather_scale(dx0, g_ref, dev(rs), C, geom)
    g_fused = torch.full((geom.rows, C), 3.0, dtype=dt, device="cuda")
    ops.layernorm_bwd(dev(dy0, dt), dev(x.reshape(T, C), dt), dev(gam), mean0, rstd0, dx1, dgs[2], dgs[3], T, C, dres=dev(dres, dt),
                      geom=geom, tokens_per_sample=T // B, dyw=g_fused, dyw_scale=dev(rs))
    check(dx1, dx0.float().cpu(), dt, "ln bwd dx unchanged by the fused gather")
    check(g_fused, g_ref.float().cpu(), dt, "ln bwd fused window gather", 2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C", [96, 192, 384, 128, 512, 1024])   # (128 / 512 / 1024: the 128-channel stem; 512 takes the 32-lanes-per-row backward instantiation)
def test_layernorm_plain_and_embed_post(dt, C):
    ops = _ops()
    B, tps = 2, 130
    T = B * tps
    x = q(rnd(T, C), dt)
    gam, bet = 1 + 0.2 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    pos, mt = rnd(tps, C, seed=3), rnd(C, seed=4, scale=0.1)
    mask = (torch.arange(tps) % 3 == 0).to(torch.uint8)
    out = torch.empty(T, C, dtype=dt, device="cuda")
    mean, rstd = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    ops.layernorm_fwd(dev(x, dt), dev(gam), dev(bet), out, mean, rstd, T, C, pos=dev(pos), mask=dev(mask), mask_token=dev(mt), tokens_per_sample=tps)
    xr, gr, br, mr = [t.clone().requires_grad_(True) for t in (x, gam, bet, mt)]
    ln = F.layer_norm(xr, (C,), gr, br, 1e-5).view(B, tps, C) + pos
    ref = torch.where(mask.bool()[None, :, None], mr.view(1, 1, C), ln).reshape(T, C)
    check(out, ref, dt, "embed post fwd")
    dy = q(rnd(T, C, seed=5), dt)
    (ref * dy).sum().backward()
    dx = torch.empty(T, C, dtype=dt, device="cuda")
    dg, db, dm = [torch.zeros(C, device="cuda") for _ in range(3)]
    ops.layernorm_bwd(dev(dy, dt), dev(x, dt), dev(gam), mean, rstd, dx, dg, db, T, C, mask=dev(mask), dmask_token=dm, tokens_per_sample=tps)
    check(dx, xr.grad, dt, "dx", 2)
    check(dg, gr.grad, dt, "dgamma", 2)
    check(db, br.grad, dt, "dbeta", 2)
    check(dm, mr.grad, dt, "dmask_token", 2)


class _Defer:
    """stands in for ops.WgradQueue: keeps the queued closures, runs them on request"""

    def __init__(self):
        self.fns = []

    def defer(self, fn):
        self.fns.append(fn)

    def add_ln_partials(self, *item):
        self.fns.append(lambda: _ops().ln_param_grad_reduce([item]))

    def run(self):
        for fn in self.fns:
            fn()
        self.fns = []


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,C,shift", [((2, 8, 8, 8), 96, 2), ((1, 5, 5, 5), 192, 2), ((8, 10, 10, 10), 384, 2), ((2, 10, 10, 10), 384, 0), ((3, 40, 12, 40), 96, 2), ((1, 5, 5, 5), 768, 0), ((2, 10, 10, 10), 512, 2)])
def test_layernorm_bwd_deferred_parameter_gradients(dt, shape, C, shift):
    """the parameter gradients as per-workgroup partial sums + a later column-sum launch (nmh_layernorm_bwd_deferred / _param_grad_reduce) against the
    atomic form of the same kernel: identical dx / dyw, dgamma / dbeta equal up to the fp32 summation order, accumulated INTO their buffers"""
    ops = _ops()
    B, H, W, D = shape
    geom = _geom(ops, B, H, W, D, shift)
    T, tps = geom.tokens, geom.tokens // B
    x, dres = q(rnd(T, C, seed=1), dt), q(rnd(T, C, seed=2), dt)
    gam = 1 + 0.2 * rnd(C, seed=3)
    mean, rstd = dev(x.mean(-1)), dev((x.var(-1, unbiased=False) + 1e-5).rsqrt())
    sc = dev(torch.rand(B) + 0.5)
    for mode in (0, 1):
        dy = q(rnd(T if mode == 0 else geom.rows, C, seed=4 + mode), dt)
        res = []
        for deferred in (False, True):
            dq = _Defer()
            dx = torch.empty(T, C, dtype=dt, device="cuda")
            dg, db = torch.full((C,), 0.25, device="cuda"), torch.full((C,), -0.5, device="cuda")
            dyw = torch.full((geom.rows, C), 7.0, dtype=dt, device="cuda") if mode == 0 else None
            ops.layernorm_bwd(dev(dy, dt), dev(x, dt), dev(gam), mean, rstd, dx, dg, db, T, C, src_mode=mode, geom=geom, dres=dev(dres, dt),
                              tokens_per_sample=tps, dyw=dyw, dyw_scale=sc if mode == 0 else None, wq=dq if deferred else None)
            if deferred:
                assert len(dq.fns) == 1 and float(dg[0]) == 0.25      # nothing added yet
                dq.run()
            torch.cuda.synchronize()
            res.append((dx, dyw, dg, db))
        assert torch.equal(res[0][0], res[1][0]), f"dx differs (mode {mode})"
        if mode == 0:
            assert torch.equal(res[0][1], res[1][1]), "dyw differs"
        check(res[1][2], res[0][2].cpu(), torch.float32, f"dgamma (mode {mode})", 5)
        check(res[1][3], res[0][3].cpu(), torch.float32, f"dbeta (mode {mode})", 5)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,C", [((2, 8, 8, 8), 96), ((1, 5, 5, 5), 96), ((1, 6, 5, 4), 192), ((1, 3, 3, 3), 384)])
def test_patch_merge_layernorm(dt, shape, C):
    from oracle import mae3d_oracle as O
    ops = _ops()
    B, H, W, D = shape
    x = q(rnd(B, H, W, D, C), dt)
    gam, bet = 1 + 0.2 * rnd(8 * C, seed=1), 0.1 * rnd(8 * C, seed=2)
    geom = ops.WinGeom(B, H, W, D, [0, 0, 0])
    H2, W2, D2 = (H + 1) // 2, (W + 1) // 2, (D + 1) // 2
    rows = B * H2 * W2 * D2
    out = torch.empty(rows, 8 * C, dtype=dt, device="cuda")
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd(dev(x, dt), dev(gam), dev(bet), out, mean, rstd, rows, 8 * C, src_mode=2, geom=geom)
    xr, gr, br = [t.clone().requires_grad_(True) for t in (x, gam, bet)]
    ref = F.layer_norm(O.patch_merge_gather(xr), (8 * C,), gr, br, 1e-5).reshape(rows, 8 * C)
    check(out, ref, dt, "merge ln fwd")
    dy = q(rnd(rows, 8 * C, seed=3), dt)
    (ref * dy).sum().backward()
    dx = torch.zeros(B * H * W * D, C, dtype=dt, device="cuda")
    dg, db = torch.zeros(8 * C, device="cuda"), torch.zeros(8 * C, device="cuda")
    ops.layernorm_bwd(dev(dy, dt), dev(x, dt), dev(gam), mean, rstd, dx, dg, db, rows, 8 * C, src_mode=2, geom=geom)
    check(dx, xr.grad.reshape(-1, C), dt, "merge dx", 2)
    check(dg, gr.grad, dt, "merge dgamma", 2)
    check(db, br.grad, dt, "merge dbeta", 2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,shift,heads", [((2, 8, 8, 8), 0, 3), ((2, 8, 8, 8), 2, 3), ((1, 5, 5, 5), 2, 6), ((1, 10, 10, 10), 2, 3), ((1, 2, 2, 2), 2, 12), ((1, 6, 8, 4), 2, 3)])
@pytest.mark.parametrize("qsplit", [1, 2, 4])
def test_window_attention_core(dt, shape, shift, heads, qsplit, monkeypatch):
    """attention core on window-ordered qkv vs the oracle's softmax path (bias + mask + pads as keys); the forward with a (window, head)
    pair on one wave or split by query tiles over 2 / 4 waves (NMH_ATTN_QSPLIT; the default picks 4 for <= 768 pairs, else 2)."""
    from oracle import mae3d_oracle as O
    ops = _ops()
    monkeypatch.setenv("NMH_ATTN_QSPLIT", str(qsplit))
    monkeypatch.setenv("NMH_ATTN_BNW", "4" if qsplit == 4 else "2")   # backward: 4 / 2 (window, head) pairs per workgroup
    B, H, W, D = shape
    C = heads * 32
    geom = _geom(ops, B, H, W, D, shift)
    nW = geom.rows // 64 // B
    qkv = q(rnd(geom.rows, 3 * C, scale=1.5), dt)
    table = rnd(343, heads, seed=1, scale=0.5)
    dout = q(rnd(geom.rows, C, seed=2), dt)
    qr, tr = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
    t = qr.view(-1, 64, 3, heads, 32).permute(2, 0, 3, 1, 4)
    attn = (t[0] * 32 ** -0.5) @ t[1].transpose(-2, -1)
    attn = attn + tr[O.rel_pos_index(4)].view(64, 64, heads).permute(2, 0, 1).unsqueeze(0)
    if sum(geom.shift) > 0:
        ids = O.window_partition(O.shift_region_ids(geom.P, geom.shift)[None, ..., None]).view(nW, 64)
        am = torch.zeros(nW, 64, 64).masked_fill((ids[:, None, :] - ids[:, :, None]) != 0, -100.0)
        attn = (attn.view(B, nW, heads, 64, 64) + am[None, :, None]).view(-1, heads, 64, 64)
    p = attn.softmax(-1)
    ref = (p @ t[2]).transpose(1, 2).reshape(-1, C)
    ref.backward(dout)
    out = torch.empty(geom.rows, C, dtype=dt, device="cuda")
    lse = torch.empty(geom.rows * heads, device="cuda")
    ops.window_attn_fwd(dev(qkv, dt), dev(table), out, lse, heads, C, geom)
    check(out, ref, dt, "attn fwd")
    dqkv = torch.empty(geom.rows, 3 * C, dtype=dt, device="cuda")
    dtab = torch.zeros(343, heads, device="cuda")
    ops.window_attn_bwd(dev(qkv, dt), dev(table), dev(dout, dt), lse, dqkv, dtab, heads, C, geom)
    check(dqkv, qr.grad, dt, "attn dqkv", 2)
    check(dtab, tr.grad, dt, "attn dbias", 2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,V,C,rmode", [(2, 1000, 48, 0), (2, 343, 96, 1), (1, 5000, 48, 1), (2, 216, 384, 2), (1, 8 ** 3, 192, 2), (2, 32 ** 3, 48, 2), (2, 32 ** 3, 48, 0),
                                         (1, 160 ** 3, 48, 0), (1, 150 ** 3, 48, 2), (1, 118 ** 3, 96, 1)],
                         ids=["1000x48", "343x96_r1", "5000x48_r1", "216x384_r2", "512x192_r2", "32cube_r2", "32cube", "160cube_streaming", "150cube_r2_streaming_ragged",
                              "118cube_c96_r1_streaming_ragged"])
def test_instnorm_fwd_bwd(dt, B, V, C, rmode):
    """the last three cases are bf16 tensors above the 300-MB threshold of the streaming form of the apply passes (short blocks, per-channel constants
    through an LDS table, non-temporal accesses: csrc/norm.hip in_apply_streaming), two of them with a ragged last block"""
    ops = _ops()
    x = q(rnd(B, V, C) * 1.5 + 0.3, dt)
    r = q(rnd(B, V, C, seed=1), dt) if rmode else None
    dout = q(rnd(B, V, C, seed=2), dt)
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if rmode else None
    inorm = lambda t: F.instance_norm(t.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    pre = inorm(xr) + (rr if rmode == 1 else inorm(rr) if rmode == 2 else 0)
    ref = F.leaky_relu(pre, 0.01)
    ref.backward(dout)
    xd = dev(x, dt)
    stats, scratch = torch.empty(B, C, 2, device="cuda"), torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
    ops.instnorm_stats(xd, stats, scratch, B, V, C)
    rd, stats_r = (dev(r, dt) if rmode else None), None
    if rmode == 2:
        stats_r = torch.empty(B, C, 2, device="cuda")
        ops.instnorm_stats(rd, stats_r, scratch, B, V, C)
    out = torch.empty(B, V, C, dtype=dt, device="cuda")
    ops.instnorm_apply(xd, stats, out, B, V, C, r=rd, stats_r=stats_r, rmode=rmode)
    check(out, ref, dt, "in fwd")
    out_ref_dev = dev(ref, dt)  # use the exact forward output so the lrelu sign pattern matches
    sums = torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
    sums_r = torch.empty(B, C, 2, dtype=torch.float64, device="cuda") if rmode == 2 else None
    dd = dev(dout, dt)
    ops.instnorm_bwd_reduce(dd, out_ref_dev, xd, stats, sums, B, V, C, r=rd, stats_r=stats_r, sums_r=sums_r, rmode=rmode)
    dx = torch.empty(B, V, C, dtype=dt, device="cuda")
    dr = torch.empty(B, V, C, dtype=dt, device="cuda") if rmode else None
    ops.instnorm_bwd_apply(dd, out_ref_dev, xd, stats, sums, dx, B, V, C, r=rd, stats_r=stats_r, sums_r=sums_r, rmode=rmode, dr=dr)
    check(dx, xr.grad, dt, "in dx", 3)
    if rmode:
        check(dr, rr.grad, dt, "in dr", 3)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("k,Cin,Cout,v,skip", [(2, 96, 48, 3, True), (4, 48, 24, 2, False), (2, 768, 384, 2, True), (4, 96, 48, 8, True)])
def test_upconv_block_pieces(dt, k, Cin, Cout, v, skip):
    """ConvTranspose3d(k=s) = GEMM + pixel shuffle (+bias, +skip concat); backward pieces incl. weight grad remap."""
    ops = _ops()
    B = 2
    x = q(rnd(B, Cin, v, v, v), dt)
    w = q(rnd(Cin, Cout, k, k, k, seed=1, scale=Cin ** -0.5), dt)
    b = rnd(Cout, seed=2, scale=0.1)
    V = v * k
    sk = q(rnd(B, Cout, V, V, V, seed=3), dt) if skip else None
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    skr = sk.clone().requires_grad_(True) if skip else None
    y = F.conv_transpose3d(xr, wr, br, stride=k)
    cat = torch.cat((y, skr), 1) if skip else y
    dcat = q(rnd(*cat.shape, seed=4), dt)
    cat.backward(dcat)
    k3 = k ** 3
    xcl = dev(x.permute(0, 2, 3, 4, 1).reshape(-1, Cin), dt)
    wf = dev(w.reshape(Cin, Cout, k3).permute(2, 1, 0).reshape(k3 * Cout, Cin), dt)     # [(t,co)][ci]
    wd = dev(w.reshape(Cin, Cout, k3).permute(0, 2, 1).reshape(Cin, k3 * Cout), dt)     # [ci][(t,co)]
    upre = ops.gemm_nt(xcl, wf)
    Cc = 2 * Cout if skip else Cout
    out = torch.empty(B * V ** 3, Cc, dtype=dt, device="cuda")
    skcl = dev(sk.permute(0, 2, 3, 4, 1).reshape(-1, Cout), dt) if skip else None
    ops.upconv_shuffle_fwd(upre, dev(b), skcl, out, B, v, k, Cout)
    check(out.view(B, V, V, V, Cc).permute(0, 4, 1, 2, 3), cat, dt, "upconv fwd")
    dc = dev(dcat.permute(0, 2, 3, 4, 1).reshape(-1, Cc), dt)
    dupre = torch.empty(B * v ** 3, k3 * Cout, dtype=dt, device="cuda")
    dskip = torch.empty(B * V ** 3, Cout, dtype=dt, device="cuda") if skip else None
    dbias = torch.zeros(Cout, device="cuda")
    ops.upconv_shuffle_bwd(dc, dupre, dskip, dbias, B, v, k, Cout, skip)
    check(dbias, br.grad, dt, "upconv dbias", 2)
    if skip:
        check(dskip.view(B, V, V, V, Cout).permute(0, 4, 1, 2, 3), skr.grad, dt, "dskip")
    dx = ops.gemm_nt(dupre, wd)
    check(dx.view(B, v, v, v, Cin).permute(0, 4, 1, 2, 3), xr.grad, dt, "upconv dx", 2)
    dW = torch.zeros(Cin, Cout, k, k, k, device="cuda")
    ops.gemm_tn(dupre, xcl, dW, omode=2, p0=Cout, p1=k3)
    check(dW, wr.grad, dt, "upconv dW", 2)
    # the product path: pixel shuffle folded into the GEMM epilogue / operand loaders
    out2 = torch.zeros(B * V ** 3, Cc, dtype=dt, device="cuda")
    ops.upconv_fwd(xcl, wf, dev(b), out2, B, v, k, Cin, Cout)
    check(out2[:, :Cout].float().cpu().view(B, V, V, V, Cout).permute(0, 4, 1, 2, 3), y.detach(), dt, "fused upconv fwd")
    if skip:
        assert out2[:, Cout:].abs().max().item() == 0.0   # the skip half is left to the caller
    dx2 = torch.empty(B * v ** 3, Cin, dtype=dt, device="cuda")
    ops.upconv_dgrad(dc, wd, dx2, B, v, k, Cin, Cout)
    check(dx2.view(B, v, v, v, Cin).permute(0, 4, 1, 2, 3), xr.grad, dt, "fused upconv dx", 2)
    dW2, db2 = torch.zeros(Cin, Cout, k, k, k, device="cuda"), torch.zeros(Cout, device="cuda")
    ops.upconv_wgrad(dc, xcl, dW2, db2, B, v, k, Cin, Cout)
    check(dW2, wr.grad, dt, "fused upconv dW", 2)
    check(db2, br.grad, dt, "fused upconv dbias", 2)
    # the shuffled-view gemm_tn entry itself (ops.upconv_wgrad dispatches bf16 gradients without a skip half to the grouped kernel)
    from nerf_mae_amd._lib import lib
    dW4, db4 = torch.zeros(Cin, Cout, k, k, k, device="cuda"), torch.zeros(Cout, device="cuda")
    lib().call("nmh_upconv_wgrad", ops.dt_of(dc), dc, dc.stride(0), xcl, dW4, db4, B, v, k, Cin, Cout, ops._st())
    check(dW4, wr.grad, dt, "shuffled-view upconv dW", 2)
    check(db4, br.grad, dt, "shuffled-view upconv dbias", 2)
    if dt == torch.bfloat16:   # the grouped kernel: k^2 folded (tx, co) problems (contiguous runs without a skip half, k strided pieces with one)
        dW3, db3 = torch.zeros(Cin, Cout, k, k, k, device="cuda"), torch.zeros(Cout, device="cuda")
        ops.upconv_wgrad_grouped(dc, xcl, dW3, db3, B, v, k, Cin, Cout)
        check(dW3, wr.grad, dt, "grouped upconv dW", 2)
        check(db3, br.grad, dt, "grouped upconv dbias", 2)


@pytest.mark.parametrize("dt", DTS)
def test_loss_head(dt):
    from oracle import mae3d_oracle as O
    ops = _ops()
    B, R, Cd = 2, 32, 48
    x = torch.stack([O.synthetic_grid((32, 32, 32), 3), O.synthetic_grid((32, 32, 32), 4)])
    valid = torch.ones_like(x)
    valid[1, :, 28:] = 0
    valid[1, :, :, 24:] = 0
    x = x * valid
    ext = torch.tensor([[32, 32, 32], [28, 24, 32]], dtype=torch.int32)
    d0 = q(rnd(B, R, R, R, Cd), dt)
    Wo, bo = rnd(4, Cd, seed=1, scale=0.2), rnd(4, seed=2, scale=0.1)
    tm = O.draw_block_mask((8, 8, 8), 0.6, rng=__import__("random").Random(5))
    d0r, Wr, br_ = d0.clone().requires_grad_(True), Wo.clone().requires_grad_(True), bo.clone().requires_grad_(True)
    pred = (d0r @ Wr.T + br_).permute(0, 4, 1, 2, 3)
    tmask = tm[None, ..., None].expand(B, -1, -1, -1, 1)
    l, lr, la, *_ = O.mae_loss(x, pred, valid, tmask)
    l.backward()
    sums = torch.empty(4, dtype=torch.float64, device="cuda")
    losses = torch.empty(3, device="cuda")
    predk = torch.empty(B, 4, R, R, R, device="cuda")
    args = (dev(d0, dt), dev(Wo), dev(bo), dev(x), dev(ext), dev(tm.to(torch.uint8)), B, R, Cd, sums)
    ops.mae_loss_fwd(*args, losses, predk)
    check(predk, pred, dt, "pred")
    np.testing.assert_allclose(losses.cpu().numpy(), [l.item(), lr.item(), la.item()], rtol=TOL[dt])
    dd0 = torch.empty(B * R ** 3, Cd, dtype=dt, device="cuda")
    dp8 = torch.empty(B * R ** 3, 8, dtype=dt, device="cuda")
    dW, db = torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda")
    ops.mae_loss_bwd(*args, dd0, dp8, dW, db)
    check(dd0, d0r.grad.reshape(-1, Cd), dt, "dd0", 2)
    check(dW, Wr.grad, dt, "dWout", 2)
    check(db, br_.grad, dt, "dbout", 2)


@pytest.mark.parametrize("dt,R", [(d, 32) for d in DTS] + [(torch.bfloat16, 120)], ids=lambda v: str(v).replace("torch.", ""))
def test_tail_backward_fused(dt, R):
    """d0 = lrelu(IN(y) + r) -> 1x1 head -> loss: the two-pass fused backward (d(d0) never stored) against autograd.  R = 120 (bf16): 2 x 120^3 x 48
    is above the 300-MB threshold of the streaming form of the apply pass (csrc/norm.hip: tail_bwd_kernel<.., ST, NT>)."""
    from oracle import mae3d_oracle as O
    ops = _ops()
    B, Cd = 2, 48
    V = R ** 3
    x = torch.stack([O.synthetic_grid((R, R, R), 3), O.synthetic_grid((R, R, R), 4)])
    valid = torch.ones_like(x)
    valid[1, :, R - 4:] = 0
    x = x * valid
    ext = torch.tensor([[R, R, R], [R - 4, R, R]], dtype=torch.int32)
    y, r = q(rnd(B, V, Cd) * 1.3 + 0.2, dt), q(rnd(B, V, Cd, seed=7), dt)
    Wo, bo = rnd(4, Cd, seed=1, scale=0.2), rnd(4, seed=2, scale=0.1)
    tm = O.draw_block_mask((R // 4,) * 3, 0.6, rng=__import__("random").Random(5))
    yr, rr, Wr, br_ = (t.clone().requires_grad_(True) for t in (y, r, Wo, bo))
    inorm = lambda t: F.instance_norm(t.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    d0_ref = F.leaky_relu(inorm(yr) + rr, 0.01)
    d0q = q(d0_ref.detach(), dt)                       # the kernels see the stored (rounded) forward output
    d0s = d0_ref + (d0q - d0_ref).detach()
    pred = (d0s.reshape(B, R, R, R, Cd) @ Wr.T + br_).permute(0, 4, 1, 2, 3)
    l, *_ = O.mae_loss(x, pred, valid, tm[None, ..., None].expand(B, -1, -1, -1, 1))
    l.backward()
    yd = dev(y, dt)
    stats, scratch = torch.empty(B, Cd, 2, device="cuda"), torch.empty(B, Cd, 2, dtype=torch.float64, device="cuda")
    ops.instnorm_stats(yd, stats, scratch, B, V, Cd)
    lsums, losses = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda")
    args = (dev(d0q.reshape(-1, Cd), dt), dev(Wo), dev(bo), dev(x), dev(ext), dev(tm.to(torch.uint8)), B, R, Cd, lsums)
    dpred = torch.empty(B * V, 4, device="cuda")
    ops.mae_loss_fwd(*args, losses, None, dpred)
    np.testing.assert_allclose(losses[0].item(), l.item(), rtol=TOL[dt])
    # one-pass forward tail (normalise + residual + LeakyReLU + head + loss terms) against the two separate kernels
    lsums2, losses2 = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda")
    d0k = torch.empty(B * V, Cd, dtype=dt, device="cuda")
    dpred2, predk = torch.empty(B * V, 4, device="cuda"), torch.empty(B, 4, R, R, R, device="cuda")
    ops.mae_tail_fwd(yd.view(-1, Cd), stats, dev(r, dt).view(-1, Cd), d0k, args[1], args[2], args[3], args[4], args[5], B, R, Cd, lsums2, losses2,
                     predk, dpred2)
    check(d0k, d0_ref.detach().reshape(-1, Cd), dt, "tail fwd d0")
    check(predk, pred.detach(), dt, "tail fwd pred", 2)
    np.testing.assert_allclose(losses2.cpu().numpy(), losses.cpu().numpy(), rtol=TOL[dt])
    check(dpred2, dpred.cpu(), dt, "tail fwd dpred", 3)
    np.testing.assert_allclose(lsums2.cpu().numpy(), lsums.cpu().numpy(), rtol=50 * TOL[dt], atol=1e-3)
    dy, dr = torch.empty(B * V, Cd, dtype=dt, device="cuda"), torch.empty(B * V, Cd, dtype=dt, device="cuda")
    dW, db = torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda")
    in_sums = torch.empty(B, Cd, 2, dtype=torch.float64, device="cuda")
    ops.mae_tail_bwd(args[0], yd.view(-1, Cd), stats, dpred, lsums, args[1], in_sums, dy, dr, dW, db, B, V, Cd)
    check(dr, rr.grad.reshape(-1, Cd), dt, "tail dr", 3)
    check(dy, yr.grad.reshape(-1, Cd), dt, "tail dy", 3)
    check(dW, Wr.grad, dt, "tail dWout", 2)
    check(db, br_.grad, dt, "tail dbout", 2)
    # and the unfused kernels give the same thing
    dd0 = torch.empty(B * V, Cd, dtype=dt, device="cuda")
    dW2, db2 = torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda")
    ops.mae_loss_bwd(*args, dd0, torch.empty(B * V, 8, dtype=dt, device="cuda"), dW2, db2)
    sums2 = torch.empty_like(in_sums)
    rd = dev(r, dt)
    ops.instnorm_bwd_reduce(dd0, args[0], yd, stats, sums2, B, V, Cd, r=rd, rmode=1)
    dy2, dr2 = torch.empty_like(dy), torch.empty_like(dr)
    ops.instnorm_bwd_apply(dd0, args[0], yd, stats, sums2, dy2, B, V, Cd, r=rd, rmode=1, dr=dr2)
    check(dy, dy2.float().cpu(), dt, "fused vs unfused dy", 3)
    check(dW, dW2.cpu(), dt, "fused vs unfused dW", 2)
    # d0 not stored: the forward skips the write, the backward rebuilds d0 from (y, stats, r) -- identical results
    lsums3, losses3, dpred3 = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.empty(B * V, 4, device="cuda")
    ops.mae_tail_fwd(yd.view(-1, Cd), stats, rd.view(-1, Cd), None, args[1], args[2], args[3], args[4], args[5], B, R, Cd, lsums3, losses3, None, dpred3)
    assert torch.equal(dpred3, dpred2) and torch.allclose(losses3, losses2, rtol=1e-6)
    dy3, dr3 = torch.empty_like(dy), torch.empty_like(dr)
    dW3, db3 = torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda")
    in_sums3 = torch.empty_like(in_sums)
    in_sumsk = torch.empty_like(in_sums)
    dyk, drk, dWk, dbk = torch.empty_like(dy), torch.empty_like(dr), torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda")
    ops.mae_tail_bwd(d0k, yd.view(-1, Cd), stats, dpred2, lsums2, args[1], in_sumsk, dyk, drk, dWk, dbk, B, V, Cd)          # stored d0 of the fused forward
    ops.mae_tail_bwd(None, yd.view(-1, Cd), stats, dpred2, lsums2, args[1], in_sums3, dy3, dr3, dW3, db3, B, V, Cd, r=rd.view(-1, Cd))
    assert torch.equal(dr3, drk)                                                        # elementwise: bit-identical
    assert torch.allclose(dy3.float(), dyk.float(), rtol=1e-2 if dt == torch.bfloat16 else 1e-4, atol=1e-6)   # goes through the fp64-atomic channel sums
    assert torch.allclose(dW3, dWk, rtol=1e-5, atol=1e-6)
    # reductions of the backward taken by the forward pass (bwd_sums): the backward is one apply pass, same gradients
    lsums4, losses4, dpred4 = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.empty(B * V, 4, device="cuda")
    bsum = torch.empty(B * Cd * 4 + 4 * Cd, dtype=torch.float64, device="cuda")
    ops.mae_tail_fwd(yd.view(-1, Cd), stats, rd.view(-1, Cd), None, args[1], args[2], args[3], args[4], args[5], B, R, Cd, lsums4, losses4, None, dpred4, bwd_sums=bsum)
    mfma = dt == torch.bfloat16 and Cd == 48   # matrix-core variant of the pass: the head dot products are summed in another order, the reduction
    #                                            operands (d(pred), x-hat) enter the MFMAs rounded to bf16
    if mfma:
        # (d0 enters the head MFMAs as bf16 and the weights as bf16 hi + lo: |error of pred| <~ 2^-17 sum |w d0| ~ 5e-4; a d0 element whose fp32 value sits on a
        #  bf16 tie can land on either side depending on whether the compiler contracts (y - mean) * rstd + r: one ulp of one input, same size)
        #  -- so the bound on a single element is one bf16 ulp of the largest |d0| through the largest head weight (x 2: d(pred) = 2 (pred - target)), and
        #  all but a vanishing fraction of the elements agree to 1e-3 (at 2 x 120^3 voxels a handful of tie cases exist, at 32^3 usually none)
        diff = (dpred4 - dpred2).abs()
        tie = 2.0 * 2.0 ** -8 * d0k.float().abs().max().item() * args[1].abs().max().item()
        assert diff.max().item() <= max(1e-3, tie), ("dpred", diff.max().item(), tie, dpred2.abs().max().item())
        assert (diff > 1e-3 + 1e-4 * dpred2.abs()).float().mean().item() < 1e-5, ("dpred outliers", (diff > 1e-3 + 1e-4 * dpred2.abs()).float().mean().item())
        assert torch.allclose(losses4, losses2, rtol=1e-5)   # head weights as bf16 hi+lo: 2^-17
    else:
        assert torch.equal(dpred4, dpred2) and torch.allclose(losses4, losses2, rtol=1e-6)
    dy4, dr4, dW4, db4, in_sums4 = torch.empty_like(dy), torch.empty_like(dr), torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda"), torch.empty_like(in_sums)
    ops.mae_tail_bwd(None, yd.view(-1, Cd), stats, dpred4, lsums4, args[1], in_sums4, dy4, dr4, dW4, db4, B, V, Cd, r=rd.view(-1, Cd), bwd_sums=bsum)
    if mfma:
        import os
        if os.environ.get("NMH_TEST_VERBOSE"):
            print("tail mfma: in_sums rel", ((in_sums4 - in_sums3).abs().max() / in_sums3.abs().max()).item(), "dW rel", ((dW4 - dW3).abs().max() / dW3.abs().max()).item(),
                  "dr mismatches", (dr4 != dr3).float().mean().item(), "dy rel", ((dy4.float() - dy3.float()).abs().max() / dy3.float().abs().max()).item())
        assert (dr4 != dr3).float().mean().item() < 1e-3 and torch.allclose(dr4.float(), dr3.float(), rtol=1e-2, atol=1e-6)   # bf16 ties of the last place only
        scale = in_sums3.abs().max().item()
        assert torch.allclose(in_sums4, in_sums3, rtol=2e-3, atol=2e-4 * scale)
        assert torch.allclose(dW4, dW3, rtol=2e-3, atol=2e-4 * dW3.abs().max().item()) and torch.allclose(db4, db3, rtol=1e-5, atol=1e-7)
    else:
        assert torch.equal(dr4, dr3)
        assert torch.allclose(in_sums4, in_sums3, rtol=1e-4, atol=1e-7)
        assert torch.allclose(dW4, dW3, rtol=1e-4, atol=1e-6) and torch.allclose(db4, db3, rtol=1e-5, atol=1e-7)
    assert torch.allclose(dy4.float(), dy3.float(), rtol=1e-2 if dt == torch.bfloat16 else 1e-4, atol=1e-6)
    if mfma:
        # the forward also writes [d0 > 0] as one byte per 8 channels; with it the backward reads neither d0 nor the residual: bit-identical outputs
        lsums5, losses5, dpred5 = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.empty(B * V, 4, device="cuda")
        bsum5 = torch.empty_like(bsum)
        smask = torch.full((B * V, 8), 0xAA, dtype=torch.uint8, device="cuda")
        ops.mae_tail_fwd(yd.view(-1, Cd), stats, rd.view(-1, Cd), None, args[1], args[2], args[3], args[4], args[5], B, R, Cd, lsums5, losses5, None, dpred5, bwd_sums=bsum5,
                         sign_mask=smask)
        assert torch.equal(dpred5, dpred4) and torch.allclose(losses5, losses4, rtol=1e-6)   # (loss sums: fp32 partials meet in order-dependent atomics)
        bits = ((smask[:, :Cd // 8].reshape(B * V, Cd // 8, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8).view(1, 1, 8)) & 1).view(B * V, Cd).bool()
        # (the sign is that of the fp32 pre-activation; where x-hat + r cancels to the last place the two kernels' fp contraction may differ in sign:
        #  such elements have |d0| at rounding level -- none at 32^3, a handful in 1.7e8 elements)
        mism = bits != (d0k.view(B * V, Cd).float() > 0)
        assert mism.float().mean().item() < 1e-6 and (mism.sum().item() == 0 or d0k.view(B * V, Cd).float().abs()[mism].max().item() < 1e-5), \
            ("sign mask", mism.sum().item(), d0k.view(B * V, Cd).float().abs()[mism].max().item() if mism.any() else 0.0)
        dy5, dr5, dW5, db5, in_sums5 = torch.empty_like(dy), torch.empty_like(dr), torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda"), torch.empty_like(in_sums)
        ops.mae_tail_bwd(None, yd.view(-1, Cd), stats, dpred5, lsums5, args[1], in_sums5, dy5, dr5, dW5, db5, B, V, Cd, bwd_sums=bsum5, sign_mask=smask)
        # (in_sums / dW come from bsum's fp64 atomics: equal up to their summation order)
        # dr: bit-identical except where the two sign sources disagree (the mask holds the sign of the forward kernel's fp32 pre-activation, the other path
        # rebuilds d0 from y, stats and r: elements that cancel to the last place, see above)
        # -- and, at the streaming size, a bf16 tie of the last place here and there: the two instantiations of the apply pass contract the 4-term head dot
        # product differently (measured 227 of 1.7e8 elements, one bf16 ulp each)
        dmis = dr5 != dr4
        far = (dmis & ~torch.isclose(dr5.float(), dr4.float(), rtol=1e-2, atol=1e-6)).view(B * V, Cd)
        assert dmis.float().mean().item() < 1e-5 and (far.sum().item() == 0 or d0k.view(B * V, Cd).float().abs()[far].max().item() < 1e-5), \
            ("dr", dmis.sum().item(), far.sum().item(), d0k.view(B * V, Cd).float().abs()[far].max().item() if far.any() else 0.0)
        assert torch.allclose(dy5.float(), dy4.float(), rtol=1e-2, atol=1e-6 if not far.any() else 1e-3)
        assert torch.allclose(in_sums5, in_sums4, rtol=1e-6, atol=1e-9 * scale) and torch.allclose(dW5, dW4, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("R", [32, 48, 64])
def test_tail_forward_with_the_residual_formed_from_the_coarse_tensor(R):
    """decoder1's tail with r = ConvTranspose3d_{k=s=4}(x coarse) + bias formed INSIDE the pass (csrc/norm.hip: tail_fwd_coarse_kernel; unetr_block.py:62-71,
    151-158 + swin_mae3d.py:1496-1549): against autograd of the same chain in fp32 on the bf16-rounded operands (forward values, and every gradient the tail backward
    makes of the pass's d(pred), sums and sign mask), and against the pass that reads the stored residual (which sees r rounded to bf16).
    R = 48: 12^3 cells (the cell -> (z, y, x) split by multiplication, rows of 12 cells against groups of 16), a sample with partial extents."""
    from oracle import mae3d_oracle as O
    ops = _ops()
    dt = torch.bfloat16
    B, Cd, gd = 2, 48, R // 4
    V = R ** 3
    x = torch.stack([O.synthetic_grid((R, R, R), 3), O.synthetic_grid((R, R, R), 4)])
    valid = torch.ones_like(x)
    valid[1, :, R - 4:] = 0
    valid[1, :, :, :, R - 6:] = 0
    x = x * valid
    ext = torch.tensor([[R, R, R], [R - 4, R, R - 6]], dtype=torch.int32)
    y = q(rnd(B, V, Cd) * 1.3 + 0.2, dt)
    xc = q(rnd(B, gd, gd, gd, 96, seed=11), dt)
    Wt = rnd(96, 48, 4, 4, 4, seed=12, scale=96 ** -0.5)
    bt = rnd(48, seed=13, scale=0.5)
    Wo, bo = rnd(4, Cd, seed=1, scale=0.2), rnd(4, seed=2, scale=0.1)
    tm = O.draw_block_mask((gd,) * 3, 0.6, rng=__import__("random").Random(5))
    yr, Wr, br_ = (t.clone().requires_grad_(True) for t in (y, Wo, bo))
    r_ref = F.conv_transpose3d(xc.permute(0, 4, 1, 2, 3), q(Wt, dt), bt, stride=4).permute(0, 2, 3, 4, 1).reshape(B, V, Cd).clone().requires_grad_(True)
    inorm = lambda t: F.instance_norm(t.permute(0, 2, 1), eps=1e-5).permute(0, 2, 1)
    d0_ref = F.leaky_relu(inorm(yr) + r_ref, 0.01)
    d0q = q(d0_ref.detach(), dt)
    d0s = d0_ref + (d0q - d0_ref).detach()
    pred = (d0s.reshape(B, R, R, R, Cd) @ Wr.T + br_).permute(0, 4, 1, 2, 3)
    l, lr, la, *_ = O.mae_loss(x, pred, valid, tm[None, ..., None].expand(B, -1, -1, -1, 1))
    l.backward()
    # packed phase weights from the scratch of the composed-weight pack, as the model makes them
    ws = torch.empty(ops.cconv_pack_ws_floats(), dtype=torch.float32, device="cuda")
    ops.cconv_pack(dev(Wt), dev(rnd(48, 48, 3, 3, 3, seed=3)), dev(bt), torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda"), torch.empty(27, 48, device="cuda"), ws)
    Wres = torch.empty(ops.tail_residual_pack_numel(), dtype=dt, device="cuda")
    ops.tail_residual_pack(ws, Wres)
    yd = dev(y, dt)
    stats, scratch = torch.empty(B, Cd, 2, device="cuda"), torch.empty(B, Cd, 2, dtype=torch.float64, device="cuda")
    ops.instnorm_stats(yd, stats, scratch, B, V, Cd)
    Wod, bod, xd, extd, tmd = dev(Wo), dev(bo), dev(x), dev(ext), dev(tm.to(torch.uint8))
    lsums, losses, dpred = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.full((B * V, 4), 7.0, device="cuda")
    bsum = torch.empty(B * Cd * 4 + 4 * Cd, dtype=torch.float64, device="cuda")
    smask = torch.full((B * V, 8), 0xAA, dtype=torch.uint8, device="cuda")
    predk = torch.empty(B, 4, R, R, R, device="cuda")
    ops.mae_tail_fwd_from_coarse(yd.view(-1, Cd), stats, dev(xc, dt), Wres, dev(bt), Wod, bod, xd, extd, tmd, B, R, Cd, lsums, losses, dpred, bsum, smask, pred=predk)
    torch.cuda.synchronize()
    check(predk, pred.detach(), dt, "pred", 2)
    np.testing.assert_allclose(losses.cpu().numpy(), [l.item(), lr.item(), la.item()], rtol=TOL[dt])
    # the pass that reads the stored residual (bf16): same thing up to the rounding of r
    rd = dev(r_ref.detach(), dt)
    lsums2, losses2, dpred2 = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.empty(B * V, 4, device="cuda")
    bsum2, smask2 = torch.empty_like(bsum), torch.empty_like(smask)
    ops.mae_tail_fwd(yd.view(-1, Cd), stats, rd.view(-1, Cd), None, Wod, bod, xd, extd, tmd, B, R, Cd, lsums2, losses2, None, dpred2, bwd_sums=bsum2, sign_mask=smask2)
    np.testing.assert_allclose(losses.cpu().numpy(), losses2.cpu().numpy(), rtol=2e-3)
    np.testing.assert_allclose(lsums.cpu().numpy(), lsums2.cpu().numpy(), rtol=5e-3, atol=1e-3 * float(lsums2.abs().max()))
    check(dpred, dpred2.cpu(), dt, "dpred against the stored-residual pass", 2)
    sc = float(bsum2.abs().max())
    assert torch.allclose(bsum, bsum2, rtol=2e-2, atol=2e-3 * sc), ((bsum - bsum2).abs().max().item(), sc)
    bits = ((smask[:, :6].reshape(B * V, 6, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8).view(1, 1, 8)) & 1).view(B * V, Cd).bool().cpu()
    mism = bits != (d0_ref.detach().reshape(B * V, Cd) > 0)
    assert mism.float().mean().item() < 1e-5 and (mism.sum().item() == 0 or d0_ref.detach().reshape(B * V, Cd).abs()[mism].max().item() < 1e-4), \
        ("sign mask", mism.sum().item(), d0_ref.detach().reshape(B * V, Cd).abs()[mism].max().item() if mism.any() else 0.0)
    # the backward made of this pass's outputs against autograd
    dy, dr = torch.empty(B * V, Cd, dtype=dt, device="cuda"), torch.empty(B * V, Cd, dtype=dt, device="cuda")
    dW, db, in_sums = torch.zeros(4, Cd, device="cuda"), torch.zeros(4, device="cuda"), torch.empty(B, Cd, 2, dtype=torch.float64, device="cuda")
    ops.mae_tail_bwd(None, yd.view(-1, Cd), stats, dpred, lsums, Wod, in_sums, dy, dr, dW, db, B, V, Cd, bwd_sums=bsum, sign_mask=smask)
    check(dr, r_ref.grad.reshape(-1, Cd), dt, "dr", 3)
    check(dy, yr.grad.reshape(-1, Cd), dt, "dy", 3)
    check(dW, Wr.grad, dt, "dWout", 2)
    check(db, br_.grad, dt, "dbout", 2)


def test_tail_from_coarse_at_full_size_and_outside_its_domain():
    """tail_fwd_coarse_kernel at the benched volume (160^3: 40^3 cells, rows of 40 cells against groups of 16, 16 cell ranges per (dz, dy) class, two ragged samples)
    against the pass that reads the stored residual -- same losses, d(pred), backward sums and sign mask up to the bf16 rounding of r -- and the entry's refusal (-4) of a
    volume whose cell count is not a multiple of 16 (the caller then stores the residual)."""
    from nerf_mae_amd._lib import NmhError
    ops = _ops()
    dt, B, R, Cd = torch.bfloat16, 2, 160, 48
    gd, V = R // 4, R ** 3
    g = torch.Generator(device="cuda").manual_seed(7)
    y = (torch.randn(B * V, Cd, device="cuda", generator=g) * 1.3 + 0.2).to(dt)
    xc = torch.randn(B, gd, gd, gd, 96, device="cuda", generator=g).to(dt)
    Wt, bt = torch.randn(96, 48, 4, 4, 4, device="cuda", generator=g) * 96 ** -0.5, torch.randn(48, device="cuda", generator=g) * 0.5
    ws = torch.empty(ops.cconv_pack_ws_floats(), dtype=torch.float32, device="cuda")
    ops.cconv_pack(Wt, torch.randn(48, 48, 3, 3, 3, device="cuda", generator=g), bt, torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda"), torch.empty(27, 48, device="cuda"), ws)
    Wres, Wup = torch.empty(ops.tail_residual_pack_numel(), dtype=dt, device="cuda"), torch.empty(ops.upconv4_pack_numel(), dtype=dt, device="cuda")
    ops.tail_residual_pack(ws, Wres)
    ops.upconv4_pack(ws, Wup)
    r = torch.empty(B * V, Cd, dtype=dt, device="cuda")
    ops.upconv4_fwd(xc, Wup, bt, r, B, gd)
    stats = torch.empty(B, Cd, 2, device="cuda")
    ops.instnorm_stats(y, stats, torch.empty(B, Cd, 2, dtype=torch.float64, device="cuda"), B, V, Cd)
    Wo, bo = torch.randn(4, Cd, device="cuda", generator=g) * 0.2, torch.randn(4, device="cuda", generator=g) * 0.1
    x = torch.rand(B, 4, R, R, R, device="cuda", generator=g)
    ext = torch.tensor([[R, R, R], [R - 7, R - 12, R - 1]], dtype=torch.int32, device="cuda")
    tm = (torch.rand(gd ** 3, device="cuda", generator=g) < 0.75).to(torch.uint8)
    outs = []
    for coarse in (True, False):
        lsums, losses, dpred = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.full((B * V, 4), 7.0, device="cuda")
        bsum, smask = torch.empty(B * Cd * 4 + 4 * Cd, dtype=torch.float64, device="cuda"), torch.full((B * V, 8), 0xAA, dtype=torch.uint8, device="cuda")
        if coarse:
            ops.mae_tail_fwd_from_coarse(y, stats, xc, Wres, bt, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, dpred, bsum, smask)
        else:
            ops.mae_tail_fwd(y, stats, r, None, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred, bwd_sums=bsum, sign_mask=smask)
        torch.cuda.synchronize()
        outs.append((lsums, losses, dpred, bsum, smask))
    (ls1, lo1, dp1, bs1, sm1), (ls0, lo0, dp0, bs0, sm0) = outs
    np.testing.assert_allclose(lo1.cpu().numpy(), lo0.cpu().numpy(), rtol=2e-3)
    np.testing.assert_allclose(ls1.cpu().numpy(), ls0.cpu().numpy(), rtol=5e-3, atol=1e-3 * float(ls0.abs().max()))
    assert (dp1 == 7.0).sum().item() == 0                      # every voxel written
    diff = (dp1 - dp0).abs()
    assert diff.max().item() < 0.25 and diff.mean().item() < 2e-3 * dp0.abs().mean().item() + 1e-4, (diff.max().item(), diff.mean().item(), dp0.abs().mean().item())
    assert ((dp1 != 0) != (dp0 != 0)).float().mean().item() < 1e-5   # the same voxels are inside the loss regions
    assert torch.allclose(bs1, bs0, rtol=2e-2, atol=2e-3 * float(bs0.abs().max()))
    assert ((sm1[:, :6] != sm0[:, :6]).sum().item()) < 2e-3 * B * V * 6   # sign bits differ only where x-hat + r cancels to the rounding of r
    # outside the domain: 24^3 has 216 cells -- the host wrapper and the C entry both refuse
    from nerf_mae_amd._lib import lib
    R2 = 24
    a2 = (y[:R2 ** 3], stats[:1], xc.view(-1, 96)[:216].contiguous(), Wres, bt, Wo, bo, x[:1, :, :R2, :R2, :R2].contiguous(), ext[:1], tm[:216])
    o2 = (torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.empty(R2 ** 3, 4, device="cuda"),
          torch.empty(Cd * 4 + 4 * Cd, dtype=torch.float64, device="cuda"), torch.empty(R2 ** 3, 8, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        ops.mae_tail_fwd_from_coarse(*a2[:2], a2[2].view(1, 6, 6, 6, 96), *a2[3:], 1, R2, Cd, *o2)
    with pytest.raises(NmhError):
        lib().call("nmh_mae_tail_fwd_from_coarse", ops.dt_of(y), *a2, 1, R2, Cd, o2[0], o2[1], None, o2[2], 0.01, o2[3], o2[4], ops._st())


@pytest.mark.parametrize("dt", DTS)
def test_instnorm_bwd_without_out(dt):
    """rmode 0: sign(out) == sign(x - mean), so `out` may be omitted."""
    ops = _ops()
    B, V, C = 2, 1500, 48
    x, dout = q(rnd(B, V, C) * 1.5 + 0.3, dt), q(rnd(B, V, C, seed=2), dt)
    xd, dd = dev(x, dt), dev(dout, dt)
    stats, scratch = torch.empty(B, C, 2, device="cuda"), torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
    ops.instnorm_stats(xd, stats, scratch, B, V, C)
    out = torch.empty(B, V, C, dtype=dt, device="cuda")
    ops.instnorm_apply(xd, stats, out, B, V, C)
    res = []
    for o in (out, None):
        sums = torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
        ops.instnorm_bwd_reduce(dd, o, xd, stats, sums, B, V, C)
        dx = torch.empty(B, V, C, dtype=dt, device="cuda")
        ops.instnorm_bwd_apply(dd, o, xd, stats, sums, dx, B, V, C)
        res.append(dx.float().cpu())
    check(res[1], res[0], dt, "dx without out", 1)


@pytest.mark.parametrize("B,V", [(1, 4096), (3, 1500), (8, 64 ** 3), (2, 41)])
def test_instnorm_bwd_apply_background_launch_is_bit_identical(B, V):
    """nmh_instnorm_bwd_apply_bg (one persistent workgroup per CU, the footprint that runs beside the 48 -> 48 weight-gradient kernel in decoder-1's backward,
    unetr_block.py:57-63 backward) == nmh_instnorm_bwd_apply(rmode 0, out = None), bit for bit; ragged voxel counts and fewer / more samples than workgroups."""
    ops = _ops()
    dt, C = torch.bfloat16, 48
    x, dout = q(rnd(B, V, C) * 1.5 + 0.3, dt), q(rnd(B, V, C, seed=2), dt)
    xd, dd = dev(x, dt), dev(dout, dt)
    stats, scratch = torch.empty(B, C, 2, device="cuda"), torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
    ops.instnorm_stats(xd, stats, scratch, B, V, C)
    sums = torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
    ops.instnorm_bwd_reduce(dd, None, xd, stats, sums, B, V, C)
    ref = torch.empty(B, V, C, dtype=dt, device="cuda")
    ops.instnorm_bwd_apply(dd, None, xd, stats, sums, ref, B, V, C)
    got = torch.full((B, V, C), float("nan"), dtype=dt, device="cuda")
    ops.instnorm_bwd_apply_bg(dd, xd, stats, sums, got, B, V, C)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("dt", DTS)
def test_patch_embed_gather_and_bias_grad(dt):
    ops = _ops()
    B, R, C = 2, 16, 96
    x = rnd(B, 4, R, R, R)
    w, b = q(rnd(C, 4, 4, 4, 4, seed=1, scale=1 / 16), dt), rnd(C, seed=2)
    A = torch.empty(B * (R // 4) ** 3, 256, dtype=dt, device="cuda")
    ops.patch_embed_gather(dev(x), A, B, R)
    y = ops.gemm_nt(A, dev(w.reshape(C, 256), dt), bias=dev(b))
    ref = F.conv3d(q(x, dt), w, b, stride=4).permute(0, 2, 3, 4, 1).reshape(-1, C)
    check(y, ref, dt, "patch embed")
    dY = q(rnd(1234, 768, seed=3), dt)
    db = torch.zeros(768, device="cuda")
    ops.bias_grad(dev(dY, dt), db, 1234, 768)
    check(db, dY.sum(0), dt, "bias grad")
    dY2 = q(rnd(333, 3072, seed=4), dt)
    db2 = torch.zeros(3072, device="cuda")
    ops.bias_grad(dev(dY2, dt), db2, 333, 3072)
    check(db2, dY2.sum(0), dt, "bias grad wide")


def test_adamw_and_clip_match_torch():
    ops = _ops()
    n = 10007
    p0, g0 = rnd(n), rnd(n, seed=1) * 3
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999))
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    coef, nrm = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    for step in range(1, 4):
        g = g0 * step
        pr.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 0.1)
        opt.step()
        gd = dev(g)
        ops.grad_sqnorm(gd, acc)
        ops.clip_coef(acc, 0.1, coef, nrm)
        hyper = dev(torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1e-2, 1 - 0.9 ** step, 1 - 0.999 ** step]))
        ops.adamw_step(p, gd, m, v, hyper, coef)
        assert abs(nrm.item() - tn.item()) / tn.item() < 1e-5
    assert relerr(p, pr) < 1e-5


def _pack_via_kernel(w, mode, dt, n_out):
    """run the single-launch pack kernel for one tensor (descriptor table of 1)"""
    import struct
    ops = _ops()
    src = w.float().cuda().contiguous()
    dst = torch.empty(n_out, dtype=dt, device="cuda")
    sh = list(w.shape)
    d = (sh[0], sh[1], int(np.prod(sh[2:])) if len(sh) > 2 else 0)
    descs = torch.frombuffer(bytearray(struct.pack("<QQiiiiq", src.data_ptr(), dst.data_ptr(), mode, d[0], d[1], d[2], n_out)), dtype=torch.uint8).cuda()
    if mode == 1:   # 2-D transpose: one block per 32x32 tile of the d0 x d1 source, blkstart = tile id
        d1 = int(np.prod(sh[1:]))
        d = (sh[0], d1, 0)
        descs = torch.frombuffer(bytearray(struct.pack("<QQiiiiq", src.data_ptr(), dst.data_ptr(), mode, d[0], d[1], d[2], n_out)), dtype=torch.uint8).cuda()
        nb = ((d[0] + 31) // 32) * ((d1 + 31) // 32)
        bst = torch.arange(nb, dtype=torch.int64).cuda()
    elif mode in (2, 3):   # conv packs: blkstart = block id over (co, 32 ci) / (ci, 32 co)
        nb = sh[0] * ((sh[1] + 31) // 32) if mode == 2 else sh[1] * ((sh[0] + 31) // 32)
        bst = torch.arange(nb, dtype=torch.int64).cuda()
    else:
        nb = (n_out + 1023) // 1024
        bst = (torch.arange(nb, dtype=torch.int64) * 1024).cuda()
    b2d = torch.zeros(nb, dtype=torch.int32, device="cuda")
    ops.pack_weights(1 if dt == torch.bfloat16 else 0, descs, b2d, bst, nb)
    torch.cuda.synchronize()
    return dst


@pytest.mark.parametrize("dt", DTS)
def test_pack_weight_modes(dt):
    w2 = rnd(24, 40)
    check(_pack_via_kernel(w2, 1, dt, w2.numel()).view(40, 24), q(w2, dt).T, dt, "transpose")
    w3 = rnd(70, 33)      # ragged 32x32 tiles on both axes
    check(_pack_via_kernel(w3, 1, dt, w3.numel()).view(33, 70), q(w3, dt).T, dt, "transpose ragged")
    wc = rnd(16, 24, 3, 3, 3)
    check(_pack_via_kernel(wc, 2, dt, wc.numel()).view(16, 27, 24), q(wc, dt).reshape(16, 24, 27).permute(0, 2, 1), dt, "conv fwd pack")
    check(_pack_via_kernel(wc, 3, dt, wc.numel()).view(24, 27, 16), q(wc, dt).reshape(16, 24, 27).flip(2).permute(1, 2, 0), dt, "conv dgrad pack")
    wr = rnd(40, 72, 3, 3, 3)   # ragged 32-channel blocks on both axes
    check(_pack_via_kernel(wr, 2, dt, wr.numel()).view(40, 27, 72), q(wr, dt).reshape(40, 72, 27).permute(0, 2, 1), dt, "conv fwd pack ragged")
    check(_pack_via_kernel(wr, 3, dt, wr.numel()).view(72, 27, 40), q(wr, dt).reshape(40, 72, 27).flip(2).permute(1, 2, 0), dt, "conv dgrad pack ragged")
    wt = rnd(24, 16, 2, 2, 2)
    check(_pack_via_kernel(wt, 4, dt, wt.numel()).view(8 * 16, 24), q(wt, dt).reshape(24, 16, 8).permute(2, 1, 0).reshape(128, 24), dt, "convT fwd pack")
    check(_pack_via_kernel(wt, 5, dt, wt.numel()).view(24, 8 * 16), q(wt, dt).reshape(24, 16, 8).permute(0, 2, 1).reshape(24, 128), dt, "convT dgrad pack")


@pytest.mark.parametrize("B,D,H,W", [(1, 8, 16, 32), (2, 4, 8, 16), (1, 12, 24, 48), (1, 6, 10, 20), (1, 40, 40, 40)])
def test_conv48_specialised_matches_reference_conv(B, D, H, W):
    """LDS-halo Cin=Cout=48 bf16 kernel (forward pack and dgrad pack, accumulate) vs F.conv3d; incl. ragged tiles"""
    ops = _ops()
    dt = torch.bfloat16
    x = q(rnd(B, 48, D, H, W), dt)
    w = q(rnd(48, 48, 3, 3, 3, seed=1, scale=(27 * 48) ** -0.5), dt)
    dy = q(rnd(B, 48, D, H, W, seed=2), dt)
    xr = x.clone().requires_grad_(True)
    y = F.conv3d(xr, w, padding=1)
    y.backward(dy)
    n = 41 * 3 * 64 * 8
    wk_f, wk_d = _pack_via_kernel(w, 6, dt, n), _pack_via_kernel(w, 7, dt, n)
    xcl = dev(x.permute(0, 2, 3, 4, 1), dt)
    acc = torch.empty(B, 48, 2, dtype=torch.float64, device="cuda")
    yk = ops.conv3d_k3_c48(xcl, wk_f, stats_acc=acc)
    check(yk.permute(0, 4, 1, 2, 3), y, dt, "conv48 fwd")
    # fused InstanceNorm statistics == statistics of the stored (bf16) output
    st = torch.empty(B, 48, 2, device="cuda")
    ops.instnorm_finalize(acc, st, B, D * H * W, 48)
    yf = yk.float().reshape(B, -1, 48)
    check(st[..., 0], yf.mean(1), torch.float32, "fused IN mean", 5)
    check(st[..., 1], (yf.var(1, unbiased=False) + 1e-5).rsqrt(), torch.float32, "fused IN rstd", 5)
    base = q(rnd(B, D, H, W, 48, seed=3), dt)
    out = dev(base, dt)
    dycl = dev(dy.permute(0, 2, 3, 4, 1), dt)
    ops.conv3d_k3_c48(dycl, wk_d, out=out, accumulate=True)
    check(out.permute(0, 4, 1, 2, 3), xr.grad + base.permute(0, 4, 1, 2, 3), dt, "conv48 dgrad+accumulate")
    wr = w.clone().requires_grad_(True)
    F.conv3d(x, wr, padding=1).backward(dy)
    dW = torch.full((48, 48, 3, 3, 3), 0.5, device="cuda")
    ops.conv3d_k3_c48_wgrad(dycl, xcl, dW)
    check(dW, wr.grad + 0.5, dt, "conv48 wgrad")


@pytest.mark.parametrize("B,D,H,W,Cin,Cout", [(1, 8, 16, 32, 96, 96), (2, 6, 10, 20, 192, 96), (1, 12, 9, 40, 96, 192), (1, 4, 8, 16, 144, 48),
                                                (1, 32, 64, 66, 96, 96)])
def test_conv48_multiblock_matches_reference_conv(B, D, H, W, Cin, Cout):
    """the LDS-halo kernel on 48-channel blocks (Cin, Cout multiples of 48: (tile, output block, input block) work items, one weight image
    per block pair): forward pack, dgrad pack with accumulate, ragged tiles -- vs F.conv3d.  Volumes of fewer than 256 tiles spread
    (tile, output block) units over the workgroups; the last case has more tiles and walks whole tiles"""
    ops = _ops()
    dt = torch.bfloat16
    x = q(rnd(B, Cin, D, H, W), dt)
    w = q(rnd(Cout, Cin, 3, 3, 3, seed=1, scale=(27 * Cin) ** -0.5), dt)
    dy = q(rnd(B, Cout, D, H, W, seed=2), dt)
    xr = x.clone().requires_grad_(True)
    y = F.conv3d(xr, w, padding=1)
    y.backward(dy)
    n = 41 * 3 * 64 * 8 * (Cin // 48) * (Cout // 48)
    wk_f, wk_d = _pack_via_kernel(w, 6, dt, n), _pack_via_kernel(w, 7, dt, n)
    xcl = dev(x.permute(0, 2, 3, 4, 1), dt)
    yk = ops.conv3d_k3_c48mb(xcl, wk_f, Cout)
    check(yk.permute(0, 4, 1, 2, 3), y, dt, "conv48mb fwd")
    base = q(rnd(B, D, H, W, Cin, seed=3), dt)
    out = dev(base, dt)
    dycl = dev(dy.permute(0, 2, 3, 4, 1), dt)
    ops.conv3d_k3_c48mb(dycl, wk_d, Cin, out=out, accumulate=True)
    check(out.permute(0, 4, 1, 2, 3), xr.grad + base.permute(0, 4, 1, 2, 3), dt, "conv48mb dgrad+accumulate", 2)
    out2 = torch.empty_like(out)
    ops.conv3d_k3_c48mb(dycl, wk_d, Cin, out=out2)
    check(out2.permute(0, 4, 1, 2, 3), xr.grad, dt, "conv48mb dgrad", 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_copy_cols_strided(dtype):
    """skip connection into / out of the concatenated decoder tensor (torch.cat(dim=1) in channels-last = column block copy)"""
    from nerf_mae_amd import ops
    M, C = 1000, 96
    skip = torch.randn(M, C, device="cuda").to(dtype)
    cat = torch.zeros(M, 2 * C, device="cuda", dtype=dtype)
    ops.copy_cols(skip, cat[:, C:])
    assert torch.equal(cat[:, C:], skip) and (cat[:, :C] == 0).all()
    back = torch.empty_like(skip)
    ops.copy_cols(cat[:, C:], back)
    assert torch.equal(back, skip)
    with pytest.raises(RuntimeError):
        ops.copy_cols(skip, cat[:, C:C + 8])


@pytest.mark.parametrize("variant", ["dma", "reg", "big", "big_reg", "dma_fg"])
@pytest.mark.parametrize("case", ["flat_many_tiles", "split_few_tiles", "mixed_ragged", "all192"])
def test_gemm_tn_grouped(case, variant, monkeypatch):
    """nmh_gemm_tn_grouped (bf16): several weight-gradient problems per launch -- flat (a workgroup walks every sample), split at
    sample-aligned ranges (+ grouped reduce), stochastic-depth row scales applied per sample, fused bias gradients, ragged N / K tiles,
    rows per sample that are not a multiple of the 64-row chunk -- against fp32 matmuls on the bf16-rounded operands; accumulates
    into dW (+=)."""
    ops = _ops()
    dt = torch.bfloat16
    # kernel variants (read per call): register-staged chunk transport, 192x192 tiles (taken when every N, K of the launch is a multiple of 192)
    monkeypatch.setenv("NMH_TNG_REG", "1" if "reg" in variant else "0")
    monkeypatch.setenv("NMH_TNG_BIG", "1" if "big" in variant else "0")
    if case == "flat_many_tiles":      # >= 384 tiles in the launch: nobody splits
        specs = [(4, 500, 384, 1536, True, True), (4, 500, 1536, 384, False, True), (4, 512, 384, 384, False, True), (4, 512, 1152, 384, False, False)]
    elif case == "split_few_tiles":    # stage-0-like: 12 tiles -> sample-aligned splits + reduce launch
        specs = [(2, 3000, 96, 384, True, True), (2, 3000, 384, 96, False, True), (2, 3000, 96, 96, False, True), (2, 3000, 288, 96, False, False)]
    elif case == "all192":             # every width a multiple of 192 (the large-tile variant's domain): flat and, with few tiles, split
        specs = [(3, 300, 384, 1536, True, True), (3, 300, 1536, 384, False, True), (3, 320, 384, 384, False, True), (3, 320, 1152, 384, False, False)]
        if variant == "big":
            specs = specs[2:3] + [(2, 2100, 192, 384, True, True)]
    else:                              # 45 problems (two launches), widths that are not multiples of 96, one sample, tiny M
        specs = [(1, 70, 128, 512, True, True), (3, 130, 512, 128, False, True), (2, 64, 64, 256, False, False)] + [(2, 200, 192, 96, i % 2 == 0, True) for i in range(42)]
    q_ = ops.WgradQueue()
    refs, outs = [], []
    for i, (nsamp, rps, N, K, scaled, bias) in enumerate(specs):
        M = nsamp * rps
        A, Bm = q(rnd(M, N, seed=3 * i), dt), q(rnd(M, K, seed=3 * i + 1), dt)
        rs = (torch.rand(nsamp, generator=torch.Generator().manual_seed(i)) > 0.3).float() / 0.7 if scaled else None
        dW0 = rnd(N, K, seed=3 * i + 2)
        db0 = rnd(N, seed=i + 50)
        As = A if rs is None else A * rs.repeat_interleave(rps)[:, None]
        refs.append((dW0 + As.T @ Bm, db0 + As.sum(0)))
        dW, db = dev(dW0), dev(db0) if bias else None
        outs.append((dW, db))
        q_.pending.append((dev(A, dt), dev(Bm, dt), dW, db, None if rs is None else dev(rs), rps))
    q_.flush(foreground="fg" in variant)   # fg: nmh_gemm_tn_grouped_fg (the split of the last flushes of a backward pass: 640 workgroups)
    q_.join()
    torch.cuda.synchronize()
    for i, ((dW, db), (rW, rb)) in enumerate(zip(outs, refs)):
        assert relerr(dW, rW) < 2e-3, (case, i, specs[i], relerr(dW, rW))    # bf16 operands, fp32 accumulation: only summation order differs
        if db is not None:
            assert relerr(db, rb) < 2e-3, (case, i, "bias", relerr(db, rb))


@pytest.mark.parametrize("case", ["stage0", "stage1", "ragged_widths", "tile_fallback_mix", "stage2_rows"])
def test_gemm_tn_grouped_streaming(case, monkeypatch):
    """The streaming weight-gradient kernel (csrc/tn_grouped.hip: gemm_tn_stream_kernel; contractions of >= 32768 rows with outputs up to 768 x 768: a
    workgroup owns a row range inside one sample and a whole output block) behind nmh_gemm_tn_grouped: every block shape on offer, several blocks along N and
    along K, widths that are not multiples of 96, rows per sample that are not a multiple of the 32-row chunk, stochastic-depth row scales, fused bias
    gradients, accumulation into dW -- against fp32 matmuls on the bf16-rounded operands, and against the tile kernels (NMH_TNS=0) on the same inputs."""
    ops = _ops()
    dt = torch.bfloat16
    if case == "stage0":       # qkv / proj / fc1 / fc2 of a 96-channel block: blocks {384,96} {192,96} {96,384}
        specs = [(4, 8200, 288, 96, False, True), (4, 8200, 96, 96, True, True), (4, 8200, 384, 96, False, True), (4, 8200, 96, 384, True, True)]
    elif case == "stage1":     # 192 channels + the patch-merging reduction: {384,192} x 2 blocks, {192,192}, {192,384} x 2 blocks along K
        specs = [(2, 16400, 576, 192, False, True), (2, 16400, 192, 192, True, True), (2, 16400, 768, 192, False, True), (2, 16400, 192, 768, True, True),
                 (2, 16400, 192, 768, False, False)]
    elif case == "ragged_widths":
        specs = [(3, 11000, 200, 104, True, True), (3, 11000, 40, 640, False, True), (1, 33000, 104, 200, True, False), (5, 7000, 768, 768, False, True)]
    elif case == "stage2_rows":  # stage-2 shapes (8000 rows): below the streaming threshold in the product, forced through the streaming blocks here (NMH_TNS_RATIO=0)
        specs = [(8, 1000, 1152, 384, False, True), (8, 1000, 1536, 384, False, True), (8, 1000, 384, 1536, False, False), (8, 1000, 384, 1536, True, True),
                 (4, 1000, 576, 192, False, True), (2, 1030, 384, 384, False, True)]
    else:                      # one launch mixes streaming problems with problems the tile kernels keep (few rows; an output beyond 768 x 768)
        specs = [(4, 8200, 384, 96, False, True), (4, 500, 384, 1536, True, True), (4, 8200, 96, 384, True, True), (2, 20000, 1152, 768, False, True)]
    results = {}
    monkeypatch.setenv("NMH_TNS_RATIO", "0")   # (the dispatch takes the streaming kernel only where the operands dwarf the partial blocks: forced here at test sizes)
    for mode in ("1", "0"):
        monkeypatch.setenv("NMH_TNS", mode)
        q_ = ops.WgradQueue()
        refs, outs = [], []
        for i, (nsamp, rps, N, K, scaled, bias) in enumerate(specs):
            M = nsamp * rps
            A, Bm = q(rnd(M, N, seed=3 * i), dt), q(rnd(M, K, seed=3 * i + 1), dt)
            rs = (torch.rand(nsamp, generator=torch.Generator().manual_seed(i)) > 0.3).float() / 0.7 if scaled else None
            dW0 = rnd(N, K, seed=3 * i + 2)
            db0 = rnd(N, seed=i + 50)
            As = A if rs is None else A * rs.repeat_interleave(rps)[:, None]
            refs.append((dW0 + As.T @ Bm, db0 + As.sum(0)))
            dW, db = dev(dW0), dev(db0) if bias else None
            outs.append((dW, db))
            q_.pending.append((dev(A, dt), dev(Bm, dt), dW, db, None if rs is None else dev(rs), rps))
        q_.flush()
        q_.join()
        torch.cuda.synchronize()
        for i, ((dW, db), (rW, rb)) in enumerate(zip(outs, refs)):
            assert relerr(dW, rW) < 2e-3, (case, mode, i, specs[i], relerr(dW, rW))    # bf16 operands, fp32 accumulation: only summation order differs
            assert rel_l2(dW, rW) < 2e-3, (case, mode, i, specs[i], rel_l2(dW, rW))
            if db is not None:
                assert relerr(db, rb) < 2e-3, (case, mode, i, "bias", relerr(db, rb))
        results[mode] = outs
    for (w1, _), (w0, _) in zip(results["1"], results["0"]):
        assert relerr(w1, w0) < 1e-3


@pytest.mark.parametrize("B,v,k,Cin,Cout,skip", [(2, 32, 2, 96, 48, False), (1, 32, 2, 48, 24, True), (1, 20, 4, 96, 48, False)])
def test_upconv_wgrad_grouped_streaming(B, v, k, Cin, Cout, skip, monkeypatch):
    """The transpose-conv weight / bias gradient (k^2 folded problems over the pixel-shuffled view of the fine gradient, contiguous runs without a skip half and
    k strided pieces with one) through the streaming kernel (>= 32768 coarse voxels per sample... the decoder-1 shape class) against autograd of
    F.conv_transpose3d on the bf16-rounded operands and against the tile kernels."""
    ops = _ops()
    dt = torch.bfloat16
    x = q(rnd(B, Cin, v, v, v), dt)
    w = q(rnd(Cin, Cout, k, k, k, seed=1, scale=Cin ** -0.5), dt)
    b = rnd(Cout, seed=2, scale=0.1)
    V = v * k
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv_transpose3d(xr, wr, br, stride=k)
    dy = q(rnd(*y.shape, seed=4), dt)
    y.backward(dy)
    Cc = 2 * Cout if skip else Cout
    dc = torch.zeros(B * V ** 3, Cc, dtype=dt, device="cuda")
    dc[:, :Cout] = dev(dy.permute(0, 2, 3, 4, 1).reshape(-1, Cout), dt)
    if skip:
        dc[:, Cout:] = 7.0    # the skip half's gradient must not leak into the transpose conv's
    xcl = dev(x.permute(0, 2, 3, 4, 1).reshape(-1, Cin), dt)
    outs = {}
    monkeypatch.setenv("NMH_TNS_RATIO", "0")
    for mode in ("1", "0"):
        monkeypatch.setenv("NMH_TNS", mode)
        dW, db = torch.zeros(Cin, Cout, k, k, k, device="cuda"), torch.zeros(Cout, device="cuda")
        ops.upconv_wgrad_grouped(dc, xcl, dW, db, B, v, k, Cin, Cout)
        torch.cuda.synchronize()
        check(dW, wr.grad, dt, f"grouped upconv dW NMH_TNS={mode}", 2)
        check(db, br.grad, dt, f"grouped upconv dbias NMH_TNS={mode}", 2)
        outs[mode] = dW
    assert relerr(outs["1"], outs["0"]) < 1e-3


def test_grad_bucket_casts():
    """nmh_grad_to_bf16 / nmh_grad_from_bf16 (bf16 gradient buckets of the data-parallel exchange): RNE cast and scaled widening, bit-exact"""
    ops = _ops()
    g = rnd(8 * 1237, seed=4, scale=3.0).cuda()
    b = torch.empty_like(g, dtype=torch.bfloat16)
    ops.grad_to_bf16(g, b)
    assert torch.equal(b, g.to(torch.bfloat16))
    out = torch.empty_like(g)
    ops.grad_from_bf16(b, out, 0.5)
    assert torch.equal(out, b.float() * 0.5)


def test_step_params_kernel_expands_block_mask_and_stores_hyper_and_extents():
    """nmh_step_params: block bits / hyper-parameters / extents as kernel arguments -> token mask (== draw_block_mask), hyper, extents"""
    import random
    from nerf_mae_amd.model import draw_block_bits, draw_block_mask
    ops = _ops()
    for g in (8, 10, 40):
        want = draw_block_mask((g, g, g), 0.6, rng=random.Random(g))
        bits = draw_block_bits((g, g, g), 0.6, rng=random.Random(g))
        tok = torch.full((g ** 3,), 7, dtype=torch.uint8, device="cuda")
        hy = torch.zeros(8, device="cuda")
        ex = torch.zeros((3, 3), dtype=torch.int32, device="cuda")
        ops.step_params(tokmask=tok, block_bits=bits, nb=bits.shape[0], g=g, hyper=[1e-4, 0.9, 0.999, 1e-8, 1e-3, 0.1, 0.001, 1.0], hyper_dev=hy,
                        extents=[[g, g - 1, 3], [1, 2, 3], [4, 5, 6]], extents_dev=ex)
        assert torch.equal(tok.cpu().view(g, g, g), want)
        assert torch.allclose(hy.cpu(), torch.tensor([1e-4, 0.9, 0.999, 1e-8, 1e-3, 0.1, 0.001, 1.0]))
        assert ex.cpu().tolist() == [[g, g - 1, 3], [1, 2, 3], [4, 5, 6]]
    hy2 = torch.zeros(8, device="cuda")
    ops.step_params(hyper=[float(i) for i in range(8)], hyper_dev=hy2)     # hyper only (FusedAdamW.update_hyper)
    assert hy2.cpu().tolist() == [float(i) for i in range(8)]


@pytest.mark.parametrize("B,D,H,W,Cin,Cout", [(1, 8, 16, 32, 64, 64), (2, 4, 8, 16, 64, 64), (1, 6, 10, 20, 64, 64), (1, 12, 9, 33, 128, 64),
                                              (2, 5, 7, 18, 64, 128), (1, 20, 20, 20, 192, 128), (1, 40, 40, 40, 64, 64)])
def test_conv64_block_kernel_matches_reference_conv(B, D, H, W, Cin, Cout):
    """LDS-halo 64-channel-block bf16 kernel (swin_b decoder1 / FPN neck; forward pack, dgrad pack + accumulate, fused InstanceNorm
    statistics, several input / output channel blocks, ragged tiles on every axis) vs F.conv3d"""
    ops = _ops()
    dt = torch.bfloat16
    x = q(rnd(B, Cin, D, H, W), dt)
    w = q(rnd(Cout, Cin, 3, 3, 3, seed=1, scale=(27 * Cin) ** -0.5), dt)
    dy = q(rnd(B, Cout, D, H, W, seed=2), dt)
    xr = x.clone().requires_grad_(True)
    y = F.conv3d(xr, w, padding=1)
    y.backward(dy)
    n = ops.conv64_pack_numel(Cin, Cout)
    wk_f, wk_d = _pack_via_kernel(w, 8, dt, n), _pack_via_kernel(w, 9, dt, n)
    xcl = dev(x.permute(0, 2, 3, 4, 1), dt)
    acc = torch.empty(B, Cout, 2, dtype=torch.float64, device="cuda")
    yk = ops.conv3d_k3_c64(xcl, wk_f, Cout, stats_acc=acc)
    check(yk.permute(0, 4, 1, 2, 3), y, dt, "conv64 fwd")
    st = torch.empty(B, Cout, 2, device="cuda")
    ops.instnorm_finalize(acc, st, B, D * H * W, Cout)
    yf = yk.float().reshape(B, -1, Cout)
    check(st[..., 0], yf.mean(1), torch.float32, "fused IN mean", 5)
    check(st[..., 1], (yf.var(1, unbiased=False) + 1e-5).rsqrt(), torch.float32, "fused IN rstd", 5)
    base = q(rnd(B, D, H, W, Cin, seed=3), dt)
    out = dev(base, dt)
    dycl = dev(dy.permute(0, 2, 3, 4, 1), dt)
    ops.conv3d_k3_c64(dycl, wk_d, Cin, out=out, accumulate=True)
    check(out.permute(0, 4, 1, 2, 3), xr.grad + base.permute(0, 4, 1, 2, 3), dt, "conv64 dgrad+accumulate")
    wr = w.clone().requires_grad_(True)
    F.conv3d(x, wr, padding=1).backward(dy)
    dW = torch.full((Cout, Cin, 3, 3, 3), 0.5, device="cuda")
    ws = torch.empty(ops.lib().call("nmh_conv3d_k3_c64_wgrad_ws_floats"), device="cuda")
    ops.lib().call("nmh_conv3d_k3_c64_wgrad", dycl, xcl, dW, ws, B, D, H, W, Cin, Cout, torch.cuda.current_stream().cuda_stream)
    check(dW, wr.grad + 0.5, dt, "conv64 wgrad")


def test_prezeroed_accumulator_arena():
    """nmh_set_prezeroed_arena / ops.AccArena: accumulators inside the registered range are NOT cleared by the entry points (the caller
    clears the arena once per step), accumulators outside it still are; a slice is handed out once per begin()"""
    ops = _ops()
    dt = torch.bfloat16
    B, V, C = 2, 1000, 48
    x = q(rnd(B * V, C), dt)
    xd = dev(x, dt)
    ref = torch.empty(B, C, 2, device="cuda")
    ops.instnorm_stats(xd, ref, torch.full((B, C, 2), 7.0, dtype=torch.float64, device="cuda"), B, V, C)   # outside the arena: cleared by the library
    mu = x.view(B, V, C).float().mean(1)
    assert torch.allclose(ref[..., 0].cpu(), mu, atol=2e-3)
    ar = ops.AccArena.get(xd.device)
    assert ar is not None
    ar.begin()
    a1, a2 = ops.acc_zeros((B, C, 2), xd.device), ops.acc_zeros((B, C, 2), xd.device)
    assert a1.data_ptr() != a2.data_ptr() and float(a1.abs().sum()) == 0.0 and float(a2.abs().sum()) == 0.0
    st = torch.empty(B, C, 2, device="cuda")
    ops.instnorm_stats(xd, st, a1, B, V, C)
    assert torch.allclose(st, ref, rtol=1e-6, atol=1e-7)
    a2.fill_(3.0)                         # a dirty slice inside the arena is the caller's bug: the library does not clear it ...
    st2 = torch.empty(B, C, 2, device="cuda")
    ops.instnorm_stats(xd, st2, a2, B, V, C)
    assert not torch.allclose(st2, ref)
    ar.begin()                            # ... the next begin() does
    a3 = ops.acc_zeros((B, C, 2), xd.device)
    assert a3.data_ptr() == a1.data_ptr() and float(ar.buf.abs().sum()) == 0.0
    ops.instnorm_stats(xd, st2, a3, B, V, C)
    assert torch.allclose(st2, ref, rtol=1e-6, atol=1e-7)
    big = ops.acc_zeros((ar.buf.numel() + 16,), xd.device)   # does not fit: an ordinary tensor outside the range
    assert not (ar.buf.data_ptr() <= big.data_ptr() < ar.buf.data_ptr() + ar.BYTES)


# ------------------------------------------------------------------------------------------------
# fused MLP branch (csrc/mlp_fused.hip): LN2 -> fc1 -> GELU -> fc2 -> stochastic-depth row scale -> + residual, forward and backward,
# against plain fp32 math on the bf16-rounded inputs (the torchvision MLP + LayerNorm of swin_mae3d.py:351-358,368) and against the
# unfused HIP chain it replaces; every rows-per-wave variant, ragged row counts (partial 64-row tiles), per-sample scales, and the
# window-ordered second output of the backward (shifted + padded geometry)
# ------------------------------------------------------------------------------------------------
def _mlp_ref(x, gam, bet, W1, b1, W2, b2, rs, tps):
    xn = F.layer_norm(x, (x.shape[1],), gam, bet, 1e-5)
    hp = xn @ W1.T + b1
    h = F.gelu(hp)
    y = h @ W2.T + b2
    scale = rs.repeat_interleave(tps)[: x.shape[0], None]
    return x + scale * y, xn, h


@pytest.mark.parametrize("C,M,tps", [(96, 200, 100), (96, 4133, 4133), (96, 70000, 35000), (128, 1000, 500), (192, 777, 259), (192, 2048, 1024), (256, 300, 150), (384, 130, 65), (384, 1000, 500)])
@pytest.mark.parametrize("mt", [1, 2, 4, "p1", "p2"])
def test_mlp_fused_forward_backward(C, M, tps, mt, monkeypatch):
    """mt = 1/2/4: the chunk-ring kernels with 16*mt rows per wave; "p1"/"p2": the persistent LDS-resident-weight kernels (C = 96)"""
    ops = _ops()
    dt = torch.bfloat16
    if mt in (4, "p1", "p2") and C != 96:
        pytest.skip("variant exists for C = 96 only")
    if isinstance(mt, str):
        monkeypatch.delenv("NMH_MLP_FWD_MT", raising=False)
        monkeypatch.delenv("NMH_MLP_BWD_MT", raising=False)
        monkeypatch.setenv("NMH_MLP96_FWD_MT", mt[1])
        monkeypatch.setenv("NMH_MLP96_BWD_MT", mt[1])
    else:
        monkeypatch.setenv("NMH_MLP_FWD_MT", str(mt))
        monkeypatch.setenv("NMH_MLP_BWD_MT", str(min(mt, 2)))
    B = (M + tps - 1) // tps
    x = q(rnd(M, C, seed=1) * 1.5 + 0.3, dt)
    gam, bet = 1 + 0.2 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    W1, b1 = q(rnd(4 * C, C, seed=4, scale=C ** -0.5), dt), 0.2 * rnd(4 * C, seed=5)
    W2, b2 = q(rnd(C, 4 * C, seed=6, scale=(4 * C) ** -0.5), dt), 0.2 * rnd(C, seed=7)
    rs = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.8, 1.0][:B] + [1.0] * max(0, B - 4))
    dy = q(rnd(M, C, seed=8), dt)
    xr, gr, br, W1r, b1r, W2r, b2r = [t.clone().requires_grad_(True) for t in (x, gam, bet, W1, b1, W2, b2)]
    ref, xn_ref, h_ref = _mlp_ref(xr, gr, br, W1r, b1r, W2r, b2r, rs, tps)
    (ref * dy).sum().backward()
    # ---- forward
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    W1d, W2Td = dev(W1, dt), dev(W2.T.contiguous(), dt)
    out = ops.mlp_fused_fwd(dev(x, dt), dev(gam), dev(bet), W1d, dev(b1), W2Td, dev(b2), rowscale=dev(rs), rows_per_scale=tps, mean=mean, rstd=rstd)
    check(out, ref, dt, "mlp fused fwd")
    mu = x.mean(1)
    check(mean, mu, torch.float32, "mlp fused mean", 10)
    check(rstd, (x.var(1, unbiased=False) + 1e-5).rsqrt(), torch.float32, "mlp fused rstd", 10)
    # the unfused chain on the same inputs: LN -> gemm(+GELU) -> gemm(+scale, +residual)
    xn_u = torch.empty(M, C, dtype=dt, device="cuda")
    m_u, r_u = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.layernorm_fwd(dev(x, dt), dev(gam), dev(bet), xn_u, m_u, r_u, M, C)
    hpre_u = torch.empty(M, 4 * C, dtype=dt, device="cuda")
    hact_u = ops.gemm_nt(xn_u, W1d, bias=dev(b1), act=1, C2=hpre_u)
    out_u = ops.gemm_nt(hact_u, dev(W2, dt), bias=dev(b2), resid=dev(x, dt), rowscale=dev(rs), rows_per_scale=tps)
    assert_close(out, out_u.float().cpu(), 1e-2, "mlp fused fwd vs unfused chain")   # two bf16 roundings of the same fp32 arithmetic
    # ---- backward (with the window-ordered second output when the rows form a token grid)
    geom = dyw = None
    dyw_scale = torch.tensor([0.5, 2.0, 1.0, 1.0][:B] + [1.0] * max(0, B - 4))
    if M == B * tps and tps in (100, 500, 150, 65):
        shape = {100: (5, 5, 4), 500: (10, 10, 5), 150: (6, 5, 5), 65: (13, 5, 1)}[tps]
        geom = ops.WinGeom(B, *shape, [2, 2, 2])
        dyw = torch.full((geom.rows, C), 3.0, dtype=dt, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx1, x1n, hact, dh = ops.mlp_fused_bwd(dev(x, dt), dev(dy, dt), dev(gam), dev(bet), W1d, dev(b1), W2Td, dg, db, rowscale=dev(rs), rows_per_scale=tps,
                                           dyw=dyw, dyw_scale=dev(dyw_scale) if dyw is not None else None, geom=geom)
    check(dx1, xr.grad, dt, "mlp fused dx1", 2)
    check(x1n, xn_ref.detach(), dt, "mlp fused x1n")
    check(hact, h_ref.detach(), dt, "mlp fused hact")
    check(dg, gr.grad, dt, "mlp fused dgamma", 2)
    check(db, br.grad, dt, "mlp fused dbeta", 2)
    # dh is the A operand of dW1 = dh^T x1n (and db1 = column sums): check it through those products in fp32
    dW1 = dh.float().T @ x1n.float()
    check(dW1, W1r.grad, dt, "mlp fused dW1 via dh", 2)
    check(dh.float().sum(0), b1r.grad, dt, "mlp fused db1 via dh", 2)
    scale_rows = rs.repeat_interleave(tps)[:M, None].cuda()
    dW2 = (dev(dy, dt).float() * scale_rows).T @ hact.float()
    check(dW2, W2r.grad, dt, "mlp fused dW2 via hact", 2)
    if dyw is not None:
        g_ref = torch.empty(geom.rows, C, dtype=dt, device="cuda")
        ops.window_gather_scale(dx1, g_ref, dev(dyw_scale), C, geom)
        assert torch.equal(dyw, g_ref), "window-ordered second output (incl. zeroed pad rows)"


# ------------------------------------------------------------------------------------------------
# decoder1 forward: ConvTranspose3d(96 -> 48, k = s = 4) composed with the 3x3x3 conv that follows (csrc/cconv.hip) against the two
# reference ops in fp32 on the bf16-rounded inputs (unetr_block.py:151-158 + :35-44), incl. the border shell (bias reaches fewer taps
# there), the fused InstanceNorm statistics, several blocks per sample and per workgroup, and the two-step HIP path it replaces
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,v", [(1, 8), (2, 8), (1, 16), (3, 24)])
def test_cconv_forward_matches_convtranspose_then_conv(B, v):
    ops = _ops()
    dt = torch.bfloat16
    x = q(rnd(B, v, v, v, 96, seed=1), dt)
    Wt = rnd(96, 48, 4, 4, 4, seed=2, scale=96 ** -0.5)
    W1 = rnd(48, 48, 3, 3, 3, seed=3, scale=(27 * 48) ** -0.5)
    bt = rnd(48, seed=4, scale=0.5)
    Wcp = torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda")
    delta = torch.empty(27, 48, device="cuda")
    ops.cconv_pack(dev(Wt), dev(W1), dev(bt), Wcp, delta)
    acc = torch.full((B, 48, 2), 7.0, dtype=torch.float64, device="cuda")
    y = ops.cconv_fwd(dev(x, dt), Wcp, delta, B, v, stats_acc=acc)
    torch.cuda.synchronize()
    u = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), Wt, bt, stride=4)
    ref = F.conv3d(u, W1, None, padding=1)
    ref = ref - torch.einsum("o,codhw->c", bt, W1)[None, :, None, None, None]     # the interior bias constant (removed by the InstanceNorm that follows)
    ref = ref.permute(0, 2, 3, 4, 1)
    check(y, ref, dt, f"cconv fwd B={B} v={v}")
    # the border shell on its own (where the bias table matters): faces of the volume
    Fv = 4 * v
    for sl in ((slice(None), 0), (slice(None), Fv - 1), (slice(None), slice(None), 0), (slice(None), slice(None), slice(None), Fv - 1)):
        check(y[sl], ref[sl], dt, "cconv fwd border face")
    yq = y.float().cpu().double()
    check(acc[:, :, 0].float(), yq.sum(dim=(1, 2, 3)).float(), torch.float32, "cconv fused IN sum", 5)
    check(acc[:, :, 1].float(), (yq * yq).sum(dim=(1, 2, 3)).float(), torch.float32, "cconv fused IN sumsq", 5)
    # against the two-step HIP path (bf16 u, then the 48 -> 48 LDS-halo conv): same function, different rounding points
    wt_p = _pack_via_kernel(Wt, 4, dt, Wt.numel())
    cat = torch.empty(B * Fv ** 3, 48, dtype=dt, device="cuda")
    ops.upconv_fwd(dev(x, dt).view(-1, 96), wt_p.view(64 * 48, 96), dev(bt), cat, B, v, 4, 96, 48)
    wk = _pack_via_kernel(W1, 6, dt, 41 * 3 * 64 * 8)
    y2 = ops.conv3d_k3_c48(cat.view(B, Fv, Fv, Fv, 48), wk)
    const = torch.einsum("o,codhw->c", bt, W1)
    assert_close(y.float().cpu(), y2.float().cpu() - const, 2e-2, "cconv vs two-step HIP path (interior + border)", elem_mult=2.0)


@pytest.mark.parametrize("B,v", [(1, 8), (2, 8), (3, 16), (2, 24)])
def test_cconv_output_mean_from_the_coarse_tensor_and_centered_forward(B, v):
    """nmh_cconv_output_mean: the per-(sample, channel) mean of the composed conv's output computed from the COARSE tensor (the output is linear in it) == the mean
    of what nmh_cconv_fwd stores, to the rounding of its bf16 outputs; nmh_cconv_fwd_centered: z == lrelu(y1 - mean) and the statistics of y1 - mean"""
    ops = _ops()
    dt = torch.bfloat16
    x = q(rnd(B, v, v, v, 96, seed=1) + 0.3 * rnd(B, 1, 1, 1, 96, seed=5), dt)   # per-sample, per-channel offsets: the mean is far from zero
    Wt = rnd(96, 48, 4, 4, 4, seed=2, scale=96 ** -0.5)
    W1 = rnd(48, 48, 3, 3, 3, seed=3, scale=(27 * 48) ** -0.5)
    bt = rnd(48, seed=4, scale=0.5)
    Wcp = torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda")
    delta = torch.empty(27, 48, device="cuda")
    ws = torch.empty(ops.cconv_pack_ws_floats(), device="cuda")
    mtab = torch.full((27, 96, 48), 3.0, device="cuda")     # zeroed by the entry
    ops.cconv_pack_centered(dev(Wt), dev(W1), dev(bt), Wcp, delta, ws, mtab)
    Wcp0 = torch.empty_like(Wcp)
    ops.cconv_pack(dev(Wt), dev(W1), dev(bt), Wcp0, torch.empty_like(delta), ws)
    assert torch.equal(Wcp, Wcp0)
    xd = dev(x, dt)
    acc = torch.zeros((B, 48, 2), dtype=torch.float64, device="cuda")
    y = ops.cconv_fwd(xd, Wcp, delta, B, v, stats_acc=acc)
    mean = ops.cconv_output_mean(xd, mtab, delta, B, v)
    torch.cuda.synchronize()
    V = (4 * v) ** 3
    yd = y.double().reshape(B, V, 48)
    std = yd.std(dim=1)
    err = ((mean.double() - yd.mean(dim=1)).abs() / std).max().item()
    assert err < 2e-4, err     # (bf16 rounding of 64 v^3 stored outputs averages out; the fp32 accumulator sums give the same to 1e-6)
    err_acc = ((mean.double() - acc[:, :, 0] / V).abs() / std).max().item()
    assert err_acc < 2e-5, err_acc
    acc2 = torch.full((B, 48, 2), 9.0, dtype=torch.float64, device="cuda")
    z = ops.cconv_fwd_centered(xd, Wcp, delta, mean, B, v, stats_acc=acc2)
    torch.cuda.synchronize()
    # z against the definition on the fp32 composed output: recomputed here from the reference ops
    u = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), Wt, bt, stride=4)
    ref = F.conv3d(u, W1, None, padding=1) - torch.einsum("o,codhw->c", bt, W1)[None, :, None, None, None]
    t = ref.permute(0, 2, 3, 4, 1) - mean.cpu()[:, None, None, None, :]
    check(z, F.leaky_relu(t, 0.01), dt, "cconv centered z", 2)
    # statistics of t: sum ~ 0 (the predicted mean IS the mean), sum of squares = V * var
    assert (acc2[:, :, 0].abs() / (V * std)).max().item() < 2e-5
    check((acc2[:, :, 1] / V).float(), (yd.var(dim=1, unbiased=False)).float(), torch.float32, "cconv centered variance", 50)


@pytest.mark.parametrize("B,D,H,W", [(1, 16, 16, 16), (2, 20, 24, 40), (3, 32, 32, 32), (1, 7, 9, 19)])
def test_conv48_input_gradient_with_fused_instnorm_backward_sums(B, D, H, W):
    """nmh_conv3d_k3_c48_bwd_reduce (decoder1 conv2's input gradient, unetr_block.py:60-63 backward): the same dX as the plain launch, bit for bit, and
    the InstanceNorm-backward sums (sum g, sum g * yhat) of the separate reduce pass over (dX, y1) -- including partial tiles and several samples
    per workgroup (the per-sample flush of the LDS accumulators)"""
    ops = _ops()
    dt = torch.bfloat16
    w = rnd(48, 48, 3, 3, 3, seed=1, scale=(27 * 48) ** -0.5)
    wkd = _pack_via_kernel(w, 7, dt, 41 * 3 * 64 * 8)
    dy = dev(q(rnd(B, D, H, W, 48, seed=2), dt), dt)
    y1 = dev(q(rnd(B, D, H, W, 48, seed=3) * 1.5 + 0.3, dt), dt)
    V = D * H * W
    st = torch.empty(B, 48, 2, device="cuda")
    ops.instnorm_stats(y1.view(B * V, 48), st, ops.acc_zeros((B, 48, 2), "cuda"), B, V, 48)
    ref_dx = ops.conv3d_k3_c48(dy, wkd)
    ref_sums = torch.zeros(B, 48, 2, dtype=torch.float64, device="cuda")
    ops.instnorm_bwd_reduce(ref_dx.view(B * V, 48), None, y1.view(B * V, 48), st, ref_sums, B, V, 48, rmode=0)
    sums = torch.full((B, 48, 2), 5.0, dtype=torch.float64, device="cuda")          # zeroed by the entry
    dx = ops.conv3d_k3_c48_bwd_reduce(dy, wkd, y1, st, sums)
    torch.cuda.synchronize()
    assert torch.equal(dx, ref_dx)
    # fp32 partial sums in a different order: relative to the size of the summands
    scale = (ref_dx.float().abs().sum(dim=(1, 2, 3)) / V).clamp_min(1e-6).cpu()      # [B,48]
    for k in range(2):
        err = ((sums[:, :, k] - ref_sums[:, :, k]).abs().cpu() / (scale * V)).max().item()
        assert err < 2e-5, (k, err)
    # against the definition, fp64 on the host
    yd, dd = y1.double().cpu(), dx.double().cpu()
    mean, rstd = st[:, :, 0].double().cpu(), st[:, :, 1].double().cpu()
    t = yd - mean[:, None, None, None, :]
    g = dd * torch.where(t > 0, 1.0, 0.01)
    check(sums[:, :, 0].float(), g.sum(dim=(1, 2, 3)).float(), torch.float32, "fused IN-backward sum g", 20)
    check(sums[:, :, 1].float(), (g * t * rstd[:, None, None, None, :]).sum(dim=(1, 2, 3)).float(), torch.float32, "fused IN-backward sum g yhat", 20)


@pytest.mark.parametrize("B,v", [(1, 8), (2, 8), (1, 16), (3, 24), (2, 40)])
def test_upconv4_forward_matches_conv_transpose(B, v):
    """decoder1's transpose conv as the persistent register-fragment kernel (csrc/cconv.hip upconv4): against F.conv_transpose3d in fp32 on the
    bf16-rounded operands, and BIT-exact against the GEMM + pixel-shuffle kernel it replaces (same bf16 weights, fp32 accumulation over the same
    96 products... in a different order: so within one bf16 ulp, not equal)"""
    ops = _ops()
    dt = torch.bfloat16
    Fv = 4 * v
    x = q(rnd(B, v, v, v, 96, seed=11), dt)
    Wt = rnd(96, 48, 4, 4, 4, seed=12, scale=96 ** -0.5)
    bt = rnd(48, seed=13, scale=0.5)
    ws = torch.empty(ops.cconv_pack_ws_floats(), dtype=torch.float32, device="cuda")
    Wcp = torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda")
    delta = torch.empty(27, 48, device="cuda")
    ops.cconv_pack(dev(Wt), dev(rnd(48, 48, 3, 3, 3, seed=3)), dev(bt), Wcp, delta, ws)
    Wup = torch.empty(ops.upconv4_pack_numel(), dtype=dt, device="cuda")
    ops.upconv4_pack(ws, Wup)
    u = torch.full((B * Fv ** 3, 48), 3.0, dtype=dt, device="cuda")
    ops.upconv4_fwd(dev(x, dt), Wup, dev(bt), u, B, v)
    torch.cuda.synchronize()
    ref = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), q(Wt, dt), bt, stride=4).permute(0, 2, 3, 4, 1)
    check(u.view(B, Fv, Fv, Fv, 48), ref, dt, f"upconv4 fwd B={B} v={v}")
    wt_p = _pack_via_kernel(Wt, 4, dt, Wt.numel())
    cat = torch.empty(B * Fv ** 3, 48, dtype=dt, device="cuda")
    ops.upconv_fwd(dev(x, dt).view(-1, 96), wt_p.view(64 * 48, 96), dev(bt), cat, B, v, 4, 96, 48)
    assert_close(u.float().cpu(), cat.float().cpu(), 8e-3, "upconv4 vs GEMM + pixel shuffle")


@pytest.mark.parametrize("B,v", [(1, 8), (2, 16), (1, 24), (2, 40)])
def test_cconv_wgrad_matches_autograd_of_the_two_ops(B, v):
    """conv1.weight gradient through the composed ConvTranspose o conv (csrc/cconv.hip: G blocks + chain rule + the bias term carried by the
    border voxels) against autograd of conv3d(conv_transpose3d(x)) in fp32 on the bf16-rounded operands, and against the two-step HIP weight
    gradient (48 -> 48 LDS-halo kernel on the bf16 up-sampled map).  dy1 is the input gradient of an InstanceNorm (zero sums per sample and
    channel), which is what conv1 feeds in the model and what the entry's bias term relies on."""
    ops = _ops()
    dt = torch.bfloat16
    Fv = 4 * v
    x = q(rnd(B, v, v, v, 96, seed=1), dt)
    Wt = rnd(96, 48, 4, 4, 4, seed=2, scale=96 ** -0.5)
    W1 = rnd(48, 48, 3, 3, 3, seed=3, scale=(27 * 48) ** -0.5)
    bt = rnd(48, seed=4, scale=0.5)
    gup = rnd(B, 48, Fv, Fv, Fv, seed=5)
    # an InstanceNorm backward makes dy1 zero-sum per (sample, channel); round to bf16 the way the pipeline stores it
    y_lin = F.conv3d(F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), Wt, bt, stride=4), W1, None, padding=1).requires_grad_(True)
    (F.instance_norm(y_lin) * gup).sum().backward()
    dy = q(y_lin.grad, dt)                                    # (B,48,F,F,F)
    W1r = W1.clone().requires_grad_(True)
    Wtr, btr = Wt.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    u = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), Wtr, btr, stride=4)
    (F.conv3d(u, W1r, None, padding=1) * dy).sum().backward()
    # HIP: pack (fills the workspace with the transposed Wt), then the composed weight gradient
    Wcp = torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda")
    delta = torch.empty(27, 48, device="cuda")
    pws = torch.empty(ops.cconv_pack_ws_floats(), device="cuda")
    ops.cconv_pack(dev(Wt), dev(W1), dev(bt), Wcp, delta, pws)
    dW = torch.zeros(48, 48, 3, 3, 3, device="cuda")
    dy_cl = dev(dy.permute(0, 2, 3, 4, 1).contiguous(), dt)
    ops.cconv_wgrad(dev(x, dt), dy_cl, pws, dev(bt), dW, B, v)
    check(dW, W1r.grad, dt, f"cconv wgrad B={B} v={v}", 2)
    # accumulation contract (+=) and the two-step HIP kernel on the bf16 up-sampled map
    ops.cconv_wgrad(dev(x, dt), dy_cl, pws, dev(bt), dW, B, v)
    check(dW, 2 * W1r.grad, dt, "cconv wgrad accumulates", 2)
    wt_p = _pack_via_kernel(Wt, 4, dt, Wt.numel())
    cat = torch.empty(B * Fv ** 3, 48, dtype=dt, device="cuda")
    ops.upconv_fwd(dev(x, dt).view(-1, 96), wt_p.view(64 * 48, 96), dev(bt), cat, B, v, 4, 96, 48)
    dW2 = torch.zeros(48, 48, 3, 3, 3, device="cuda")
    ops.conv3d_k3_c48_wgrad(dy_cl, cat.view(B, Fv, Fv, Fv, 48), dW2)
    assert_close(dW.float().cpu() / 2, dW2.float().cpu(), 2e-2, "cconv wgrad vs two-step HIP weight gradient")
    # the transpose conv's own parameter gradients through conv1, from the same G blocks (what cconv_dgrad's callers need: conv1's input
    # gradient on the fine grid is not formed there); += contract checked by the pre-filled buffers
    dW3 = torch.zeros(48, 48, 3, 3, 3, device="cuda")
    dWt = torch.full((96, 48, 4, 4, 4), 0.25, device="cuda")
    dbt = torch.full((48,), -0.5, device="cuda")
    ops.cconv_wgrad(dev(x, dt), dy_cl, pws, dev(bt), dW3, B, v, dWt=dWt, dbt=dbt)
    check(dW3, W1r.grad, dt, "cconv wgrad (with dWt)", 2)
    check(dWt - 0.25, Wtr.grad, dt, f"transpose conv weight gradient through the composition B={B} v={v}", 2)
    check(dbt + 0.5, btr.grad, dt, "transpose conv bias gradient through the composition (border classes)", 3)
    # the two phases of the entry (persistent kernel / the small launches behind it, which a caller may issue on a side stream) == the one-call form
    dW4, dWt4, dbt4 = torch.zeros_like(dW3), torch.full_like(dWt, 0.25), torch.full_like(dbt, -0.5)
    ops.cconv_wgrad(dev(x, dt), dy_cl, pws, dev(bt), dW4, B, v, dWt=dWt4, dbt=dbt4, phase=1)
    assert float(dW4.abs().max()) == 0.0                      # phase 1 only fills the workspace
    ops.cconv_wgrad(dev(x, dt), dy_cl, pws, dev(bt), dW4, B, v, dWt=dWt4, dbt=dbt4, phase=2)
    assert_close(dW4.cpu(), dW3.cpu(), 1e-5, "cconv wgrad in two phases (dW1)")
    assert_close((dWt4 - 0.25).cpu(), (dWt - 0.25).cpu(), 1e-5, "cconv wgrad in two phases (dWt)")


@pytest.mark.parametrize("B,v", [(1, 8), (2, 8), (1, 16), (3, 24), (2, 40)])
def test_cconv_dgrad_matches_autograd_of_the_two_ops(B, v):
    """input gradient through the composition (csrc/cconv.hip cconv_dgrad: a stride-4 convolution of dy1 with the 6^3 kernel of transposed composed
    blocks) against autograd of conv3d(conv_transpose3d(x)) in fp32, with and without the pre-existing dx of the residual branch, and against the
    two-step HIP path (48 -> 48 LDS-halo input gradient on the fine grid, then the transpose conv's)"""
    ops = _ops()
    dt = torch.bfloat16
    Fv = 4 * v
    Wt = rnd(96, 48, 4, 4, 4, seed=2, scale=96 ** -0.5)
    W1 = rnd(48, 48, 3, 3, 3, seed=3, scale=(27 * 48) ** -0.5)
    bt = rnd(48, seed=4, scale=0.5)
    dy = q(rnd(B, Fv, Fv, Fv, 48, seed=6), dt)
    xr = torch.zeros(B, 96, v, v, v, requires_grad=True)
    (F.conv3d(F.conv_transpose3d(xr, Wt, bt, stride=4), W1, None, padding=1) * dy.permute(0, 4, 1, 2, 3)).sum().backward()
    ref = xr.grad.permute(0, 2, 3, 4, 1).reshape(-1, 96)
    Wcp = torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda")
    delta = torch.empty(27, 48, device="cuda")
    ops.cconv_pack(dev(Wt), dev(W1), dev(bt), Wcp, delta)
    Wdp = torch.empty(ops.cconv_dgrad_pack_numel(), dtype=dt, device="cuda")
    ops.cconv_dgrad_pack(Wcp, Wdp)
    dy_d = dev(dy, dt)
    dx = ops.cconv_dgrad(dy_d, Wdp, B, v)
    torch.cuda.synchronize()
    check(dx, ref, dt, f"cconv dgrad B={B} v={v}")
    # border cells on their own (zero padding of the fine grid = missing window voxels)
    d5, r5 = dx.view(B, v, v, v, 96), ref.view(B, v, v, v, 96)
    for sl in ((slice(None), 0), (slice(None), v - 1), (slice(None), slice(None), 0), (slice(None), slice(None), slice(None), v - 1)):
        check(d5[sl], r5[sl], dt, "cconv dgrad border face")
    add = q(rnd(B * v ** 3, 96, seed=7), dt)
    buf = dev(add, dt).clone()
    ops.cconv_dgrad(dy_d, Wdp, B, v, add=buf, out=buf)           # in place: the residual branch's dx is already there
    check(buf, ref + add, dt, "cconv dgrad added to an existing dx")
    # two-step HIP path
    wkd = _pack_via_kernel(W1, 7, dt, 41 * 3 * 64 * 8)
    dcat = ops.conv3d_k3_c48(dy_d.view(B, Fv, Fv, Fv, 48), wkd)
    wtd = _pack_via_kernel(Wt, 5, dt, Wt.numel())
    dx2 = torch.empty(B * v ** 3, 96, dtype=dt, device="cuda")
    ops.upconv_dgrad(dcat.view(-1, 48), wtd.view(96, 64 * 48), dx2, B, v, 4, 96, 48)
    assert_close(dx.float().cpu(), dx2.float().cpu(), 2e-2, "cconv dgrad vs two-step HIP input gradient", elem_mult=2.0)
