"""Fused Swin-block kernels (csrc/swin_block.hip: nmh_swin_*) against the oracle's fp32 restatement of a Swin block
(oracle/mae3d_oracle.py SwinBlock3D = swin_mae3d.py:310-369 with shifted_window_attention :27-197) and against the unfused HIP chain,
bf16, all six window geometries of tests/test_kernels_gpu.py (plain, shifted, padded, padded + shifted, window >= volume, ragged axes),
widths 96 / 192 / 384, forward and every gradient.  Tolerances are bf16 rounding (tests/_metrics.py three-part metric)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._metrics import assert_close  # noqa: E402

BF = torch.bfloat16
TOL = 3e-2
GEOMS = [((2, 8, 8, 8), 0), ((2, 8, 8, 8), 2), ((1, 5, 5, 5), 2), ((2, 10, 10, 10), 2), ((1, 2, 2, 2), 2), ((1, 6, 8, 4), 2)]
WIDTHS = [96, 192, 384]


def _ops():
    from nerf_mae_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 1000)
    return torch.randn(*shape, generator=g) * scale


def qb(t):
    return t.to(BF).float()


def dev(t, dt=None):
    return (t if dt is None else t.to(dt)).cuda().contiguous()


def make_block(C, shift, seed=0):
    """oracle block with bf16-representable Linear weights (the fused kernels see the same values) and non-trivial biases / affine parameters"""
    from oracle import mae3d_oracle as O
    torch.manual_seed(seed)
    blk = O.SwinBlock3D(C, C // 32, [shift] * 3, 0.0)
    with torch.no_grad():
        for name, p in blk.named_parameters():
            if p.dim() == 2 and "table" not in name:
                p.copy_(qb(torch.randn_like(p) * (p.shape[1] ** -0.5)))
            elif "table" in name:
                p.copy_(torch.randn_like(p) * 0.5)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            else:
                p.copy_(0.2 * torch.randn_like(p))
    return blk.double()


def streams(ops, blk, C, kinds):
    """weight streams of the requested kinds from the block's fp32 parameters"""
    w = {"qkv": dev(blk.attn.qkv.weight.float()), "proj": dev(blk.attn.proj.weight.float()), "fc1": dev(blk.mlp[0].weight.float()), "fc2": dev(blk.mlp[3].weight.float())}
    src = {ops.SWIN_ATTN_FWD: ("qkv", "proj"), ops.SWIN_MLP_FWD: ("fc1", "fc2")}
    out, items = {}, []
    for k in kinds:
        out[k] = torch.empty(ops.swin_stream_numel(k, C), dtype=BF, device="cuda")
        a, b = src[k]
        items.append((w[a], w[b] if b else None, out[k], k, C))
    arr = ops.swin_pack_items(items)
    ops.swin_pack(arr)
    torch.cuda.synchronize()
    return out, w


def reference(blk, x, sd1, sd2):
    """fp64 block with explicit per-sample branch scales; returns x1, x2 (leaf x requires grad)"""
    B = x.shape[0]
    s1 = sd1.double().view(B, 1, 1, 1, 1)
    s2 = sd2.double().view(B, 1, 1, 1, 1)
    x1 = x + s1 * blk.attn(blk.norm1(x))
    x2 = x1 + s2 * blk.mlp(blk.norm2(x1))
    return x1, x2


@pytest.mark.parametrize("C", WIDTHS)
@pytest.mark.parametrize("shape,shift", GEOMS)
def test_swin_block_forward(C, shape, shift):
    ops = _ops()
    B, H, W, D = shape
    geom = ops.WinGeom(B, H, W, D, [shift] * 3)
    T, heads = geom.tokens, C // 32
    tps = T // B
    blk = make_block(C, shift)
    x = qb(rnd(B, H, W, D, C, seed=3) * 1.3 + 0.1)
    sd1 = torch.tensor([1.0 / 0.9, 0.0] if B == 2 else [1.0 / 0.95])
    sd2 = torch.tensor([1.0, 1.0 / 0.8] if B == 2 else [1.0 / 0.9])
    with torch.no_grad():
        x1_ref, x2_ref = reference(blk, x.double(), sd1, sd2)
    st, w = streams(ops, blk, C, [ops.SWIN_ATTN_FWD, ops.SWIN_MLP_FWD])
    f = lambda p: dev(p.detach().float())
    xd = dev(x.view(T, C), BF)
    x1, xnw, mean1, rstd1, qkv, o, lse = ops.swin_attn_fwd(xd, f(blk.norm1.weight), f(blk.norm1.bias), st[ops.SWIN_ATTN_FWD], f(blk.attn.qkv.bias),
                                                           f(blk.attn.relative_position_bias_table), f(blk.attn.proj.bias), geom, rowscale=dev(sd1), rows_per_scale=tps)
    torch.cuda.synchronize()
    # --- the saved tensors against the unfused HIP kernels (same layouts by contract)
    xnw_u = torch.empty_like(xnw)
    m_u, r_u = torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    ops.layernorm_fwd(xd, f(blk.norm1.weight), f(blk.norm1.bias), xnw_u, m_u, r_u, geom.rows, C, src_mode=1, geom=geom)
    assert_close(xnw, xnw_u.float().cpu(), 1e-2, "xnw")
    assert_close(mean1, m_u.cpu(), 1e-4, "mean1")
    assert_close(rstd1, r_u.cpu(), 1e-4, "rstd1")
    qkv_u = ops.gemm_nt(xnw_u, dev(w["qkv"], BF), bias=f(blk.attn.qkv.bias))
    assert_close(qkv, qkv_u.float().cpu(), TOL, "qkv")
    o_u, lse_u = torch.empty_like(o), torch.empty_like(lse)
    ops.window_attn_fwd(qkv, f(blk.attn.relative_position_bias_table), o_u, lse_u, heads, C, geom)   # on the fused kernel's own qkv: isolates the attention core
    assert_close(o, o_u.float().cpu(), TOL, "o")
    assert_close(lse, lse_u.cpu(), 1e-3, "lse")
    assert_close(x1, x1_ref.view(T, C), TOL, "x1")
    # --- MLP branch on the fused x1: one workgroup per row tile, and (C = 384) two per tile with the partial sums combined by the last to arrive
    x1f = x1.float().cpu().double()
    with torch.no_grad():
        n2 = blk.norm2(x1f)
        hp_ref = blk.mlp[0](n2)
        x2_own = x1f + sd2.double().repeat_interleave(tps)[:, None] * blk.mlp(n2)
    mu = x1f.mean(-1)
    outs = {}
    for split in (False, True):
        for rep in range(3 if split else 1):
            x2, x1n, hp, mean2, rstd2, hact = ops.swin_mlp_fwd(x1, f(blk.norm2.weight), f(blk.norm2.bias), st[ops.SWIN_MLP_FWD], f(blk.mlp[0].bias), f(blk.mlp[3].bias),
                                                               rowscale=dev(sd2), rows_per_scale=tps, want_hact=True, split=split)
            torch.cuda.synchronize()
            if rep:
                assert torch.equal(x2, outs[split]), "the split kernel is not reproducible"
            outs[split] = x2
        name = f" (split={split})"
        assert_close(x1n, n2, 1e-2, "x1n" + name)
        assert_close(hp, hp_ref, TOL, "hp" + name)
        assert_close(hact, torch.nn.functional.gelu(hp.float().cpu().double()), TOL, "hact" + name)
        assert_close(x2, x2_own, TOL, "x2 (from the fused x1)" + name)
        assert_close(x2, x2_ref.view(T, C), TOL, "x2" + name, elem_mult=2.0)
        assert_close(mean2, mu, 1e-4, "mean2" + name)
        assert_close(rstd2, (x1f.var(-1, unbiased=False) + 1e-5).rsqrt(), 1e-4, "rstd2" + name)
    if C == 384:
        assert ops.swin_mlp_split_ws(T, C, x1.device) is not None
        assert int(ops.swin_mlp_split_ws(T, C, x1.device)[:1024].sum()) == 0      # the arrival counters are back at zero
        assert_close(outs[True], outs[False].float().cpu(), 1e-2, "split vs one workgroup per tile")   # (fp32 sums in another order, one bf16 rounding)


@pytest.mark.parametrize("C", WIDTHS)
@pytest.mark.parametrize("shape,shift", GEOMS)
def test_swin_block_backward(C, shape, shift):
    """the product's backward of a block behind the fused forward: the unfused chain (nmh_gemm_nt with the GELU' epilogue, nmh_layernorm_bwd with the
    window-ordered second output, nmh_window_attn_bwd, ...) run on the tensors the FUSED forward kernels saved -- the input gradient, every LayerNorm /
    bias-table gradient and, through plain matmuls of the operands the kernels write, every Linear weight / bias gradient against fp64 autograd of the oracle block"""
    ops = _ops()
    B, H, W, D = shape
    geom = ops.WinGeom(B, H, W, D, [shift] * 3)
    T, heads = geom.tokens, C // 32
    tps = T // B
    blk = make_block(C, shift, seed=1)
    x = qb(rnd(B, H, W, D, C, seed=4) * 1.3 + 0.1)
    dy = qb(rnd(B, H, W, D, C, seed=5))
    sd1 = torch.tensor([1.0 / 0.9, 0.5] if B == 2 else [1.0 / 0.95])
    sd2 = torch.tensor([1.0, 1.0 / 0.8] if B == 2 else [1.0 / 0.9])
    xr = x.double().requires_grad_(True)
    x1_ref, x2_ref = reference(blk, xr, sd1, sd2)
    x1_ref.retain_grad()
    x2_ref.backward(dy.double())
    st, w = streams(ops, blk, C, [ops.SWIN_ATTN_FWD, ops.SWIN_MLP_FWD])
    f = lambda p: dev(p.detach().float())
    xd, dyd = dev(x.view(T, C), BF), dev(dy.view(T, C), BF)
    sd1d, sd2d = dev(sd1), dev(sd2)
    g1, b1n, g2, b2n, tab = f(blk.norm1.weight), f(blk.norm1.bias), f(blk.norm2.weight), f(blk.norm2.bias), f(blk.attn.relative_position_bias_table)
    x1, xnw, mean1, rstd1, qkv, o, lse = ops.swin_attn_fwd(xd, g1, b1n, st[ops.SWIN_ATTN_FWD], f(blk.attn.qkv.bias), tab, f(blk.attn.proj.bias), geom, rowscale=sd1d, rows_per_scale=tps)
    x2, x1n, hp, mean2, rstd2, hact = ops.swin_mlp_fwd(x1, g2, b2n, st[ops.SWIN_MLP_FWD], f(blk.mlp[0].bias), f(blk.mlp[3].bias), rowscale=sd2d, rows_per_scale=tps,
                                                      want_hact=True)
    z = lambda *s: torch.zeros(*s, device="cuda")
    # ---- MLP branch
    dg2, db2 = z(C), z(C)
    dh = ops.gemm_nt(dyd, dev(w["fc2"].T.contiguous(), BF), act=2, C2=hp, rowscale=sd2d, rows_per_scale=tps)
    dx1n = ops.gemm_nt(dh, dev(w["fc1"].T.contiguous(), BF))
    dx1, dyw = torch.empty_like(x1), torch.empty(geom.rows, C, dtype=BF, device="cuda")
    ops.layernorm_bwd(dx1n, x1, g2, mean2, rstd2, dx1, dg2, db2, T, C, dres=dyd, geom=geom, tokens_per_sample=tps, dyw=dyw, dyw_scale=sd1d)
    torch.cuda.synchronize()
    assert_close(hact, torch.nn.functional.gelu(hp.float().cpu().double()), 1e-2, "hact")
    assert_close(dx1, x1_ref.grad.view(T, C), TOL, "dx1", elem_mult=2.0)
    assert_close(dg2, blk.norm2.weight.grad, TOL, "dgamma2")
    assert_close(db2, blk.norm2.bias.grad, TOL, "dbeta2")
    dyw_u = torch.empty_like(dyw)
    ops.window_gather_scale(dx1, dyw_u, sd1d, C, geom)
    assert_close(dyw, dyw_u.float().cpu(), 1e-2, "dyw (window-ordered, scaled copy)")
    rs2 = sd2.double().repeat_interleave(tps)[:, None]
    dhf, hactf, x1nf, dyf = dh.float().cpu().double(), hact.float().cpu().double(), x1n.float().cpu().double(), dy.view(T, C).double()
    assert_close(dhf.T @ x1nf, blk.mlp[0].weight.grad, 2 * TOL, "fc1 weight gradient from (dh, x1n)")   # (products of two bf16-rounded operands over as few as 8 rows)
    assert_close(dhf.sum(0), blk.mlp[0].bias.grad, 2 * TOL, "fc1 bias gradient")
    assert_close((rs2 * dyf).T @ hactf, blk.mlp[3].weight.grad, 2 * TOL, "fc2 weight gradient from (dy, hact)")
    # ---- attention branch: dO + attention core
    dtab = z(343, heads)
    do_u = ops.gemm_nt(dyw, dev(w["proj"].T.contiguous(), BF))
    dqkv = torch.empty_like(qkv)
    ops.window_attn_bwd(qkv, tab, do_u, lse, dqkv, dtab, heads, C, geom)
    torch.cuda.synchronize()
    assert_close(dtab, blk.attn.relative_position_bias_table.grad, 2 * TOL, "d(bias table) vs autograd")
    dywf, of, dqf, xnwf = dyw.float().cpu().double(), o.float().cpu().double(), dqkv.float().cpu().double(), xnw.float().cpu().double()
    assert_close(dywf.T @ of, blk.attn.proj.weight.grad, 2 * TOL, "proj weight gradient from (dyw, o)")
    assert_close(dqf.T @ xnwf, blk.attn.qkv.weight.grad, 2 * TOL, "qkv weight gradient from (dqkv, xnw)")
    assert_close(dqf.sum(0), blk.attn.qkv.bias.grad, 2 * TOL, "qkv bias gradient", elem_mult=2.0)
    # ---- QKV + LayerNorm-1 backward
    dg1, db1 = z(C), z(C)
    dxnw = ops.gemm_nt(dqkv, dev(w["qkv"].T.contiguous(), BF))
    dx = torch.empty_like(xd)
    ops.layernorm_bwd(dxnw, xd, g1, mean1, rstd1, dx, dg1, db1, T, C, src_mode=1, geom=geom, dres=dx1)
    torch.cuda.synchronize()
    assert_close(dx, xr.grad.view(T, C), 2 * TOL, "dx vs autograd", elem_mult=2.0)
    assert_close(dg1, blk.norm1.weight.grad, 2 * TOL, "dgamma1 vs autograd")
    assert_close(db1, blk.norm1.bias.grad, 2 * TOL, "dbeta1 vs autograd")


def test_swin_mlp_split_full_size_is_reproducible_and_matches_unsplit():
    """stage-2 shape of the headline config (8 grids: 8000 rows = 125 row tiles, 250 workgroups): 40 back-to-back launches of the two-workgroups-per-tile
    kernel on one stream give bit-identical outputs (the combine does not depend on which workgroup arrives last, and the counters re-arm), and they
    agree with the one-workgroup-per-tile kernel to bf16 rounding"""
    ops = _ops()
    C, M, tps = 384, 8000, 1000
    g = torch.Generator().manual_seed(5)
    x1 = (torch.randn(M, C, generator=g) * 1.2).to(BF).cuda()
    W1, W2 = torch.randn(4 * C, C, generator=g) * C ** -0.5, torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5
    b1, b2 = torch.randn(4 * C, generator=g).cuda() * 0.1, torch.randn(C, generator=g).cuda() * 0.1
    gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    rs = (torch.rand(M // tps, generator=g) + 0.5).cuda()
    st = torch.empty(ops.swin_stream_numel(ops.SWIN_MLP_FWD, C), dtype=BF, device="cuda")
    ops.swin_pack(ops.swin_pack_items([(W1.cuda(), W2.cuda(), st, ops.SWIN_MLP_FWD, C)]))
    ref = ops.swin_mlp_fwd(x1, gam, bet, st, b1, b2, rowscale=rs, rows_per_scale=tps, split=False)
    outs = [ops.swin_mlp_fwd(x1, gam, bet, st, b1, b2, rowscale=rs, rows_per_scale=tps, split=True) for _ in range(40)]
    torch.cuda.synchronize()
    for i, o in enumerate(outs[1:]):
        assert torch.equal(o[0], outs[0][0]), f"launch {i + 1} differs from launch 0"
    for a, b, name in zip(outs[0], ref, ("x2", "x1n", "hp", "mean", "rstd")):
        assert_close(a, b.float().cpu(), 1e-2, name)
    assert int(ops.swin_mlp_split_ws(M, C, x1.device)[:1024].sum()) == 0
    with torch.no_grad():
        xf = x1.float().cpu().double()
        n2 = torch.nn.functional.layer_norm(xf, (C,), gam.cpu().double(), bet.cpu().double(), 1e-5)
        y = torch.nn.functional.gelu(n2 @ W1.to(BF).double().T + b1.cpu().double()) @ W2.to(BF).double().T + b2.cpu().double()
        x2_ref = xf + rs.cpu().double().repeat_interleave(tps)[:, None] * y
    assert_close(outs[0][0], x2_ref, TOL, "x2 vs fp64")


@pytest.mark.parametrize("C", [96, 384])
@pytest.mark.parametrize("shape,shift", [((2, 10, 10, 10), 2), ((1, 5, 5, 5), 2), ((2, 10, 10, 10), 0), ((1, 6, 8, 4), 2), ((2, 8, 8, 8), 2)])
def test_token_ordered_attention_backward(C, shape, shift):
    """the attention branch's backward with window order confined to the attention kernels (model._BlockFn, padded stages): fused forward with token-ordered
    saves -> proj input gradient on T rows -> nmh_window_attn_bwd_tokens -> qkv input gradient on T rows -> plain LayerNorm backward.  Against (1) the
    window-ordered kernels on the same data, row by row -- the token rows bit-identical -- and (2) fp64 autograd of the oracle block, including the
    weight / bias gradients formed from the token-ordered operands plus the pad rows' share of the qkv bias gradient (nmh_window_pad_rows_colsum)"""
    ops = _ops()
    B, H, W, D = shape
    geom = ops.WinGeom(B, H, W, D, [shift] * 3)
    T, heads = geom.tokens, C // 32
    tps = T // B
    blk = make_block(C, shift, seed=2)
    x = qb(rnd(B, H, W, D, C, seed=6) * 1.3 + 0.1)
    dy1 = qb(rnd(B, H, W, D, C, seed=7))     # gradient arriving at x1 (the MLP branch's backward output in the model)
    sd1 = torch.tensor([1.0 / 0.9, 0.5] if B == 2 else [1.0 / 0.95])
    xr = x.double().requires_grad_(True)
    x1_ref = xr + sd1.double().view(B, 1, 1, 1, 1) * blk.attn(blk.norm1(xr))
    x1_ref.backward(dy1.double())
    st, w = streams(ops, blk, C, [ops.SWIN_ATTN_FWD])
    f = lambda p: dev(p.detach().float())
    z = lambda *s: torch.zeros(*s, device="cuda")
    xd, dx1 = dev(x.view(T, C), BF), dev(dy1.view(T, C), BF)
    sd1d = dev(sd1)
    g1, b1n, tab = f(blk.norm1.weight), f(blk.norm1.bias), f(blk.attn.relative_position_bias_table)
    args = (xd, g1, b1n, st[ops.SWIN_ATTN_FWD], f(blk.attn.qkv.bias), tab, f(blk.attn.proj.bias), geom)
    x1w, xnw, mean1, rstd1, qkv, o_w, lse = ops.swin_attn_fwd(*args, rowscale=sd1d, rows_per_scale=tps)
    x1t, xn_t, mean1t, rstd1t, qkv_t, o_t, lse_t = ops.swin_attn_fwd(*args, rowscale=sd1d, rows_per_scale=tps, token_saves=True)
    torch.cuda.synchronize()
    # window row of every token (the reference's pad -> roll -> partition applied to an index volume)
    from oracle import mae3d_oracle as O
    idx = torch.arange(1, T + 1, dtype=torch.float32).view(B, H, W, D, 1)
    pad = [g_ - s_ for g_, s_ in zip(geom.P, (H, W, D))]
    ip = torch.nn.functional.pad(idx, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    if shift:
        ip = torch.roll(ip, shifts=[-s_ for s_ in geom.shift], dims=(1, 2, 3))
    tok_of_row = O.window_partition(ip).reshape(-1).long() - 1            # -1: pad row
    real = (tok_of_row >= 0).cuda()
    row_of_tok = torch.empty(T, dtype=torch.long)
    row_of_tok[tok_of_row[tok_of_row >= 0]] = torch.nonzero(tok_of_row >= 0).flatten()
    row_of_tok = row_of_tok.cuda()
    assert torch.equal(x1t, x1w) and torch.equal(qkv_t, qkv) and torch.equal(lse_t, lse)
    assert torch.equal(xn_t, xnw[row_of_tok]) and torch.equal(o_t, o_w[row_of_tok]), "token-ordered saves differ from the window-ordered ones"
    # ---- window-ordered chain (the kernels the model used before)
    WpT, WqT = dev(w["proj"].T.contiguous(), BF), dev(w["qkv"].T.contiguous(), BF)
    dyw = torch.empty_like(xnw)
    ops.window_gather_scale(dx1, dyw, sd1d, C, geom)
    do_w = ops.gemm_nt(dyw, WpT)
    dqkv_w, dtab_w = torch.empty_like(qkv), z(343, heads)
    ops.window_attn_bwd(qkv, tab, do_w, lse, dqkv_w, dtab_w, heads, C, geom)
    # ---- token-ordered chain
    # (a) the kernel alone, on the window chain's own dO gathered to token order: the same arithmetic, so the rows must be bit-identical
    dq_x, dq_padx, dtab_x = torch.empty(T, 3 * C, dtype=BF, device="cuda"), torch.full_like(qkv, float("nan")), z(343, heads)
    ops.window_attn_bwd_tokens(qkv, tab, do_w[row_of_tok].contiguous(), lse, dq_x, dq_padx, dtab_x, heads, C, geom)
    torch.cuda.synchronize()
    assert torch.equal(dq_x, dqkv_w[row_of_tok]), "d(qkv) of the tokens differs from the window-ordered kernel's rows"
    if int((~real).sum()):
        assert torch.equal(dq_padx[~real], dqkv_w[~real]), "d(qkv) of the pad rows differs"
        assert bool(torch.isnan(dq_padx[real].float()).all()), "token rows of the pad buffer were written"
    assert_close(dtab_x, dtab_w.cpu(), 1e-3, "d(bias table) (fp32 atomics in another order)")
    # (b) the chain the model runs: dO from the token-ordered gradient with the stochastic-depth scale in the GEMM's epilogue
    do_t = ops.gemm_nt(dx1, WpT, rowscale=sd1d, rows_per_scale=tps)
    dq_t, dq_pad, dtab_t = torch.empty(T, 3 * C, dtype=BF, device="cuda"), torch.full_like(qkv, float("nan")), z(343, heads)
    ops.window_attn_bwd_tokens(qkv, tab, do_t, lse, dq_t, dq_pad, dtab_t, heads, C, geom)
    torch.cuda.synchronize()
    assert_close(do_t, do_w[row_of_tok].float().cpu(), 1e-2, "dO (scaled in the epilogue vs scaled before the product)")
    assert_close(dq_t, dqkv_w[row_of_tok].float().cpu(), 2 * TOL, "d(qkv) of the tokens (dO rounded once instead of twice)")
    assert_close(dtab_t, dtab_w.cpu(), TOL, "d(bias table)")
    dbq = z(3 * C)
    ops.window_pad_rows_colsum(dq_pad, dbq, geom)
    ref_pad = dq_pad[~real].float().sum(0).cpu() if int((~real).sum()) else torch.zeros(3 * C)
    assert_close(dbq, ref_pad, 1e-3, "pad rows' column sums")
    # the grouped entry (all blocks of a stage in one launch): three buffers -> three accumulators, each += its own pad-row sums
    bufs = [dq_pad, (dq_pad.float() * 2).to(BF), (dq_pad.float() * -0.5).to(BF)]
    accs = [z(3 * C), torch.full((3 * C,), 1.0, device="cuda"), z(3 * C)]
    ops.window_pad_rows_colsum_grouped(list(zip(bufs, accs)), geom)
    torch.cuda.synchronize()
    assert_close(accs[0], ref_pad, 1e-3, "grouped pad-row sums, item 0")
    assert_close(accs[1], 1.0 + 2 * ref_pad, 1e-2, "grouped pad-row sums, item 1 (accumulates)")
    assert_close(accs[2], -0.5 * ref_pad, 1e-2, "grouped pad-row sums, item 2")
    dxn = ops.gemm_nt(dq_t, WqT)
    dx, dg1, db1 = torch.empty_like(xd), z(C), z(C)
    ops.layernorm_bwd(dxn, xd, g1, mean1, rstd1, dx, dg1, db1, T, C, dres=dx1)
    torch.cuda.synchronize()
    # ---- against autograd
    assert_close(dx, xr.grad.view(T, C), 2 * TOL, "dx vs autograd", elem_mult=2.0)
    assert_close(dg1, blk.norm1.weight.grad, 2 * TOL, "dgamma1 vs autograd")
    assert_close(db1, blk.norm1.bias.grad, 2 * TOL, "dbeta1 vs autograd")
    assert_close(dtab_t, blk.attn.relative_position_bias_table.grad, 2 * TOL, "d(bias table) vs autograd")
    rs1 = sd1.double().repeat_interleave(tps)[:, None]
    dyf, of, dqf, xnf = dy1.view(T, C).double(), o_t.float().cpu().double(), dq_t.float().cpu().double(), xn_t.float().cpu().double()
    assert_close((rs1 * dyf).T @ of, blk.attn.proj.weight.grad, 2 * TOL, "proj weight gradient from (sd1 dx1, o) in token order")
    assert_close((rs1 * dyf).sum(0), blk.attn.proj.bias.grad, 2 * TOL, "proj bias gradient")
    assert_close(dqf.T @ xnf, blk.attn.qkv.weight.grad, 2 * TOL, "qkv weight gradient from (dqkv, LN1 x) in token order")
    assert_close(dqf.sum(0) + dbq.cpu().double(), blk.attn.qkv.bias.grad, 2 * TOL, "qkv bias gradient = token rows + pad rows", elem_mult=2.0)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,shift,C", [((2, 10, 10, 10), 2, 384), ((1, 5, 5, 5), 2, 768), ((1, 5, 5, 5), 0, 96), ((1, 6, 8, 4), 2, 192), ((2, 8, 8, 8), 2, 96)])
def test_unfused_forward_with_token_ordered_saves(dt, shape, shift, C):
    """nmh_layernorm_fwd_window_tokens / nmh_window_attn_fwd_tokens (the unfused chain's side of the token-ordered backward): the window-ordered LN output is
    the plain kernel's, the token-ordered copy and the scattered attention output are the same rows -- bit for bit -- at their tokens"""
    ops = _ops()
    from oracle import mae3d_oracle as O
    B, H, W, D = shape
    geom = ops.WinGeom(B, H, W, D, [shift] * 3)
    T, heads = geom.tokens, C // 32
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(T, C, generator=g) * 1.3 + 0.1).to(dt).cuda()
    gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    idx = torch.arange(1, T + 1, dtype=torch.float32).view(B, H, W, D, 1)
    pad = [g_ - s_ for g_, s_ in zip(geom.P, (H, W, D))]
    ip = torch.nn.functional.pad(idx, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    if shift:
        ip = torch.roll(ip, shifts=[-s_ for s_ in geom.shift], dims=(1, 2, 3))
    tok_of_row = O.window_partition(ip).reshape(-1).long() - 1
    row_of_tok = torch.empty(T, dtype=torch.long)
    row_of_tok[tok_of_row[tok_of_row >= 0]] = torch.nonzero(tok_of_row >= 0).flatten()
    row_of_tok = row_of_tok.cuda()
    e = lambda *s: torch.empty(*s, dtype=dt, device="cuda")
    xw_ref, m_ref, r_ref = e(geom.rows, C), torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    ops.layernorm_fwd(x, gam, bet, xw_ref, m_ref, r_ref, geom.rows, C, src_mode=1, geom=geom)
    xw, xt, m, r = e(geom.rows, C), torch.full((T, C), float("nan"), dtype=dt, device="cuda"), torch.empty(T, device="cuda"), torch.empty(T, device="cuda")
    ops.layernorm_fwd_window_tokens(x, gam, bet, xw, xt, m, r, C, geom)
    torch.cuda.synchronize()
    assert torch.equal(xw, xw_ref) and torch.equal(m, m_ref) and torch.equal(r, r_ref)
    assert torch.equal(xt, xw_ref[row_of_tok])
    qkv = (torch.randn(geom.rows, 3 * C, generator=g) * 0.7).to(dt).cuda()
    tab = (torch.randn(343, heads, generator=g) * 0.2).cuda()
    o_ref, lse_ref = e(geom.rows, C), torch.empty(geom.rows * heads, device="cuda")
    ops.window_attn_fwd(qkv, tab, o_ref, lse_ref, heads, C, geom)
    o_t, lse = torch.full((T, C), float("nan"), dtype=dt, device="cuda"), torch.empty(geom.rows * heads, device="cuda")
    ops.window_attn_fwd_tokens(qkv, tab, o_t, lse, heads, C, geom)
    torch.cuda.synchronize()
    assert torch.equal(o_t, o_ref[row_of_tok]) and torch.equal(lse, lse_ref)
