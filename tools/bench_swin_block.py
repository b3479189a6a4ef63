"""fused Swin-block kernels (csrc/swin_block.hip) against the unfused chains they replace, per stage shape (graph replay of 20 calls, best of 5).
usage: python tools/bench_swin_block.py [grids per step = 8] [widths, e.g. 384,192]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops


def bench(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


grids = int(sys.argv[1]) if len(sys.argv) > 1 else 8
widths = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [384, 192, 96]
dt = torch.bfloat16
SIDE = {96: 40, 192: 20, 384: 10}
for C in widths:
    s = SIDE[C]
    tps, heads = s ** 3, C // 32
    M = tps * grids
    dev = "cuda"
    x = torch.randn(M, C, device=dev).to(dt); dy = torch.randn(M, C, device=dev).to(dt)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    Wqkv = torch.randn(3 * C, C, device=dev) * C ** -0.5; bqkv = torch.zeros(3 * C, device=dev)
    Wp = torch.randn(C, C, device=dev) * C ** -0.5; bp = torch.zeros(C, device=dev)
    W1 = torch.randn(4 * C, C, device=dev) * C ** -0.5; b1 = torch.zeros(4 * C, device=dev)
    W2 = torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5; b2 = torch.zeros(C, device=dev)
    table = torch.randn(343, heads, device=dev) * 0.02
    rs = torch.ones(grids, device=dev)
    geom = ops.WinGeom(grids, s, s, s, [2, 2, 2])
    kinds = [ops.SWIN_ATTN_FWD, ops.SWIN_MLP_FWD]
    src = {ops.SWIN_ATTN_FWD: (Wqkv, Wp), ops.SWIN_MLP_FWD: (W1, W2)}
    st = {k: torch.empty(ops.swin_stream_numel(k, C), dtype=dt, device=dev) for k in kinds}
    arr = ops.swin_pack_items([(src[k][0], src[k][1], st[k], k, C) for k in kinds])
    res = {"pack": bench(lambda: ops.swin_pack(arr))}
    Wqkv_b, Wp_b, W1_b, W2_b = Wqkv.to(dt), Wp.to(dt), W1.to(dt), W2.to(dt)
    W2T_b, W1T_b, WqkvT_b, WpT_b = W2_b.T.contiguous(), W1_b.T.contiguous(), Wqkv_b.T.contiguous(), Wp_b.T.contiguous()
    xnw = torch.empty(geom.rows, C, dtype=dt, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    qkv = torch.empty(geom.rows, 3 * C, dtype=dt, device=dev); o = torch.empty(geom.rows, C, dtype=dt, device=dev); lse = torch.empty(geom.rows * heads, device=dev)
    x1 = torch.empty_like(x); xn = torch.empty_like(x); out = torch.empty_like(x)
    hpre = torch.empty(M, 4 * C, dtype=dt, device=dev); hact = torch.empty_like(hpre)

    def unf_attn_fwd():
        ops.layernorm_fwd(x, gam, bet, xnw, mean, rstd, geom.rows, C, src_mode=1, geom=geom)
        ops.gemm_nt(xnw, Wqkv_b, bias=bqkv, out=qkv)
        ops.window_attn_fwd(qkv, table, o, lse, heads, C, geom)
        ops.gemm_nt_window_scatter(o, Wp_b, x1, x, bp, rs, tps, geom)

    def unf_mlp_fwd():
        ops.layernorm_fwd(x, gam, bet, xn, mean, rstd, M, C)
        ops.gemm_nt(xn, W1_b, bias=b1, act=1, C2=hpre, out=hact)
        ops.gemm_nt(hact, W2_b, bias=b2, resid=x, rowscale=rs, rows_per_scale=tps, out=out)

    res["unfused attn fwd"] = bench(unf_attn_fwd)
    res["fused attn fwd"] = bench(lambda: ops.swin_attn_fwd(x, gam, bet, st[ops.SWIN_ATTN_FWD], bqkv, table, bp, geom, rowscale=rs, rows_per_scale=tps))
    res["unfused mlp fwd"] = bench(unf_mlp_fwd)
    res["fused mlp fwd"] = bench(lambda: ops.swin_mlp_fwd(x, gam, bet, st[ops.SWIN_MLP_FWD], b1, b2, rowscale=rs, rows_per_scale=tps))
    if ops.swin_mlp_split_ws(M, C, x.device) is not None:
        res["fused mlp fwd, one workgroup per tile"] = bench(lambda: ops.swin_mlp_fwd(x, gam, bet, st[ops.SWIN_MLP_FWD], b1, b2, rowscale=rs, rows_per_scale=tps, split=False))
        res["fused mlp fwd + hact"] = bench(lambda: ops.swin_mlp_fwd(x, gam, bet, st[ops.SWIN_MLP_FWD], b1, b2, rowscale=rs, rows_per_scale=tps, want_hact=True))
        res["fused mlp fwd + hact, one workgroup per tile"] = bench(lambda: ops.swin_mlp_fwd(x, gam, bet, st[ops.SWIN_MLP_FWD], b1, b2, rowscale=rs, rows_per_scale=tps, want_hact=True, split=False))
    if True:   # the unfused backward chains (the fused backward kernels of round 4 were removed in round 5)
        dh = torch.empty_like(hpre); dxn = torch.empty_like(x); dx1 = torch.empty_like(x); dx = torch.empty_like(x)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dyw = torch.empty(geom.rows, C, dtype=dt, device=dev); do = torch.empty_like(dyw); dqkv = torch.empty_like(qkv); dxnw = torch.empty_like(dyw)
        dtab = torch.zeros(343, heads, device=dev)
        unf_mlp_fwd(); unf_attn_fwd()

        def unf_mlp_bwd():
            ops.gemm_nt(dy, W2T_b, act=2, C2=hpre, rowscale=rs, rows_per_scale=tps, out=dh)
            ops.gemm_nt(dh, W1T_b, out=dxn)
            ops.layernorm_bwd(dxn, x, gam, mean, rstd, dx1, dg, db, M, C, dres=dy, geom=geom, tokens_per_sample=tps, dyw=dyw, dyw_scale=rs)

        def unf_attn_bwd():
            ops.gemm_nt(dyw, WpT_b, out=do)
            ops.window_attn_bwd(qkv, table, do, lse, dqkv, dtab, heads, C, geom)
            ops.gemm_nt(dqkv, WqkvT_b, out=dxnw)
            ops.layernorm_bwd(dxnw, x, gam, mean, rstd, dx, dg, db, M, C, src_mode=1, geom=geom, dres=dx1)

        res["unfused mlp bwd"] = bench(unf_mlp_bwd)
        res["unfused attn bwd"] = bench(unf_attn_bwd)

        res["unfused proj-dgrad + attn core bwd"] = bench(lambda: (ops.gemm_nt(dyw, WpT_b, out=do), ops.window_attn_bwd(qkv, table, do, lse, dqkv, dtab, heads, C, geom)))
        res["unfused qkv-dgrad + LN1 bwd"] = bench(lambda: (ops.gemm_nt(dqkv, WqkvT_b, out=dxnw), ops.layernorm_bwd(dxnw, x, gam, mean, rstd, dx, dg, db, M, C, src_mode=1, geom=geom, dres=dx1)))
    print(f"C={C} grids={grids} rows={M} window rows={geom.rows}: " + "  ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)
