"""fused MLP branch (csrc/mlp_fused.hip) against the unfused chain it replaces, per stage shape, forward and backward (graph replay of 20 calls)"""
import os, sys, time
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
dt = torch.bfloat16
for C, tps in ((96, 64000), (192, 8000), (384, 1000)):
    M = tps * grids
    x = torch.randn(M, C, device="cuda").to(dt); dy = torch.randn(M, C, device="cuda").to(dt)
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    W1 = (torch.randn(4 * C, C, device="cuda") * C ** -0.5).to(dt); b1 = torch.zeros(4 * C, device="cuda")
    W2 = (torch.randn(C, 4 * C, device="cuda") * (4 * C) ** -0.5).to(dt); b2 = torch.zeros(C, device="cuda")
    W2T, W1T = W2.T.contiguous(), W1.T.contiguous()
    rs = torch.ones(grids, device="cuda")
    out = torch.empty_like(x); xn = torch.empty_like(x); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    hpre = torch.empty(M, 4 * C, dtype=dt, device="cuda"); hact = torch.empty_like(hpre); dh = torch.empty_like(hpre); dxn = torch.empty_like(x); dx = torch.empty_like(x)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    geom = ops.WinGeom(grids, *{64000: (40, 40, 40), 8000: (20, 20, 20), 1000: (10, 10, 10)}[tps], [2, 2, 2])
    dyw = torch.empty(geom.rows, C, dtype=dt, device="cuda")

    def unf_fwd():
        ops.layernorm_fwd(x, gam, bet, xn, mean, rstd, M, C)
        ops.gemm_nt(xn, W1, bias=b1, act=1, C2=hpre, out=hact)
        ops.gemm_nt(hact, W2, bias=b2, resid=x, rowscale=rs, rows_per_scale=tps, out=out)

    def unf_bwd():
        ops.gemm_nt(dy, W2T, act=2, C2=hpre, rowscale=rs, rows_per_scale=tps, out=dh)
        ops.gemm_nt(dh, W1T, out=dxn)
        ops.layernorm_bwd(dxn, x, gam, mean, rstd, dx, dg, db, M, C, dres=dy, geom=geom, tokens_per_sample=tps, dyw=dyw, dyw_scale=rs)

    res = {"unfused fwd": bench(unf_fwd), "unfused bwd": bench(unf_bwd)}
    for mt in ((1, 2, 4) if C == 96 else (1, 2)):
        os.environ["NMH_MLP_FWD_MT"] = str(mt)
        res[f"fused fwd MT{mt}"] = bench(lambda: ops.mlp_fused_fwd(x, gam, bet, W1, b1, W2T, b2, rowscale=rs, rows_per_scale=tps, out=out))
    os.environ.pop("NMH_MLP_FWD_MT")
    for mt in (1, 2):
        os.environ["NMH_MLP_BWD_MT"] = str(mt)
        try:
            res[f"fused bwd MT{mt}"] = bench(lambda: ops.mlp_fused_bwd(x, dy, gam, bet, W1, b1, W2T, dg, db, rowscale=rs, rows_per_scale=tps, dyw=dyw, dyw_scale=rs, geom=geom))
        except Exception as e:
            res[f"fused bwd MT{mt}"] = float("nan")
    os.environ.pop("NMH_MLP_BWD_MT")
    if C == 96:
        for mt in (1, 2):
            os.environ["NMH_MLP96_FWD_MT"] = str(mt); os.environ["NMH_MLP96_BWD_MT"] = str(mt)
            res[f"p96 fwd MT{mt}"] = bench(lambda: ops.mlp_fused_fwd(x, gam, bet, W1, b1, W2T, b2, rowscale=rs, rows_per_scale=tps, out=out))
            res[f"p96 bwd MT{mt}"] = bench(lambda: ops.mlp_fused_bwd(x, dy, gam, bet, W1, b1, W2T, dg, db, rowscale=rs, rows_per_scale=tps, dyw=dyw, dyw_scale=rs, geom=geom))
    print(f"C={C} M={M}: " + "  ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)
