"""fused Swin-block forward kernels with the weight stream hot in L2 (one stream replayed), L2-cold (a rotation of 18 streams = one stage-2 pass) and
MALL-cold (a rotation whose total exceeds the 256-MB infinity cache): what a step pays for streaming each block's weights for the first time.
usage: python tools/bench_swin_cold.py [grids per step = 8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops


def bench(fns, reps=5):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fns[0]()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for f in fns:
                f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / len(fns) * 1e3)
    return best


grids = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt, dev = torch.bfloat16, "cuda"
for C, side in ((384, 10), (192, 20)):
    tps, heads = side ** 3, C // 32
    M = tps * grids
    x = torch.randn(M, C, device=dev).to(dt)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    bqkv, bp, b1, b2 = torch.zeros(3 * C, device=dev), torch.zeros(C, device=dev), torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev)
    table = torch.randn(343, heads, device=dev) * 0.02
    rs = torch.ones(grids, device=dev)
    geom = ops.WinGeom(grids, side, side, side, [2, 2, 2])
    na, nm = ops.swin_stream_numel(ops.SWIN_ATTN_FWD, C), ops.swin_stream_numel(ops.SWIN_MLP_FWD, C)
    for nrot in (1, 18, 160):
        sa = [(torch.randn(na, device=dev) * 0.02).to(dt) for _ in range(nrot)]
        sm = [(torch.randn(nm, device=dev) * 0.02).to(dt) for _ in range(nrot)]
        n = max(nrot, 20)
        ta = bench([(lambda i=i: ops.swin_attn_fwd(x, gam, bet, sa[i % nrot], bqkv, table, bp, geom, rowscale=rs, rows_per_scale=tps)) for i in range(n)])
        tm = bench([(lambda i=i: ops.swin_mlp_fwd(x, gam, bet, sm[i % nrot], b1, b2, rowscale=rs, rows_per_scale=tps)) for i in range(n)])
        both = bench([(lambda i=i: (ops.swin_attn_fwd(x, gam, bet, sa[i % nrot], bqkv, table, bp, geom, rowscale=rs, rows_per_scale=tps),
                                    ops.swin_mlp_fwd(x, gam, bet, sm[i % nrot], b1, b2, rowscale=rs, rows_per_scale=tps))) for i in range(n)])
        print(f"C={C} grids={grids} streams in rotation {nrot:3d} ({nrot * (na + nm) * 2 / 1e6:6.1f} MB): attn fwd {ta:6.1f} us  mlp fwd {tm:6.1f} us  block (attn + mlp) {both:6.1f} us", flush=True)
