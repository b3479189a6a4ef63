"""window attention fwd/bwd at the encoder stage shapes inside a replayed graph"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from nerf_mae_amd import ops
dt = torch.bfloat16
def run(B, S, C, heads, shift):
    geom = ops.WinGeom(B, S, S, S, [shift] * 3)
    rows = geom.rows
    qkv = torch.randn(rows, 3 * C, device='cuda').to(dt)
    out = torch.empty(rows, C, device='cuda', dtype=dt)
    lse = torch.empty(rows * heads, device='cuda')
    table = torch.randn(343, heads, device='cuda') * 0.02
    do = torch.randn(rows, C, device='cuda').to(dt)
    dqkv = torch.empty_like(qkv)
    dtab = torch.zeros(343, heads, device='cuda')
    N = 20
    def f():
        for _ in range(N): ops.window_attn_fwd(qkv, table, out, lse, heads, C, geom)
    def b():
        for _ in range(N): ops.window_attn_bwd(qkv, table, do, lse, dqkv, dtab, heads, C, geom)
    res = []
    for fn in (f, b):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): fn()
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t) / 5 / N * 1e6)
    print(f"attn B={B} {S}^3 C={C} heads={heads} shift={shift}: fwd {res[0]:.1f} us  bwd {res[1]:.1f} us")
run(1, 10, 384, 12, 2); run(2, 10, 384, 12, 2); run(1, 5, 768, 24, 0); run(2, 5, 768, 24, 0); run(8, 40, 96, 3, 2); run(8, 20, 192, 6, 2); run(8, 10, 384, 12, 2); run(4, 10, 384, 12, 2); run(4, 10, 384, 12, 0); run(4, 40, 96, 3, 2); run(4, 20, 192, 6, 0); run(4, 5, 768, 24, 0)
