"""NMH_ATTN_DBG=1: per-phase s_memtime totals of wave 0 of every workgroup of attn_bwd4_kernel (diagnostic build)"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from nerf_mae_amd import ops
dt = torch.bfloat16
names = ["first request", "LDS write of the rows (incl. wait for the loads)", "barrier 1", "phase A", "barrier 2", "phase B (+ next request)", "barrier 3"]
for B, S, C, heads, shift in ((8, 10, 384, 12, 2), (8, 40, 96, 3, 2), (1, 10, 384, 12, 2)):
    geom = ops.WinGeom(B, S, S, S, [shift] * 3)
    rows = geom.rows
    qkv = torch.randn(rows, 3 * C, device='cuda').to(dt)
    lse = torch.randn(rows * heads, device='cuda')
    table = torch.randn(343, heads, device='cuda') * 0.02
    do = torch.randn(rows, C, device='cuda').to(dt)
    dqkv = torch.empty_like(qkv)
    dtab = torch.zeros(343, heads, device='cuda')
    for _ in range(3):
        ops.window_attn_bwd(qkv, table, do, lse, dqkv, dtab, heads, C, geom)
    torch.cuda.synchronize()
    dtab.zero_()
    ops.window_attn_bwd(qkv, table, do, lse, dqkv, dtab, heads, C, geom)
    torch.cuda.synchronize()
    v = dtab.flatten()[:8].double().cpu()
    nblk = v[7].item()
    nwin = rows // 64
    print(f"B={B} {S}^3 C={C} heads={heads}: {int(nblk)} workgroups, {nwin * heads / nblk:.2f} windows each; s_memtime ticks per workgroup (100 MHz ticks = 10 ns):")
    for n, x in zip(names, v[:7]):
        print(f"   {n:55s} {x.item() / nblk:10.1f}")
    print(f"   total {v[:7].sum().item() / nblk:10.1f}")
