"""what bounds the wide-output short-K GEMMs: plain (no bias / activation) and tiny-K variants of 512000 x 384"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16
def t(name, M, N, K, bias=False, act=0):
    A = torch.randn(M, K, device='cuda').to(dt); W = (torch.randn(N, K, device='cuda') * K ** -0.5).to(dt)
    out = torch.empty(M, N, dtype=dt, device='cuda'); b = torch.randn(N, device='cuda') if bias else None
    kw = dict(act=1, C2=torch.empty_like(out)) if act else {}
    fn = lambda: ops.gemm_nt(A, W, bias=b, out=out, **kw)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 20
    nb = (M * K + M * N * (2 if act else 1)) * 2
    print(f"{name:28s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {nb/ms/1e6:7.1f} GB/s")
t("plain", 512000, 384, 96)
t("bias", 512000, 384, 96, bias=True)
t("bias+gelu dual", 512000, 384, 96, bias=True, act=1)
t("plain K=32", 512000, 384, 32)
t("plain N=768", 512000, 768, 96)
t("plain N=96", 512000, 96, 96)
