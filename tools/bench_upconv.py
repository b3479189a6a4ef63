"""the k = s = 4 transpose conv of decoder1 (96 -> 48, 40^3 -> 160^3, 4 grids): forward, input gradient, weight gradient"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_mae_amd import ops
B, v, k, Cin, Cout = 4, 40, 4, 96, 48
dt = torch.bfloat16
x = torch.randn(B * v ** 3, Cin, device='cuda').to(dt)
Wt = (torch.randn(k ** 3 * Cout, Cin, device='cuda') * Cin ** -0.5).to(dt)
Wd = Wt.t().contiguous()
bias = torch.randn(Cout, device='cuda')
cat = torch.empty(B * (v * k) ** 3, Cout, dtype=dt, device='cuda')
dcat = torch.randn_like(cat)
dx = torch.empty_like(x)
dW, db = torch.zeros(Cin, Cout, k, k, k, device='cuda'), torch.zeros(Cout, device='cuda')
def t(name, fn, nbytes):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name:14s}: {ms * 1e3:8.1f} us   {nbytes / ms / 1e6:7.1f} GB/s (HBM-minimal bytes)")
nb = cat.numel() * 2 + x.numel() * 2
t("upconv fwd", lambda: ops.upconv_fwd(x, Wt, bias, cat, B, v, k, Cin, Cout), nb)
t("upconv dgrad", lambda: ops.upconv_dgrad(dcat, Wd, dx, B, v, k, Cin, Cout), nb)
t("upconv wgrad", lambda: ops.upconv_wgrad(dcat, x, dW, db, B, v, k, Cin, Cout), nb)
