"""micro-benchmark of the HBM-bound passes at decoder1 size"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
V, C = 160 ** 3, 48
dt = torch.bfloat16
x = torch.randn(B, V, C, device='cuda').to(dt); r = torch.randn_like(x); d = torch.randn_like(x); o = torch.empty_like(x); o2 = torch.empty_like(x)
stats = torch.empty(B, C, 2, device='cuda'); scr = torch.empty(B, C, 2, dtype=torch.float64, device='cuda'); sums = torch.empty_like(scr)
GB = x.numel() * 2 / 1e9
def t(name, fn, nbytes):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name:28s} {ms:7.3f} ms  {nbytes / ms:7.1f} GB/s")
t("in_stats (1 read)", lambda: ops.instnorm_stats(x, stats, scr, B, V, C), GB * 1e3)
t("in_apply rmode0 (1r 1w)", lambda: ops.instnorm_apply(x, stats, o, B, V, C), 2 * GB * 1e3)
t("in_apply rmode1 (2r 1w)", lambda: ops.instnorm_apply(x, stats, o, B, V, C, r=r, rmode=1), 3 * GB * 1e3)
t("in_bwd_reduce (3r)", lambda: ops.instnorm_bwd_reduce(d, o, x, stats, sums, B, V, C, rmode=1), 3 * GB * 1e3)
t("in_bwd_apply rmode1 (3r 2w)", lambda: ops.instnorm_bwd_apply(d, o, x, stats, sums, o2, B, V, C, rmode=1, dr=r), 5 * GB * 1e3)
t("torch copy (1r 1w)", lambda: o.copy_(x), 2 * GB * 1e3)
t("torch add (2r 1w)", lambda: torch.add(x, r, out=o), 3 * GB * 1e3)
