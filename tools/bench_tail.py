"""Micro-benchmark of the decoder-tail backward kernels at the bench shape (B grids of 160^3 x 48)."""
import sys
import time
import torch
sys.path.insert(0, ".")
from nerf_mae_amd import ops

B, R, Cd = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 160, 48
V = R ** 3
dev = "cuda"
dt = torch.bfloat16
d0 = torch.randn(B * V, Cd, device=dev, dtype=dt)
y = torch.randn(B * V, Cd, device=dev, dtype=dt)
x = torch.rand(B, 4, R, R, R, device=dev)
ext = torch.tensor([[R, R, R]] * B, dtype=torch.int32, device=dev)
tm = (torch.rand(40 ** 3, device=dev) < 0.75).to(torch.uint8)
Wo, bo = torch.randn(4, Cd, device=dev) * 0.1, torch.zeros(4, device=dev)
stats = torch.empty(B, Cd, 2, device=dev)
scr = torch.empty(B, Cd, 2, dtype=torch.float64, device=dev)
ops.instnorm_stats(y, stats, scr, B, V, Cd)
lsums, losses = torch.empty(8, dtype=torch.float64, device=dev), torch.empty(3, device=dev)
dpred = torch.empty(B * V, 4, device=dev)
ops.mae_loss_fwd(d0, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred)
dy, dr, dd0 = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
dW, db = torch.zeros(4, Cd, device=dev), torch.zeros(4, device=dev)
sums = torch.empty(B, Cd, 2, dtype=torch.float64, device=dev)


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


r = torch.randn_like(y)


def fused():
    ops.mae_tail_bwd(None, y, stats, dpred, lsums, Wo, sums, dy, dr, dW, db, B, V, Cd, r=r)


def fused_stored():
    ops.mae_tail_bwd(d0, y, stats, dpred, lsums, Wo, sums, dy, dr, dW, db, B, V, Cd)


def unfused():
    ops.mae_loss_bwd(d0, Wo, bo, x, ext, tm, B, R, Cd, lsums, dd0, torch.empty(1, 8, device=dev, dtype=dt), dW, db)
    ops.instnorm_bwd_reduce(dd0, d0, y, stats, sums, B, V, Cd, r=d0, rmode=1)
    ops.instnorm_bwd_apply(dd0, d0, y, stats, sums, dy, B, V, Cd, r=d0, rmode=1, dr=dr)


print(f"fused tail bwd  {timeit(fused):.3f} ms (stored d0: {timeit(fused_stored):.3f})   unfused {timeit(unfused):.3f} ms   loss fwd {timeit(lambda: ops.mae_loss_fwd(d0, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None)):.3f} / with dpred {timeit(lambda: ops.mae_loss_fwd(d0, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred)):.3f} ms")
out = torch.empty_like(y)
t_f0 = timeit(lambda: ops.mae_tail_fwd(y, stats, r, None, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred))
t_f = timeit(lambda: ops.mae_tail_fwd(y, stats, r, out, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred))
t_u = timeit(lambda: (ops.instnorm_apply(y, stats, out, B, V, Cd, r=r, rmode=1), ops.mae_loss_fwd(out, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred)))
print(f"fused tail fwd {t_f:.3f} ms (d0 not stored: {t_f0:.3f})   unfused (in_apply + loss fwd) {t_u:.3f} ms")
bsum = torch.empty(B * Cd * 4 + 4 * Cd, dtype=torch.float64, device=dev)
t_m = timeit(lambda: ops.mae_tail_fwd(y, stats, r, None, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred, bwd_sums=bsum))
print(f"fused tail fwd + backward sums (the step's launch; bf16 C=48: matrix-core kernel) {t_m:.3f} ms")
smask = torch.empty(B * V, 8, dtype=torch.uint8, device=dev)
ops.mae_tail_fwd(y, stats, r, None, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, None, dpred, bwd_sums=bsum, sign_mask=smask)
t_b = timeit(lambda: ops.mae_tail_bwd(None, y, stats, dpred, lsums, Wo, sums, dy, dr, dW, db, B, V, Cd, bwd_sums=bsum, sign_mask=smask))
gb = B * V * (3 * Cd * 2 + 8 + 16) / 1e9
print(f"tail bwd apply pass with the forward's sums + sign mask (the step's launch) {t_b:.3f} ms ({gb / t_b:.2f} TB/s)")
