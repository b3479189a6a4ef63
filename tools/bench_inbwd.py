"""InstanceNorm backward passes at the bench shape, with and without re-reading the activation."""
import sys
import time
import torch
sys.path.insert(0, ".")
from nerf_mae_amd import ops

B, R, C = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 160, 48
V = R ** 3
dt = torch.bfloat16
x = torch.randn(B * V, C, device="cuda", dtype=dt)
dout = torch.randn(B * V, C, device="cuda", dtype=dt)
out = torch.empty_like(x)
dx = torch.empty_like(x)
stats = torch.empty(B, C, 2, device="cuda")
scr = torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
sums = torch.empty(B, C, 2, dtype=torch.float64, device="cuda")
ops.instnorm_stats(x, stats, scr, B, V, C)
ops.instnorm_apply(x, stats, out, B, V, C)


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


gb = B * V * C * 2 / 1e9
for o in (out, None):
    t1 = timeit(lambda: ops.instnorm_bwd_reduce(dout, o, x, stats, sums, B, V, C))
    t2 = timeit(lambda: ops.instnorm_bwd_apply(dout, o, x, stats, sums, dx, B, V, C))
    n = 3 if o is not None else 2
    print(f"out={'yes' if o is not None else 'no '}  reduce {t1:.3f} ms ({n * gb / t1:.2f} TB/s)   apply {t2:.3f} ms ({(n + 1) * gb / t2:.2f} TB/s)")
t = timeit(lambda: ops.instnorm_apply(x, stats, out, B, V, C))
print(f"fwd apply {t:.3f} ms ({2 * gb / t:.2f} TB/s)")
