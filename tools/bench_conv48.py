"""micro-benchmark of the specialised decoder1 conv kernels (forward/dgrad and wgrad) at 160^3 x 48, bf16"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from nerf_mae_amd import ops
from tests.test_kernels_gpu import _pack_via_kernel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = 160
dt = torch.bfloat16
x = torch.randn(B, R, R, R, 48, device='cuda').to(dt)
dy = torch.randn(B, R, R, R, 48, device='cuda').to(dt)
w = torch.randn(48, 48, 3, 3, 3) * (27 * 48) ** -0.5
wk = _pack_via_kernel(w, 6, dt, 41 * 3 * 64 * 8)
wk8 = wk.repeat(B).contiguous()
y = torch.empty_like(x)
dW = torch.zeros(48, 48, 3, 3, 3, device='cuda')
fl = 2.0 * 27 * 48 * 48 * R ** 3 * B
st = torch.empty(B, 48, 2, device='cuda'); sums = torch.zeros(B, 48, 2, dtype=torch.float64, device='cuda')
ops.instnorm_stats(x.view(-1, 48), st, ops.acc_zeros((B, 48, 2), 'cuda'), B, R ** 3, 48)
for name, fn in (("conv48 fwd", lambda: ops.conv3d_k3_c48(x, wk, out=y)), ("conv48 fwd + stats", lambda: ops.conv3d_k3_c48(x, wk, out=y, stats_acc=sums)),
                 ("stand-alone InstanceNorm apply", lambda: ops.instnorm_apply(x.view(-1, 48), st, dy.view(-1, 48), B, R ** 3, 48)),
                 ("conv48 fwd + stats, one weight image per sample (centered decoder1)", lambda: ops.conv3d_k3_c48_per_sample(x, wk8, out=y, stats_acc=sums)),
                 ("conv48 dgrad + IN-backward sums, centered (reads z)", lambda: ops.conv3d_k3_c48_bwd_reduce_centered(dy, wk, x, st, sums, out=y)),
                 ("conv48 wgrad, scaled reduce", lambda: ops.conv3d_k3_c48_wgrad_scaled(dy, x, st, dW)),
                 ("conv48 dgrad + IN-backward sums", lambda: ops.conv3d_k3_c48_bwd_reduce(dy, wk, x, st, sums, out=y)),
                 ("separate IN-backward reduce", lambda: ops.instnorm_bwd_reduce(y.view(-1, 48), None, x.view(-1, 48), st, sums, B, R ** 3, 48, rmode=0)),
                 ("conv48 wgrad", lambda: ops.conv3d_k3_c48_wgrad(dy, x, dW))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 20
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"{name}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / 2500 * 100:.1f}% of bf16 MFMA peak)")
