"""micro-benchmark of the 64-channel-block LDS-halo conv kernels vs the generic implicit-GEMM path"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
from tests.test_kernels_gpu import _pack_via_kernel
dt = torch.bfloat16
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B, S, Cin, Cout in ((int(sys.argv[1]) if len(sys.argv) > 1 else 4, 160, 64, 64), (4, 40, 256, 256), (4, 40, 256, 128), (4, 20, 512, 256)):
    x = torch.randn(B, S, S, S, Cin, device='cuda').to(dt)
    dy = torch.randn(B, S, S, S, Cout, device='cuda').to(dt)
    w = torch.randn(Cout, Cin, 3, 3, 3) * (27 * Cin) ** -0.5
    wk = _pack_via_kernel(w, 8, dt, ops.conv64_pack_numel(Cin, Cout))
    wg = _pack_via_kernel(w, 2, dt, w.numel())
    fl = 2.0 * 27 * Cin * Cout * S ** 3 * B
    y = torch.empty(B, S, S, S, Cout, device='cuda', dtype=dt)
    m1 = t(lambda: ops.conv3d_k3_c64(x, wk, Cout, out=y))
    m2 = t(lambda: ops.conv3d_k3(x, wg, Cout, out=y))
    dW = torch.zeros(Cout, Cin, 3, 3, 3, device='cuda')
    m3 = t(lambda: ops.conv3d_k3_wgrad(dy, x, dW))
    print(f"B={B} {S}^3 {Cin}->{Cout}: halo64 {m1:.3f} ms = {fl / m1 / 1e9:.0f} TF/s | generic fwd {m2:.3f} ms = {fl / m2 / 1e9:.0f} TF/s | wgrad (dispatched) {m3:.3f} ms = {fl / m3 / 1e9:.0f} TF/s")
