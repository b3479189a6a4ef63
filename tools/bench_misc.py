"""isolated timings of a few step kernels at the 8-grid shapes"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 512000
A = torch.randn(M, 96, device='cuda').to(dt); W = torch.randn(192, 96, device='cuda').to(dt); C = torch.randn(M, 192, device='cuda').to(dt)
print("gemm_nt M=512000 N=192 K=96 plain      %.0f us" % t(lambda: ops.gemm_nt(A, W, out=C)))
print("gemm_nt M=512000 N=192 K=96 accumulate %.0f us" % t(lambda: ops.gemm_nt(A, W, out=C, accumulate=True)))
W2 = torch.randn(96, 96, device='cuda').to(dt); C2 = torch.randn(M, 96, device='cuda').to(dt)
print("gemm_nt M=512000 N=96 K=96 accumulate  %.0f us" % t(lambda: ops.gemm_nt(A, W2, out=C2, accumulate=True)))
W3 = torch.randn(288, 96, device='cuda').to(dt); C3 = torch.randn(M, 288, device='cuda').to(dt)
print("gemm_nt M=512000 N=288 K=96 plain      %.0f us" % t(lambda: ops.gemm_nt(A, W3, out=C3)))
print("gemm_nt M=512000 N=288 K=96 accumulate %.0f us" % t(lambda: ops.gemm_nt(A, W3, out=C3, accumulate=True)))
