"""loops one of the conv48 kernels for N seconds (clock / power sampling under load: tools/pmc_conv48.sh)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_mae_amd import ops
from tests.test_kernels_gpu import _pack_via_kernel
kind, B, secs = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
x = torch.randn(B, 160, 160, 160, 48, device='cuda').to(torch.bfloat16)
dy = torch.randn(B, 160, 160, 160, 48, device='cuda').to(torch.bfloat16)
wk = _pack_via_kernel(torch.randn(48, 48, 3, 3, 3) * (27 * 48) ** -0.5, 6, torch.bfloat16, 41 * 3 * 64 * 8)
y = torch.empty_like(x)
dW = torch.zeros(48, 48, 3, 3, 3, device='cuda')
fn = (lambda: ops.conv3d_k3_c48(x, wk, out=y)) if kind == "fwd" else (lambda: ops.conv3d_k3_c48_wgrad(dy, x, dW))
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(50): fn()
    torch.cuda.synchronize()
