"""per-phase cycle accounting of conv48_wgrad_kernel (NMH_W48_DBG=1: s_memtime stamps written into the partial-sum workspace)"""
import os, sys
os.environ["NMH_W48_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops
B, R, dt = 4, 160, torch.bfloat16
x = torch.randn(B, R, R, R, 48, device='cuda').to(dt)
dy = torch.randn(B, R, R, R, 48, device='cuda').to(dt)
dW = torch.zeros(48, 48, 3, 3, 3, device='cuda')
for _ in range(3): ops.conv3d_k3_c48_wgrad(dy, x, dW)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): ops.conv3d_k3_c48_wgrad(dy, x, dW)
b.record(); torch.cuda.synchronize()
print(f"wgrad (instrumented): {a.elapsed_time(b) / 10:.4f} ms")
ws = ops._C48_WS[x.device.index]
P = 81 * 3 * 256
c = ws[256 * P:256 * P + 256 * 8 * 8 * 2].view(torch.int64).view(256, 8, 8).double()
tiles = 250.0
names = ["origin + loop top", "k-loop (8 steps)", "barrier 1", "halo -> LDS", "barrier 2"]
m = c.mean(dim=(0, 1)) / tiles
for i, n in enumerate(names):
    print(f"   {n:>18}: {m[i].item():9.0f} cycles/tile   (min wave {c[:, :, i].min().item() / tiles:9.0f}, max wave {c[:, :, i].max().item() / tiles:9.0f})")
print(f"   {'TOTAL':>18}: {m[:5].sum().item():9.0f} cycles/tile;  MFMA floor 8 x 31 x 16 x 2 waves/SIMD = 7936")
