"""the implicit-GEMM 3x3x3 convs of the 10^3 / 20^3 decoder levels at 8 grids (forward and input-gradient shapes) under the k-split knobs of csrc/gemm.hip"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16
def t(B, S, Cin, Cout):
    x = torch.randn(B, S, S, S, Cin, device='cuda').to(dt); w = (torch.randn(Cout, 27, Cin, device='cuda') * (27 * Cin) ** -0.5).to(dt)
    fn = lambda: ops.conv3d_k3(x, w, Cout)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    return ms * 1e3
tot = 0.0
out = []
for shp in [(8, 10, 768, 384), (8, 10, 384, 384), (8, 10, 384, 768), (8, 20, 384, 192), (8, 20, 192, 192), (8, 20, 192, 384)]:
    us = t(*shp); tot += us; out.append(f"{shp[1]}^3 {shp[2]}->{shp[3]}: {us:.0f}")
print(" | ".join(out), f"| sum {tot:.0f} us")
