"""generic conv3d wgrad at the decoder shapes"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16
def t(B, S, Cin, Cout):
    X = torch.randn(B, S, S, S, Cin, device='cuda').to(dt); dY = torch.randn(B, S, S, S, Cout, device='cuda').to(dt)
    dW = torch.zeros(Cout, Cin, 3, 3, 3, device='cuda')
    fn = lambda: ops.conv3d_k3_wgrad(dY, X, dW)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"conv wgrad B={B} {S}^3 {Cin}->{Cout}: {ms*1e3:8.1f} us  {2.0*27*Cin*Cout*B*S**3/ms/1e9:7.1f} TF/s")
for a in ((4, 40, 192, 96), (4, 40, 96, 96), (4, 20, 384, 192), (4, 20, 192, 192), (4, 10, 768, 384), (4, 10, 384, 384)):
    t(*a)
