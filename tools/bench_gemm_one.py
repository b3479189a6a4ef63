"""one nmh_gemm_nt shape for counter collection: python tools/bench_gemm_one.py M N K [bias] [act]"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
bias, act = (len(sys.argv) > 4 and sys.argv[4] == '1'), (len(sys.argv) > 5 and sys.argv[5] == '1')
dt = torch.bfloat16
A = torch.randn(M, K, device='cuda').to(dt); W = (torch.randn(N, K, device='cuda') * K ** -0.5).to(dt)
out = torch.empty(M, N, dtype=dt, device='cuda'); b = torch.randn(N, device='cuda') if bias else None
kw = dict(act=1, C2=torch.empty_like(out)) if act else {}
for _ in range(4): ops.gemm_nt(A, W, bias=b, out=out, **kw)
torch.cuda.synchronize()
