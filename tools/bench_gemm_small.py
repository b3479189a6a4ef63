"""small-M Linear shapes of the 1-2 grid step (stage 3/4 of the encoder): graph-replayed launch time of nmh_gemm_nt per shape.
python tools/bench_gemm_small.py   (A/B with NMH_GEMM_DMA_DEEP=0/1)"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16
shapes = [(1000, 384, 1536), (1000, 1536, 384), (1728, 384, 384), (1728, 1152, 384), (2000, 384, 1536), (4000, 384, 1536),
          (125, 768, 3072), (125, 3072, 768), (216, 768, 768), (512, 768, 1536)]
for M, N, K in shapes:
    A = torch.randn(M, K, device='cuda').to(dt); W = (torch.randn(N, K, device='cuda') * K ** -0.5).to(dt)
    b = torch.randn(N, device='cuda'); out = torch.empty(M, N, dtype=dt, device='cuda')
    for _ in range(3): ops.gemm_nt(A, W, bias=b, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): ops.gemm_nt(A, W, bias=b, out=out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"M={M:5d} N={N:5d} K={K:5d}: {e0.elapsed_time(e1) / 50 * 1e3:7.2f} us/launch")
