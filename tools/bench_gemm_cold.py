"""the stage-2 Linears at 8 grids per GPU with the operands rotating through NSET distinct buffer sets (what a training step sees: every
launch reads activations / weights and writes outputs that were not touched by the previous launches) against one hot set"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16


def t(name, M, N, K, nset, act=0):
    sets = []
    for _ in range(nset):
        A = torch.randn(M, K, device='cuda').to(dt); W = (torch.randn(N, K, device='cuda') * K ** -0.5).to(dt)
        out = torch.empty(M, N, dtype=dt, device='cuda'); bias = torch.randn(N, device='cuda')
        extra = dict(act=1, C2=torch.empty_like(out)) if act else {}
        sets.append((A, W, bias, out, extra))
    def body():
        for A, W, bias, out, extra in sets: ops.gemm_nt(A, W, bias=bias, out=out, **extra)
    reps = max(1, 40 // nset)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): body()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); b.record(); torch.cuda.synchronize()
    print(f"{name:22s} M={M} N={N} K={K} sets={nset:3d}: {a.elapsed_time(b) / (2 * reps * nset) * 1e3:7.1f} us")


for nset in (1, 40):
    t("fc1 s2 (gelu dual)", 8000, 1536, 384, nset, act=1)
    t("fc2 s2", 8000, 384, 1536, nset)
    t("qkv s2", 13824, 1152, 384, nset)
    t("proj s2", 13824, 384, 384, nset)
