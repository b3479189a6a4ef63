"""tile-shape sweep (NMH_GEMM_CFG override, read per dispatch) of the stage-2 MLP GEMMs: fc1 forward (bias + GELU + pre-activation copy) and the
fc2 input gradient (GELU' of the stored pre-activation), M rows x 1536 x 384; graph replay over rotating buffers (cold-ish caches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
N, K, NB = 1536, 384, 6
dt = torch.bfloat16
A = [torch.randn(M, K, device="cuda").to(dt) for _ in range(NB)]
W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
b = torch.randn(N, device="cuda")
out = [torch.empty(M, N, dtype=dt, device="cuda") for _ in range(NB)]
pre = [torch.empty(M, N, dtype=dt, device="cuda") for _ in range(NB)]

def run(kind):
    for i in range(NB):
        if kind == "fwd":
            ops.gemm_nt(A[i], W, bias=b, out=out[i], act=1, C2=pre[i])
        else:
            ops.gemm_nt(A[i], W, out=out[i], act=2, C2=pre[i])

def bench(kind):
    run(kind); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(kind); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(3): run(kind)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (3 * NB))
    return best * 1e3

for cfg in [None, "1,6", "1,8", "2,6", "2,8", "4,6", "4,8", "dma3", "dma4"]:
    os.environ.pop("NMH_GEMM_CFG", None)
    if cfg and not cfg.startswith("dma"):
        os.environ["NMH_GEMM_CFG"] = cfg
    if cfg and cfg.startswith("dma"):
        continue      # (NMH_GEMM_DMA is read once per process: run the script again with NMH_GEMM_DMA=3/4 for those)
    print(f"M={M} cfg={cfg or 'default':8s} fc1 fwd {bench('fwd'):7.1f} us   fc2 dgrad {bench('bwd'):7.1f} us", flush=True)
