import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16
def t(name, M, N, K, **kw):
    A = torch.randn(M, K, device='cuda').to(dt); W = (torch.randn(N, K, device='cuda') * K ** -0.5).to(dt)
    out = torch.empty(M, N, dtype=dt, device='cuda'); bias = torch.randn(N, device='cuda')
    extra = {}
    if kw.get('act') == 1: extra = dict(act=1, C2=torch.empty_like(out))
    fn = lambda: ops.gemm_nt(A, W, bias=bias, out=out, **extra)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    nbytes = (M * K + N * K + M * N * (2 if extra else 1)) * 2
    print(f"{name:34s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s  {nbytes/ms/1e6:7.1f} GB/s")
t("convT dec1", 256000, 3072, 96)
t("qkv stage0", 256000, 288, 96)
t("fc1 stage0 (gelu dual)", 256000, 384, 96, act=1)
t("fc2 stage0", 256000, 96, 384)
t("proj stage0", 256000, 96, 96)
t("fc1 stage1", 32000, 768, 192, act=1)
t("qkv stage2", 6912, 1152, 384)
t("fc1 stage2", 4000, 1536, 384, act=1)
t("fc2 stage2", 4000, 384, 1536)
t("proj stage2", 6912, 384, 384)
t("qkv dgrad stage2", 6912, 384, 1152)
t("fc1 stage3", 500, 3072, 768, act=1)
t("fc2 stage3", 500, 768, 3072)
t("fc2 stage1", 32000, 192, 768)
t("convT dgrad dec1", 256000, 96, 3072)
