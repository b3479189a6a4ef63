import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
dt = torch.bfloat16
def t(name, M, N, K):
    A = torch.randn(M, N, device='cuda').to(dt); B = torch.randn(M, K, device='cuda').to(dt); dW = torch.zeros(N, K, device='cuda')
    fn = lambda: ops.gemm_tn(A, B, dW)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{name:22s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s  {(M*N+M*K)*2/ms/1e6:7.1f} GB/s(min traffic)")
t("qkv wgrad s0", 256000, 288, 96)
t("fc1 wgrad s0", 256000, 384, 96)
t("fc2 wgrad s0", 256000, 96, 384)
t("proj wgrad s0", 256000, 96, 96)
t("convT wgrad dec1", 256000, 3072, 96)
t("fc1 wgrad s1", 32000, 768, 192)
t("qkv wgrad s2", 6912, 1152, 384)
t("fc1 wgrad s2", 4000, 1536, 384)
t("fc2 wgrad s2", 4000, 384, 1536)
t("proj wgrad s2", 6912, 384, 384)
t("fc1 wgrad s3", 500, 3072, 768)
