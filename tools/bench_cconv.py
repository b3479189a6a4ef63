"""composed ConvTranspose(k=s=4) o conv3x3x3 forward (csrc/cconv.hip) against the 48 -> 48 LDS-halo conv it replaces, 160^3, graph replay of 5 calls"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops

def bench(fn, n=5, reps=4):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best

for B in [int(v) for v in sys.argv[1:]] or [8, 1]:
    v = 40
    x = torch.randn(B, v, v, v, 96, device="cuda").to(torch.bfloat16)
    Wt = torch.randn(96, 48, 4, 4, 4, device="cuda") * 96 ** -0.5
    W1 = torch.randn(48, 48, 3, 3, 3, device="cuda") * (27 * 48) ** -0.5
    bt = torch.randn(48, device="cuda")
    Wcp = torch.empty(ops.cconv_pack_numel(), dtype=torch.bfloat16, device="cuda"); delta = torch.empty(27, 48, device="cuda")
    t_pack = bench(lambda: ops.cconv_pack(Wt, W1, bt, Wcp, delta))
    y = torch.empty(B, 160, 160, 160, 48, dtype=torch.bfloat16, device="cuda")
    acc = torch.zeros(B, 48, 2, dtype=torch.float64, device="cuda")
    t_cc = bench(lambda: ops.cconv_fwd(x, Wcp, delta, B, v, out=y, stats_acc=acc))
    u = torch.randn(B, 160, 160, 160, 48, device="cuda").to(torch.bfloat16)
    from tests.test_kernels_gpu import _pack_via_kernel
    wk = _pack_via_kernel(W1.cpu(), 6, torch.bfloat16, 41 * 3 * 64 * 8)
    t_c48 = bench(lambda: ops.conv3d_k3_c48(u, wk, out=y, stats_acc=acc))
    dy = torch.randn(B, 160, 160, 160, 48, device="cuda").to(torch.bfloat16)
    pws = torch.empty(ops.cconv_pack_ws_floats(), device="cuda")
    ops.cconv_pack(Wt, W1, bt, Wcp, delta, pws)
    dW = torch.zeros(48, 48, 3, 3, 3, device="cuda")
    t_cw = bench(lambda: ops.cconv_wgrad(x, dy, pws, bt, dW, B, v))
    t_w48 = bench(lambda: ops.conv3d_k3_c48_wgrad(dy, u, dW))
    print(f"B={B}: cconv_wgrad {t_cw:.3f} ms   conv48_wgrad {t_w48:.3f} ms", flush=True)
    Wdp = torch.empty(ops.cconv_dgrad_pack_numel(), dtype=torch.bfloat16, device="cuda")
    t_dp = bench(lambda: ops.cconv_dgrad_pack(Wcp, Wdp))
    dxb = torch.empty(B * v ** 3, 96, dtype=torch.bfloat16, device="cuda")
    t_dg = bench(lambda: ops.cconv_dgrad(dy.view(-1, 48), Wdp, B, v, out=dxb))
    wkd = _pack_via_kernel(W1.cpu(), 7, torch.bfloat16, 41 * 3 * 64 * 8)
    t_d48 = bench(lambda: ops.conv3d_k3_c48(dy, wkd, out=y))
    print(f"B={B}: cconv_dgrad {t_dg:.3f} ms ({2.0 * 216 * 96 * 48 * v ** 3 * B / t_dg / 1e9:.0f} TFLOP/s of composed work)   conv48 input gradient {t_d48:.3f} ms   dgrad pack {t_dp * 1e3:.0f} us", flush=True)
    Wup = torch.empty(ops.upconv4_pack_numel(), dtype=torch.bfloat16, device="cuda")
    ops.upconv4_pack(pws, Wup)
    cat = torch.empty(B * 160 ** 3, 48, dtype=torch.bfloat16, device="cuda")
    t_u4 = bench(lambda: ops.upconv4_fwd(x, Wup, bt, cat, B, v))
    wt_p = _pack_via_kernel(Wt.cpu(), 4, torch.bfloat16, Wt.numel())
    t_u = bench(lambda: ops.upconv_fwd(x.view(-1, 96), wt_p.view(64 * 48, 96), bt, cat, B, v, 4, 96, 48))
    print(f"B={B}: upconv4_fwd {t_u4:.3f} ms ({B * 160 ** 3 * 96 / t_u4 / 1e6:.0f} GB/s of output)   GEMM + pixel shuffle {t_u:.3f} ms", flush=True)
    fl = 2.0 * 216 * 96 * 48 * v ** 3 * B
    print(f"B={B}: cconv_fwd {t_cc:.3f} ms ({fl / t_cc / 1e9:.0f} TFLOP/s of composed work, {2.0 * 27 * 48 * 48 * 160 ** 3 * B / t_cc / 1e9:.0f} of the two-step FLOPs)   "
          f"conv48 {t_c48:.3f} ms   pack {t_pack * 1e3:.0f} us", flush=True)
