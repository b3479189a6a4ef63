"""the persistent kernel of the composed conv1 weight gradient alone (csrc/cconv.hip: cconv_wgrad_dma_kernel, NMH_CCW_DMA=0: the register-staged
cconv_wgrad_kernel), 160^3, graph replay of 5 calls; the two builds of the stage are compared across processes (the switch is read once)"""
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


if __name__ == "__main__":
    for B in [int(v) for v in sys.argv[1:]] or [8, 4, 1]:
        v = 40
        x = torch.randn(B, v, v, v, 96, device="cuda").to(torch.bfloat16)
        Wt = torch.randn(96, 48, 4, 4, 4, device="cuda") * 96 ** -0.5
        W1 = torch.randn(48, 48, 3, 3, 3, device="cuda") * (27 * 48) ** -0.5
        bt = torch.randn(48, device="cuda")
        Wcp = torch.empty(ops.cconv_pack_numel(), dtype=torch.bfloat16, device="cuda"); delta = torch.empty(27, 48, device="cuda")
        pws = torch.empty(ops.cconv_pack_ws_floats(), device="cuda")
        ops.cconv_pack(Wt, W1, bt, Wcp, delta, pws)
        dy = torch.randn(B, 160, 160, 160, 48, device="cuda").to(torch.bfloat16)
        dW = torch.zeros(48, 48, 3, 3, 3, device="cuda")
        t1 = bench(lambda: ops.cconv_wgrad(x, dy, pws, bt, dW, B, v, phase=1))
        t0 = bench(lambda: ops.cconv_wgrad(x, dy, pws, bt, dW, B, v))
        print(f"NMH_CCW_DMA={os.environ.get('NMH_CCW_DMA', '1')} B={B}: G-block kernel {t1:.3f} ms ({2.0 * 216 * 96 * 48 * v ** 3 * B / t1 / 1e9:.0f} TFLOP/s executed, "
              f"{B * 160 ** 3 * 96 / t1 / 1e6:.0f} GB/s of dy)   whole entry {t0:.3f} ms", flush=True)
