"""streaming weight-gradient kernel (csrc/tn_grouped.hip: gemm_tn_stream_kernel) against the tile kernels (NMH_TNS=0) on the long-contraction problem sets of a
training step at 8 grids: one stage-0 block, one stage-1 block (+ the patch-merging reduction), the decoder-1 transpose-conv gradient.  Isolated, eager, 10 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops
from nerf_mae_amd._lib import lib
dev = torch.device("cuda")
dt = torch.bfloat16
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def block_set(T, C, rps, extra=()):
    probs = []
    for N, K, rs in ((3 * C, C, False), (C, C, True), (4 * C, C, False), (C, 4 * C, True)) + tuple(extra):
        a = torch.randn(T, N, device=dev).to(dt); b = torch.randn(T, K, device=dev).to(dt)
        w = torch.zeros(N, K, device=dev); bias = torch.zeros(N, device=dev)
        r = (torch.rand(T // rps, device=dev) + 0.5) if rs else None
        probs.append((a, b, w, bias, r, rps))
    return probs


def run_set(probs):
    arr = (ops._TnProblem * len(probs))()
    for i, (a, b, w, bias, r, rps) in enumerate(probs):
        arr[i] = ops._TnProblem(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), w.data_ptr(), b.shape[1], bias.data_ptr(), 0 if r is None else r.data_ptr(),
                                a.shape[0], a.shape[1], b.shape[1], rps)
    ws = ops._tn_workspace(dev)
    return lambda: lib().call("nmh_gemm_tn_grouped", ops.BF16, arr, len(probs), ws, ws.numel(), ops._st())


def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def compare(name, probs, fn, nbytes):
    res = {}
    for mode in ("0", "1"):
        os.environ["NMH_TNS"] = mode
        for p in probs:
            p[2].zero_(); p[3].zero_()
        fn(); torch.cuda.synchronize()
        res[mode] = [(p[2].clone(), p[3].clone()) for p in probs]
        us = timeit(fn)
        print(f"{name:28s} NMH_TNS={mode}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s of operand bytes", flush=True)
    for i, ((w0, b0), (w1, b1)) in enumerate(zip(res["0"], res["1"])):
        ew = ((w0 - w1).abs().max() / (w0.abs().max() + 1e-9)).item(); eb = ((b0 - b1).abs().max() / (b0.abs().max() + 1e-9)).item()
        if ew > 2e-3 or eb > 2e-3:
            print(f"   !! problem {i}: dW rel diff {ew:.2e}, dbias rel diff {eb:.2e}")


T0, T1 = 64000 * G, 8000 * G
s0 = block_set(T0, 96, 64000)
compare("stage-0 block (4 problems)", s0, run_set(s0), sum((p[0].numel() + p[1].numel()) * 2 for p in s0))
del s0; torch.cuda.empty_cache()
s1 = block_set(T1, 192, 8000, extra=((192, 768, False),))
compare("stage-1 block + merge (5)", s1, run_set(s1), sum((p[0].numel() + p[1].numel()) * 2 for p in s1))
del s1; torch.cuda.empty_cache()
B, v, k, Cin, Cout = G, 40, 4, 96, 48
dcat = torch.randn(B * (v * k) ** 3, Cout, device=dev).to(dt)
x = torch.randn(B * v ** 3, Cin, device=dev).to(dt)
dW = torch.zeros(Cin, Cout, k, k, k, device=dev); db = torch.zeros(Cout, device=dev)
up = [(dcat, x, dW, db, None, 0)]
compare("decoder-1 transpose conv", up, lambda: ops.upconv_wgrad_grouped(dcat, x, dW, db, B, v, k, Cin, Cout), (dcat.numel() + x.numel()) * 2)
