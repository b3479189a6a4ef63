"""grouped weight-gradient GEMM (nmh_gemm_tn_grouped) micro-benchmark: one 16-problem launch of stage-2-like shapes, with distinct operands
per problem or with every problem aliasing the same operands (cache-resident) -- tells a memory-level bound from a CU-level one"""
import sys, ctypes, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
from nerf_mae_amd._lib import lib

dev = torch.device('cuda')
ws = ops._tn_workspace(dev)


def run(name, M, N, K, nprob=16, alias=False, reps=20):
    nA = 1 if alias else nprob
    A = [torch.randn(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nA)]
    B = [torch.randn(M, K, device=dev, dtype=torch.bfloat16) for _ in range(nA)]
    dW = [torch.zeros(N, K, device=dev) for _ in range(nprob)]
    arr = (ops._TnProblem * nprob)()
    for i in range(nprob):
        a, b = A[i % nA], B[i % nA]
        arr[i] = ops._TnProblem(a.data_ptr(), N, b.data_ptr(), K, dW[i].data_ptr(), K, 0, 0, M, N, K, M)
    call = lambda: lib().call("nmh_gemm_tn_grouped", ops.BF16, arr, nprob, ws, ws.numel(), ops._st())
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * N * K * nprob
    tiles = nprob * ((N + 95) // 96) * ((K + 95) // 96)
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:5d} x{nprob:2d} alias={int(alias)} tiles={tiles:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")


for alias in (False, True):
    run("qkv s2", 13824, 1152, 384, alias=alias)
    run("fc1 s2", 8000, 1536, 384, alias=alias)
    run("fc2 s2", 8000, 384, 1536, alias=alias)
    run("proj s2", 13824, 384, 384, alias=alias)
run("qkv s2 x32", 13824, 1152, 384, nprob=32)
run("qkv s2 x40", 13824, 1152, 384, nprob=40)
run("fc1 s2 x36", 8000, 1536, 384, nprob=36)
run("qkv s2 1-grid", 1728, 1152, 384)
run("fc1 s2 1-grid", 1000, 1536, 384)
