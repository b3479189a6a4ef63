"""one stage-2 flush group of the grouped weight-gradient kernel (6 blocks x {qkv, proj, fc1, fc2}, 8 grids) with the attention branch's problems on window
rows (13824, before the token-ordered backward) and on token rows (8000), isolated"""
import sys, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
from nerf_mae_amd._lib import lib
dev = torch.device('cuda')
ws = ops._tn_workspace(dev)
C, B = 384, 8


def run(name, rows_attn, rps_attn, nblk=6, reps=20):
    probs = []
    keep = []
    for _ in range(nblk):
        for M, N, K, rps in ((rows_attn, 3 * C, C, rps_attn), (rows_attn, C, C, rps_attn), (8000, 4 * C, C, 1000), (8000, C, 4 * C, 1000)):
            a = torch.randn(M, N, device=dev, dtype=torch.bfloat16); b = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            w = torch.zeros(N, K, device=dev); bias = torch.zeros(N, device=dev)
            keep += [a, b, w, bias]
            probs.append((a, b, w, bias, M, N, K, rps))
    arr = (ops._TnProblem * len(probs))()
    for i, (a, b, w, bias, M, N, K, rps) in enumerate(probs):
        arr[i] = ops._TnProblem(a.data_ptr(), N, b.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), 0, M, N, K, rps)
    call = lambda: lib().call("nmh_gemm_tn_grouped", ops.BF16, arr, len(probs), ws, ws.numel(), ops._st())
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = sum(2.0 * M * N * K for *_, M, N, K, _ in probs)
    print(f"{name:34s} {nblk} blocks: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s")


run("attention problems on window rows", 13824, 1728)
run("attention problems on token rows", 8000, 1000)
run("attention problems on token rows", 8000, 1000, nblk=9)
run("attention problems on window rows", 13824, 1728, nblk=9)
