"""LayerNorm backward at the encoder stage shapes (rows, C) in a replayed graph of 20 calls (what a training step sees)."""
import sys
import time
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from nerf_mae_amd import ops

dt = torch.bfloat16


def run(rows, C):
    x = torch.randn(rows, C, device="cuda").to(dt)
    dy = torch.randn(rows, C, device="cuda").to(dt)
    dres = torch.randn(rows, C, device="cuda").to(dt)
    gamma = torch.randn(C, device="cuda")
    mean, rstd = torch.zeros(rows, device="cuda"), torch.ones(rows, device="cuda")
    dx = torch.empty_like(x)
    dg, db = torch.zeros(1 << 20, device="cuda"), torch.zeros(1 << 20, device="cuda")   # cold-ish accumulators: a new slice per call
    N = 20
    def body():
        for i in range(N):
            ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg[i * 4096:], db[i * 4096:], rows, C, dres=dres)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print(f"ln_bwd rows={rows} C={C}: {(time.perf_counter() - t) / 10 / N * 1e6:.1f} us per call")


for rows, C in ((8000, 384), (64000, 192), (512000, 96), (64000, 96), (1000, 384)):
    run(rows, C)
