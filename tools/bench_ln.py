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


for rows, C in ((8000, 384), (64000, 192), (512000, 96), (64000, 96), (1000, 384), (2000, 384), (4000, 384)):
    run(rows, C)


def run_dyw(B, S, C, shift):
    """the LayerNorm-2 backward of a Swin block as the step issues it: + residual gradient, + second output in window order (dyw)"""
    rows = B * S ** 3
    geom = ops.WinGeom(B, S, S, S, (shift,) * 3)
    x = torch.randn(rows, C, device="cuda").to(dt); dy = torch.randn_like(x); dres = torch.randn_like(x)
    gamma = torch.randn(C, device="cuda"); mean, rstd = torch.zeros(rows, device="cuda"), torch.ones(rows, device="cuda")
    dx = torch.empty_like(x); dyw = torch.empty(geom.rows, C, device="cuda", dtype=dt); sc = torch.ones(B, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    fn = lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db, rows, C, dres=dres, geom=geom, tokens_per_sample=S ** 3, dyw=dyw, dyw_scale=sc)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    print(f"ln_bwd + dyw B={B} {S}^3 C={C} shift={shift}: {a.elapsed_time(b) / 10 * 1e3:.1f} us")


run_dyw(8, 40, 96, 0); run_dyw(8, 40, 96, 2); run_dyw(8, 10, 384, 2); run_dyw(1, 10, 384, 2); run_dyw(2, 10, 384, 2)


def run_dyw_cold(B, S, C, shift):
    """the same launch with every operand cold: 2 GB of other traffic between the timed launches"""
    rows = B * S ** 3
    geom = ops.WinGeom(B, S, S, S, (shift,) * 3)
    x = torch.randn(rows, C, device="cuda").to(dt); dy = torch.randn_like(x); dres = torch.randn_like(x)
    gamma = torch.randn(C, device="cuda"); mean, rstd = torch.zeros(rows, device="cuda"), torch.ones(rows, device="cuda")
    dx = torch.empty_like(x); dyw = torch.empty(geom.rows, C, device="cuda", dtype=dt); sc = torch.ones(B, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    junk = torch.empty(1 << 29, device="cuda"); junk2 = torch.empty(1 << 29, device="cuda")
    fn = lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db, rows, C, dres=dres, geom=geom, tokens_per_sample=S ** 3, dyw=dyw, dyw_scale=sc)
    tot = 0.0
    for i in range(6):
        junk2.copy_(junk)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i: tot += a.elapsed_time(b)
    print(f"ln_bwd + dyw B={B} {S}^3 C={C} shift={shift} COLD: {tot / 5 * 1e3:.1f} us")


run_dyw_cold(8, 40, 96, 2); run_dyw_cold(8, 20, 192, 2); run_dyw_cold(8, 10, 384, 2)
