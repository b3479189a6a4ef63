"""Per-kernel cost of back-to-back tiny kernels inside a replayed HIP graph (the floor a launch-bound stage pays per kernel)."""
import sys
import time
import torch
sys.path.insert(0, ".")
from nerf_mae_amd import ops

x = torch.zeros(1024, device="cuda")
big = torch.zeros(4000 * 384, device="cuda")
for name, buf, n in (("fill 1 elem", x, 1), ("fill 1.5M floats", big, big.numel())):
    N = 1000
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.fill_f32(buf, 0.0, n) if hasattr(ops, "fill_f32") else buf.zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            ops.fill_f32(buf, 0.0, n) if hasattr(ops, "fill_f32") else buf.zero_()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 5 / N * 1e6:.2f} us per kernel in graph replay")
