"""time of the per-step weight pack + optimizer kernels of swin_s"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from nerf_mae_amd.model import build_model
m = build_model('swin_s', 160, 0.75, 0.1).cuda(); m.train(); m.flatten_parameters(); m._ensure_ready(torch.device('cuda'))
def timeit(f, n=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print(f"pack_weights: {timeit(m._packer.run):.3f} ms")
