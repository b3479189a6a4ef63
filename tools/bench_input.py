"""throughput of the input pipeline: stored scene (host) -> padded device batch"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from nerf_mae_amd import data, ops
bt = data.GridBatcher(160, "cuda")
for dt in (np.float32, np.uint8):
    scenes = [data.synthetic_scene((160, 160, 160), i, dtype=dt) for i in range(4)]
    dev = [torch.from_numpy(s).cuda() for s in scenes]
    out = torch.empty(4, 4, 160, 160, 160, device="cuda")
    for name, src in (("host->device+kernel", scenes), ("kernel only (scene resident)", dev)):
        bt(src, flags=[3, 0, 5, 6], out=out); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): bt(src, flags=[3, 0, 5, 6], out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 5 * 1e3
        print(f"{np.dtype(dt).name:8s} {name:30s}: {ms:7.2f} ms per batch of 4  -> {4e3 / ms:7.1f} grids/s")
